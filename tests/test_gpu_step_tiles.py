"""The fused decoder-step kernels with several 16-row tiles per workgroup (csrc/a2s_step.hip dec_gru_step_rt / dec_bwd_products_rt, round 5)
against the tiled-GEMM step they replace on calls of hundreds of rows: the whole fused training step both ways -- loss terms, gradient norm,
updated parameters.  80 clips x 5 teacher-forced bars = 400 rows per call -> 4 tiles per workgroup; 40 clips -> 200 rows -> 2."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("B", [80, 40])
def test_row_tiled_step_kernels_equal_tiled_gemm_step(dev, B):
    import models
    from piano_a2s_amd import hip, spec, synthetic, train
    L = hip.lib()
    cfg = spec.default_cfg(freq_bins=48, max_length=(24, 14))
    batch = synthetic.make_batch(B, cfg, 61, frames=61, upper_range=(4, 22), lower_range=(3, 12), full_tail=0.05)
    dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    torch.manual_seed(9)
    init = models.ScoreTranscription(**cfg).state_dict()
    prev = L.a2s_debug_get(b"dec_fused_max_rows")
    res = []
    try:
        for max_rows in (0, 4096):
            hip.check(L.a2s_debug_set(b"dec_fused_max_rows", max_rows), "debug_set")
            m = models.ScoreTranscription(**cfg)
            m.load_state_dict(init)
            m = m.to(dev).train()
            step = train.TrainStep(m, dropout=False, clip_groups=False)
            n0 = L.a2s_launch_count()
            losses = step(dbatch, 1.0, rng=random.Random(5))            # tf = 1: all five bars in one call of 5 B rows
            torch.cuda.synchronize()
            res.append((losses[:, 0].double().cpu(), step.opt.ctl.double().cpu(), step.flat.double().cpu(), L.a2s_launch_count() - n0))
            del step, m
    finally:
        hip.check(L.a2s_debug_set(b"dec_fused_max_rows", prev), "debug_set")
    (l0, c0, p0, n_tiled), (l1, c1, p1, n_fused) = res
    assert n_fused < 0.8 * n_tiled, f"the fused step must have been taken: {n_fused} launches against {n_tiled}"
    assert torch.isfinite(l1).all() and float(c1[2]) == 1.0
    assert torch.allclose(l0, l1, rtol=2e-5, atol=0), (l0, l1)
    assert abs(float(c0[0]) - float(c1[0])) <= 1e-4 * float(c0[0]), (c0, c1)
    assert float((p0 - p1).abs().max()) <= 2e-5 * float(p0.abs().max())
