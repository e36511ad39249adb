"""The mid-size fused decoder-step kernels (csrc/a2s_step.hip dec_gru_mid / dec_bwd_mid, round 6: 64-row workgroups, weights staged through LDS)
against the library-style step they replace on calls of hundreds of rows: the whole fused training step three ways -- (a) every per-step product
a tiled GEMM launch (a2s_debug_set("dec_mid", 0), the few-row path off), (b) the bulk loop with the mid-size kernels (the default for calls
above the few-row limit), (c) the few-row path forced onto every call, which hands launches over more than 160 rows to the same kernels -- loss
terms, gradient norm, updated parameters.  80 clips x 5 teacher-forced bars = 400 rows per call; 40 clips -> 200 rows."""
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
def test_mid_size_step_kernels_equal_tiled_gemm_step(dev, B):
    import models
    from piano_a2s_amd import hip, spec, synthetic, train
    L = hip.lib()
    cfg = spec.default_cfg(freq_bins=48, max_length=(24, 14))
    batch = synthetic.make_batch(B, cfg, 61, frames=61, upper_range=(4, 22), lower_range=(3, 12), full_tail=0.05)
    dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    torch.manual_seed(9)
    init = models.ScoreTranscription(**cfg).state_dict()
    prev = L.a2s_debug_get(b"dec_fused_max_rows"), L.a2s_debug_get(b"dec_mid")
    res = []
    try:
        for max_rows, mid in ((0, 0), (0, 1), (4096, 1)):
            hip.check(L.a2s_debug_set(b"dec_fused_max_rows", max_rows), "debug_set")
            hip.check(L.a2s_debug_set(b"dec_mid", mid), "debug_set")
            m = models.ScoreTranscription(**cfg)
            m.load_state_dict(init)
            m = m.to(dev).train()
            step = train.TrainStep(m, dropout=False, clip_groups=False)
            n0, m0 = L.a2s_launch_count(), L.a2s_debug_get(b"dec_mid_launches")
            losses = step(dbatch, 1.0, rng=random.Random(5))            # tf = 1: all five bars in one call of 5 B rows
            torch.cuda.synchronize()
            res.append((losses[:, 0].double().cpu(), step.opt.ctl.double().cpu(), step.flat.double().cpu(), L.a2s_launch_count() - n0,
                        L.a2s_debug_get(b"dec_mid_launches") - m0, step.decode_steps))
            del step, m
    finally:
        hip.check(L.a2s_debug_set(b"dec_fused_max_rows", prev[0]), "debug_set")
        hip.check(L.a2s_debug_set(b"dec_mid", prev[1]), "debug_set")
    (l0, c0, p0, n_tiled, mid0, steps), rest = res[0], res[1:]
    assert mid0 == 0, "the baseline must not have used the mid-size kernels"
    for name, (l1, c1, p1, n1, mid1, _) in zip(("bulk loop", "few-row path"), rest):
        assert n1 < 0.9 * n_tiled, f"{name}: {n1} launches against {n_tiled}"
        # (the bulk loop: one dec_gru_mid per forward step and one dec_bwd_mid per backward step; the few-row path: only its launches over > 160 rows)
        assert mid1 >= (steps if name == "bulk loop" else 1), f"{name}: {mid1} mid-size launches for {steps} decode steps"
        assert torch.isfinite(l1).all() and float(c1[2]) == 1.0
        assert torch.allclose(l0, l1, rtol=2e-5, atol=0), (name, l0, l1)
        assert abs(float(c0[0]) - float(c1[0])) <= 1e-4 * float(c0[0]), (name, c0, c1)
        assert float((p0 - p1).abs().max()) <= 2e-5 * float(p0.abs().max()), name
