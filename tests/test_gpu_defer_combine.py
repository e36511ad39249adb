"""The attention combine folded into the few-row GRU step (csrc/a2s_step.hip dec_gru_step_cmb, round 5: the few-clip launches of a training step
leave their softmax combine to the kernel that consumes the contexts -- one launch less per decode step on the long-clip chain) against the
stand-alone combine kernel: the whole fused training step both ways.  The two paths use the same expressions in the same order, so the forward
outputs must be IDENTICAL; the backward (which reads the contexts / normalised weights the folded combine left in memory, including the zeros of
skipped rows) must agree to rounding of its atomics.  (Round 5 also folded the backward's gate kernel and dq sum into its products and had a
512-thread backward sweep: measured slower, removed in round 6 -- HISTORY.md.)"""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,tf,full_tail", [(6, 0.6, 0.1), (20, 0.5, 0.0), (3, 0.0, 0.3)])
def test_folded_combine_equals_combine_kernel(dev, B, tf, full_tail):
    import models
    from piano_a2s_amd import hip, spec, synthetic, train
    L = hip.lib()
    cfg = spec.default_cfg(freq_bins=48, max_length=(24, 14))
    batch = synthetic.make_batch(B, cfg, 17 + B, frames=97, upper_range=(4, 22), lower_range=(3, 12), full_tail=full_tail)
    dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    torch.manual_seed(9)
    init = models.ScoreTranscription(**cfg).state_dict()
    prev = (L.a2s_debug_get(b"attn_defer_combine"), L.a2s_debug_get(b"dec_persist"))
    res = []
    try:
        hip.check(L.a2s_debug_set(b"dec_persist", 0), "debug_set")       # (<= 8 clips would take the persistent decoder: this test is about the launch-per-step kernels)
        for defer in (0, 1):
            hip.check(L.a2s_debug_set(b"attn_defer_combine", defer), "debug_set")
            m = models.ScoreTranscription(**cfg)
            m.load_state_dict(init)
            m = m.to(dev).train()
            step = train.TrainStep(m, dropout=False, clip_groups=False)
            n0 = L.a2s_launch_count()
            losses = step(dbatch, tf, rng=random.Random(5))
            torch.cuda.synchronize()
            outs = [o.clone() for o in step.last_outputs]
            res.append((losses[:, 0].double().cpu(), step.opt.ctl.double().cpu(), step.flat.double().cpu(), L.a2s_launch_count() - n0, outs))
            del step, m
    finally:
        hip.check(L.a2s_debug_set(b"attn_defer_combine", prev[0]), "debug_set")
        hip.check(L.a2s_debug_set(b"dec_persist", prev[1]), "debug_set")
    (l0, c0, p0, n_sep, o0), (l1, c1, p1, n_fold, o1) = res
    assert n_fold < 0.97 * n_sep, f"the folded kernel must have been taken: {n_fold} launches against {n_sep}"
    from piano_a2s_amd.spec import PAD
    live = {"up": (batch[3] != PAD).to(dev), "lo": (batch[5] != PAD).to(dev)}       # (the fused step leaves positions with <pad> targets unwritten)
    for a, b, name in zip(o0, o1, ("ts", "key", "up", "lo")):
        if name in live:
            a, b = a[live[name]], b[live[name]]
        assert torch.equal(a, b), f"{name}: forward outputs differ, max {float((a - b).abs().max()):.3e}"
    assert torch.isfinite(l1).all() and float(c1[2]) == 1.0
    assert torch.allclose(l0, l1, rtol=1e-6, atol=0), (l0, l1)
    assert abs(float(c0[0]) - float(c1[0])) <= 1e-5 * float(c0[0]), (c0, c1)
    assert float((p0 - p1).abs().max()) <= 2e-6 * float(p0.abs().max())
