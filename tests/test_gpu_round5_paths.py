"""Differential check of everything round 5 put on the training step's default path -- the few-clip forward sweep, the attention combine folded into the
few-row GRU step, the long-clip sub-groups, the ConvStack enqueued before the decoder is planned, the backward's weight-gradient products over the
(step, row) pairs that ran only, the late weight gradients, the bar-level attention on the split-T kernels -- against the same step with all of them switched off, on odd shapes of the full-width model:
same losses, same clip norm, same parameters after the update (the switches reorder work and drop exact zeros, nothing else)."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


CASES = [  # B, frames, max_length, bars, tf, full_tail, planner settings
    (12, 151, (40, 24), 5, 0.6, 0.2, {"step_cost": 4.0}),
    (33, 97, (30, 17), 4, 0.8, 0.1, {"step_cost": 8.0}),
    (70, 1201, (14, 9), 3, 0.7, 0.05, {}),
    (5, 203, (21, 33), 3, 0.5, 0.3, {}),
    (20, 64, (12, 9), 5, 1.0, 0.0, {}),
    (9, 301, (50, 20), 5, 0.0, 0.25, {"step_cost": 4.0}),
]


@pytest.mark.parametrize("B,frames,maxlen,bars,tf,tail,plan_kw", CASES)
def test_round5_default_path_equals_the_plain_step(dev, B, frames, maxlen, bars, tf, tail, plan_kw):
    import models
    from piano_a2s_amd import engine, hip, spec, synthetic, train
    L = hip.lib()
    cfg = spec.default_cfg(max_length=maxlen, max_bars=bars)
    batch = synthetic.make_batch(B, cfg, 100 + B, frames=frames, upper_range=(1, maxlen[0]), lower_range=(1, maxlen[1]), full_tail=tail)
    dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    torch.manual_seed(B)
    init = models.ScoreTranscription(**cfg).state_dict()
    keys = (b"attn_defer_combine", b"attn_deep")
    prev = [L.a2s_debug_get(k) for k in keys]
    prev_live, prev_bar = engine._LIVE_ROWS, engine._BAR_ATTN_SPLIT
    res = []
    try:
        for on in (False, True):
            hip.check(L.a2s_debug_set(b"attn_defer_combine", 1 if on else 0), "debug_set")
            hip.check(L.a2s_debug_set(b"attn_deep", 24 if on else 0), "debug_set")
            engine._LIVE_ROWS = on
            engine._BAR_ATTN_SPLIT = on               # (the bar-level attention of groups of >= 32 clips on the split-T kernels)
            m = models.ScoreTranscription(**cfg)
            m.load_state_dict(init)
            m = m.to(dev).train()
            step = train.TrainStep(m, dropout=True, group_plan=plan_kw)
            step.long_subgroups, step.early_convstack, step.late_wgrads = on, on, on
            torch.manual_seed(1234)                   # the dropout masks
            losses = step(dbatch, tf, rng=random.Random(7))
            torch.cuda.synchronize()
            res.append((losses[:, 0].double().cpu(), step.opt.ctl.double().cpu(), step.flat.double().cpu(), step._last[2]))
            del step, m
    finally:
        for k, v in zip(keys, prev):
            hip.check(L.a2s_debug_set(k, v), "debug_set")
        engine._LIVE_ROWS, engine._BAR_ATTN_SPLIT = prev_live, prev_bar
    (l0, c0, p0, g0), (l1, c1, p1, g1) = res
    assert torch.isfinite(l1).all() and float(c1[2]) == 1.0, (l1, c1)
    assert torch.allclose(l0, l1, rtol=2e-6, atol=0), (l0, l1, g0, g1)
    assert abs(float(c0[0]) - float(c1[0])) <= 2e-5 * float(c0[0]), (c0, c1)
    assert float((p0 - p1).abs().max()) <= 5e-6 * float(p0.abs().max()), (float((p0 - p1).abs().max()), g0, g1)
