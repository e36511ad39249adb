"""Size-independent properties at the bench's FULL size (BASELINE.json configs[1]: 256 clips of 1201 x 480 frames, 5 bars, the
16.36 M-parameter model, 1 % full-length bars) -- the oracle cannot run this size in test time, so the fused step is checked against
itself: the training-only shortcuts that the bench relies on (finished rows skipped, teacher-forced bars fused, long clips decoded as a
concurrent clip group with its own permutation of the minibatch, decoder backward pipelined behind each group's forward) must leave the
loss terms, the gradient norm and the updated parameters where the plain per-bar step over the whole minibatch puts them."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    if torch.cuda.get_device_properties(0).total_memory < 200 * 2 ** 30:
        pytest.skip("the full-size minibatch needs ~170 GiB of HBM")
    return torch.device("cuda:0")


def test_full_size_minibatch_shortcuts_do_not_change_the_step(dev):
    import models
    from piano_a2s_amd import spec, synthetic, train
    cfg = spec.default_cfg()
    B = 256
    batch = synthetic.make_batch(B, cfg, 1234, full_tail=0.01)              # the bench's first minibatch
    dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    torch.manual_seed(1234)
    init = models.ScoreTranscription(**cfg).state_dict()
    res = []
    for fast in (False, True):
        m = models.ScoreTranscription(**cfg)
        m.load_state_dict(init)
        m = m.to(dev).train()
        step = train.TrainStep(m, dropout=False, skip_finished_rows=fast, fuse_bars=fast, clip_groups=fast)
        losses = step(dbatch, 0.7, rng=random.Random(99))
        torch.cuda.synchronize()
        if fast:
            assert step._last[2] is not None and len(step._last[2]) >= 2, "the full-size minibatch did not split into clip groups"
        res.append((losses[:, 0].double().cpu(), step.opt.ctl.double().cpu(), step.flat.double().cpu(), step.decode_steps))
        del step, m
        torch.cuda.empty_cache()
    (l0, c0, p0, s0), (l1, c1, p1, s1) = res
    assert torch.isfinite(l0).all() and float(c0[2]) == 1.0 and float(c1[2]) == 1.0          # both steps applied their update
    assert s1 < s0, (s0, s1)                                                                  # the shortcuts really ran fewer decode steps
    assert torch.allclose(l0, l1, rtol=2e-5, atol=0), (l0, l1)
    assert abs(float(c0[0]) - float(c1[0])) <= 1e-4 * float(c0[0]), (float(c0[0]), float(c1[0]))      # gradient norm
    assert float((p0 - p1).abs().max()) <= 2e-5 * float(p0.abs().max()), float((p0 - p1).abs().max())


def test_full_size_step_is_invariant_to_the_order_of_the_clips(dev):
    """Nothing in the objective depends on where a clip sits in the minibatch (BatchNorm statistics, the four loss normalisations and
    the gradient are sums over clips): the default fused step on a reversed minibatch -- which changes the planner's permutation and
    every workgroup-to-clip assignment -- gives the same loss terms, gradient norm and update.  Teacher forcing 1.0: the coin protocol
    draws per bar, not per clip, but a free-running bar feeds back its own argmax, where a rounding-level difference could flip a token."""
    import models
    from piano_a2s_amd import spec, synthetic, train
    cfg = spec.default_cfg()
    B = 256
    batch = synthetic.make_batch(B, cfg, 4321, full_tail=0.01)
    rev = torch.arange(B - 1, -1, -1)
    torch.manual_seed(4321)
    init = models.ScoreTranscription(**cfg).state_dict()
    res = []
    for order in (None, rev):
        b = [t.index_select(0, order) if (torch.is_tensor(t) and order is not None and t.dim() > 0 and t.shape[0] == B) else t for t in batch]
        dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in b]
        m = models.ScoreTranscription(**cfg)
        m.load_state_dict(init)
        m = m.to(dev).train()
        step = train.TrainStep(m, dropout=False)
        losses = step(dbatch, 1.0, rng=random.Random(5))
        torch.cuda.synchronize()
        res.append((losses[:, 0].double().cpu(), step.opt.ctl.double().cpu(), step.flat.double().cpu()))
        del step, m, dbatch
        torch.cuda.empty_cache()
    (l0, c0, p0), (l1, c1, p1) = res
    assert float(c0[2]) == 1.0 and float(c1[2]) == 1.0
    assert torch.allclose(l0, l1, rtol=2e-5, atol=0), (l0, l1)
    assert abs(float(c0[0]) - float(c1[0])) <= 1e-4 * float(c0[0]), (float(c0[0]), float(c1[0]))
    assert float((p0 - p1).abs().max()) <= 2e-5 * float(p0.abs().max()), float((p0 - p1).abs().max())


def test_long_clip_subgroups_do_not_change_the_step(dev):
    """Round 5: the long-clip group cut in two by the bar segment of each clip's longest bar (train.split_long_group; three clip groups, the two
    long ones with their staves one after the other on one stream each) against the single long-clip group: same loss, gradient norm, update."""
    import models
    from piano_a2s_amd import spec, synthetic, train
    cfg = spec.default_cfg()
    B = 256
    batch = synthetic.make_batch(B, cfg, 1234, full_tail=0.01)
    dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    torch.manual_seed(1234)
    init = models.ScoreTranscription(**cfg).state_dict()
    res = []
    for sub in (False, True):
        m = models.ScoreTranscription(**cfg)
        m.load_state_dict(init)
        m = m.to(dev).train()
        step = train.TrainStep(m, dropout=False)
        step.long_subgroups = sub
        n3 = 0
        for seed in (99, 100, 101):            # three coin sequences (1 .. 3 bar segments): at least one must make the sub-groups part
            m.load_state_dict(init)
            losses = step(dbatch, 0.7, rng=random.Random(seed))
            torch.cuda.synchronize()
            n3 += len(step._last[2]) == 3
            res.append((losses[:, 0].double().cpu(), step.opt.ctl.double().cpu(), len(step._last[2])))
        if sub:
            assert n3 >= 1, "no step ran as three clip groups"
        del step, m
        torch.cuda.empty_cache()
    for (l0, c0, g0), (l1, c1, g1) in zip(res[:3], res[3:]):
        assert g0 == 2 and float(c0[2]) == 1.0 and float(c1[2]) == 1.0
        assert torch.allclose(l0, l1, rtol=2e-5, atol=0), (l0, l1, g1)
        assert abs(float(c0[0]) - float(c1[0])) <= 1e-4 * float(c0[0]), (float(c0[0]), float(c1[0]), g1)
