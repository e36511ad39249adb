"""Stream-ordering properties of the fused training step that only show when the host does NOT synchronise around it:
  * the targets the host plans from are read on a side stream -- that read must wait for uploads the caller enqueued on the current
    stream just before the step (the real trainer: non_blocking copies from pinned memory behind the spectrogram's DMA, recipe._to_device);
  * an abort of a persistent kernel recorded by the asynchronous latch read must be consumed even when every poll comes too early;
  * the late fold of the clip groups' gradient buffers covers the decoder's parameters only: nothing else may be non-zero in them."""
import random
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _busy(dev, ms=80):
    """Keep the current stream busy for roughly `ms` (fp32 matrix products that nothing reads)."""
    a = torch.randn(8192, 8192, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    a @ a
    e1.record()
    torch.cuda.synchronize()
    n = max(2, int(ms / max(e0.elapsed_time(e1), 0.1)))

    def run():
        for _ in range(n):
            a @ a
    return run


def test_targets_uploaded_just_before_the_step_are_the_ones_planned_from(dev):
    import models
    from piano_a2s_amd import spec, synthetic, train
    from piano_a2s_amd.spec import PAD
    cfg = spec.default_cfg(max_length=(40, 24))
    batch = synthetic.make_batch(12, cfg, 77, frames=151, upper_range=(3, 12), lower_range=(2, 8), full_tail=0.0, full_rows=((3, 1, "up"), (8, 3, "lo")))
    torch.manual_seed(5)
    init = models.ScoreTranscription(**cfg).state_dict()
    busy = _busy(dev)
    res = []
    for racy in (False, True):
        m = models.ScoreTranscription(**cfg)
        m.load_state_dict(init)
        m = m.to(dev).train()
        step = train.TrainStep(m, dropout=False, group_plan={"step_cost": 4.0})
        assert step.early_convstack
        if racy:
            # device tensors that hold STALE targets (all <pad>: a plan made from them decodes nothing) until the uploads below land
            pinned = [t.pin_memory() if torch.is_tensor(t) else t for t in batch]
            dbatch = [torch.full_like(t, PAD, device=dev) if (torch.is_tensor(t) and t.dtype == torch.int64) else
                      (t.to(dev) if torch.is_tensor(t) else t) for t in batch]
            torch.cuda.synchronize()
            busy()                                                   # ... which sit behind ~80 ms of work on the current stream
            for d, p in zip(dbatch, pinned):
                if torch.is_tensor(d) and d.dtype == torch.int64:
                    d.copy_(p, non_blocking=True)
        else:
            dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
            torch.cuda.synchronize()
        losses = step(dbatch, 0.7, rng=random.Random(3))
        torch.cuda.synchronize()
        res.append((losses[:, 0].double().cpu(), step.decode_steps, step.flat.double().cpu(), float(step.opt.ctl[2])))
    (l0, s0, p0, a0), (l1, s1, p1, a1) = res
    assert a0 == 1.0 and a1 == 1.0
    assert s0 == s1 and s0 > 0, f"the plan was made from stale targets: {s1} decode steps against {s0}"
    assert torch.allclose(l0, l1, rtol=2e-6, atol=0), (l0, l1)
    assert float((p0 - p1).abs().max()) <= 5e-6 * float(p0.abs().max())


def test_abort_latch_read_is_consumed_even_if_every_poll_comes_too_early(dev):
    import os
    from piano_a2s_amd import hip
    L = hip.lib()
    latch = hip.abort_latch(dev)
    latch.zero_()
    torch.cuda.synchronize()
    hip.post_persist_abort_read(dev)              # a clean read, consumed
    torch.cuda.synchronize()
    assert hip.poll_persist_abort(dev) == 0
    busy = _busy(dev)
    aborts = hip.PERSIST_ABORTS
    try:
        latch.fill_(4)                            # "a persistent note-decoder launch gave up" ...
        busy()
        hip.post_persist_abort_read(dev)          # ... read behind a busy stream
        assert hip.poll_persist_abort(dev) == 0, "the read cannot have landed yet"
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            hip.post_persist_abort_read(dev)      # the next step's post: the unconsumed read must be looked at, not overwritten
        assert hip.PERSIST_ABORTS == aborts + 1 and w and "persistent" in str(w[0].message)
        assert L.a2s_debug_get(b"dec_persist") == 0 and L.a2s_debug_get(b"gru_persist") == 0
        torch.cuda.synchronize()
        assert hip.poll_persist_abort(dev) == 0 and int(latch.item()) == 0
    finally:
        for key, env in ((b"gru_persist", "A2S_GRU_PERSIST"), (b"dec_persist", "A2S_DEC_PERSIST")):
            hip.check(L.a2s_debug_set(key, 1), "a2s_debug_set")
            os.environ[env] = "1"
        torch.cuda.synchronize()
        latch.zero_()


def test_group_buffers_hold_decoder_gradients_only(dev):
    """engine_bwd.Backward.finish folds flat[late_join:] += group_flat[late_join:] for the groups >= 1: anything a group wrote below the
    decoder's parameters would be lost."""
    import models
    from piano_a2s_amd import engine_bwd, spec, synthetic, train
    cfg = spec.default_cfg(max_length=(40, 24))
    batch = synthetic.make_batch(12, cfg, 78, frames=151, upper_range=(3, 12), lower_range=(2, 8), full_tail=0.0, full_rows=((3, 1, "up"), (8, 3, "lo")))
    dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    torch.manual_seed(6)
    m = models.ScoreTranscription(**cfg).to(dev).train()
    step = train.TrainStep(m, dropout=True, group_plan={"step_cost": 4.0})
    assert step.late_wgrads
    engine_bwd.Backward.check_fold = True
    try:
        losses = step(dbatch, 0.7, rng=random.Random(4))
        torch.cuda.synchronize()
    finally:
        engine_bwd.Backward.check_fold = False
    assert len(step._last[2]) >= 2, "the case must split into clip groups"
    assert torch.isfinite(losses[:, 0]).all() and float(step.opt.ctl[2]) == 1.0
