"""The two paths of the persistent kernels that an undisturbed MI355X never takes: the agent-scope (write-through) granule hand-off
used when the workgroups that exchange state do NOT share an XCD, and the bounded-wait abort (outputs poisoned with NaN where the caller
reads them, process-wide latch set, host switches the persistent paths off).  Both are forced through a2s_debug_set hooks:
"persist_force_agent", "persist_inject_abort" (include/a2s.h)."""
import ctypes as C
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


@pytest.fixture
def hooks(dev):
    """Sets / clears the hooks and restores every switch afterwards (also the ones an observed abort turns off)."""
    import os
    from piano_a2s_amd import hip
    L = hip.lib()
    hip.abort_latch(dev).zero_()

    def set_(key, v):
        hip.check(L.a2s_debug_set(key, v), "a2s_debug_set")
    yield set_
    for key in (b"persist_force_agent", b"persist_inject_abort"):
        set_(key, 0)
    for key, env in ((b"gru_persist", "A2S_GRU_PERSIST"), (b"dec_persist", "A2S_DEC_PERSIST")):
        set_(key, 1)
        os.environ[env] = "1"
    torch.cuda.synchronize()
    hip.abort_latch(dev).zero_()


def test_device_geometry_is_queried(dev):
    from piano_a2s_amd import hip
    L = hip.lib()
    cus, xccs = L.a2s_debug_get(b"device_cus"), L.a2s_debug_get(b"device_xccs")
    assert cus == torch.cuda.get_device_properties(0).multi_processor_count and xccs >= 1, (cus, xccs)


@pytest.mark.parametrize("B,T", [(37, 29), (256, 33)])
def test_encoder_agent_scope_handoff(dev, hooks, B, T):
    from piano_a2s_amd import hip
    from tests.test_gpu_persist import _rel, _run_direction
    L = hip.lib()
    H = 256
    g = torch.Generator().manual_seed(300 + B)
    gi = (torch.randn(B, T, 3 * H, generator=g) * 0.8).to(dev)
    w_hh = (torch.randn(3 * H, H, generator=g) * 0.08).to(dev)
    b_hh = (torch.randn(3 * H, generator=g) * 0.1).to(dev)
    dout = torch.randn(B, T, 2 * H, generator=g).to(dev)
    dhn = torch.randn(B, H, generator=g).to(dev)
    a = _run_direction(L, hip, dev, gi, w_hh, b_hh, dout, dhn, B, T, H, 0, persist=False)
    hooks(b"persist_force_agent", 1)
    b = _run_direction(L, hip, dev, gi, w_hh, b_hh, dout, dhn, B, T, H, 0, persist=True)
    for k in ("out", "hn", "gates"):
        assert torch.equal(a[k], b[k]), k
    for k in ("dgi", "dghs", "dgh_first"):
        assert torch.isfinite(b[k]).all() and _rel(b[k], a[k]) < 2e-6, k
    assert int(hip.abort_latch(dev).item()) == 0


def test_encoder_abort_poisons_and_latches(dev, hooks):
    from piano_a2s_amd import hip
    from tests.test_gpu_persist import _run_direction
    L = hip.lib()
    B, T, H = 37, 29, 256
    g = torch.Generator().manual_seed(9)
    gi = (torch.randn(B, T, 3 * H, generator=g) * 0.8).to(dev)
    w_hh = (torch.randn(3 * H, H, generator=g) * 0.08).to(dev)
    b_hh = (torch.randn(3 * H, generator=g) * 0.1).to(dev)
    dout = torch.randn(B, T, 2 * H, generator=g).to(dev)
    dhn = torch.randn(B, H, generator=g).to(dev)
    hooks(b"persist_inject_abort", 1)
    r = _run_direction(L, hip, dev, gi, w_hh, b_hh, dout, dhn, B, T, H, 0, persist=True)
    assert torch.isnan(r["hn"]).all(), "the final state (which the bridge and the loss depend on) must be poisoned"
    assert torch.isnan(r["dgh_first"][:, :H]).all(), "what the deferred weight-gradient products read must be poisoned (the r-gate columns of every row)"
    assert int(hip.abort_latch(dev).item()) & 3 == 3
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        bits = hip.check_persist_abort(dev)
    assert bits & 3 == 3 and w and "persistent" in str(w[0].message)
    assert L.a2s_debug_get(b"gru_persist") == 0 and L.a2s_debug_get(b"dec_persist") == 0, "an observed abort switches the persistent paths off"
    assert hip.check_persist_abort(dev) == 0, "the latch is cleared once reported"


@pytest.mark.parametrize("B,frames,tf", [(3, 97, 1.0), (8, 301, 0.6)])
def test_decoder_agent_scope_handoff(dev, hooks, B, frames, tf):
    from piano_a2s_amd import spec, synthetic
    from tests.test_gpu_dec_persist import _cfg, _forward
    cfg = _cfg()
    st = spec.procedural_state(cfg, 140 + B, eos_bias=2.0, lively="token")
    S = {k: v.to(dev) for k, v in st.items()}
    batch = synthetic.make_batch(B, cfg, 27 + B, frames=frames, upper_range=(5, 30), lower_range=(3, 18), full_tail=0.15, spectrogram="ridges")
    o0, c0, _ = _forward(cfg, {k: v.clone() for k, v in S.items()}, batch, dev, False, tf, 3)
    hooks(b"persist_force_agent", 1)
    o1, c1, _ = _forward(cfg, {k: v.clone() for k, v in S.items()}, batch, dev, True, tf, 3)
    assert all(c["used"] for c in c1)
    for a, b in zip(c0, c1):
        assert a["steps"] == b["steps"] and torch.equal(a["ids"], b["ids"]) and torch.equal(a["lengths"], b["lengths"])
        n = a["steps"]
        for name in ("h", "x", "q", "o", "gates", "attw"):
            ta, tb = a[name][:n + 1 if name == "h" else n], b[name][:n + 1 if name == "h" else n]
            assert torch.isfinite(tb).all(), name
            assert float((ta - tb).abs().max()) / max(1.0, float(ta.abs().max())) < 2e-5, name
    for name, a, b in zip(("ts", "key", "up", "lo"), o0, o1):
        assert float((a - b).abs().max()) < 2e-5, name


def test_decoder_abort_skips_the_update_then_falls_back(dev, hooks):
    """A persistent note-decoder launch that gives up: the step's loss is non-finite, the update is skipped ON THE DEVICE (ctl[2] == 0, parameters
    untouched); the next step sees the latch, switches to the launch-per-step kernels and trains normally."""
    import models
    from piano_a2s_amd import hip, synthetic, train
    from tests.test_gpu_dec_persist import _cfg
    L = hip.lib()
    cfg = _cfg()
    batch = synthetic.make_batch(4, cfg, 41, frames=121, upper_range=(4, 12), lower_range=(3, 9), full_tail=0.1)
    dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    torch.manual_seed(3)
    m = models.ScoreTranscription(**cfg).to(dev).train()
    step = train.TrainStep(m, dropout=False, clip_groups=False)       # (one clip group: the 4 clips decode on the persistent path)
    before = step.flat.clone()
    hooks(b"gru_persist", 0)                       # (the encoder's own abort would poison the step as well: this test is about the decoder's)
    hooks(b"persist_inject_abort", 1)
    launches = L.a2s_debug_get(b"dec_persist_launches")
    losses = step(dbatch, 0.7, rng=random.Random(2))
    torch.cuda.synchronize()
    assert L.a2s_debug_get(b"dec_persist_launches") > launches, "the 4-clip step must have taken the persistent decoder"
    assert not torch.isfinite(losses[:, 0]).all(), "the poison must reach the loss"
    assert float(step.opt.ctl[2]) == 0.0 and torch.equal(step.flat, before), "non-finite loss: update skipped, parameters untouched"
    assert int(hip.abort_latch(dev).item()) & 4
    hooks(b"persist_inject_abort", 0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        losses = step(dbatch, 0.7, rng=random.Random(2))
        torch.cuda.synchronize()
    assert w and "persistent" in str(w[0].message)
    assert L.a2s_debug_get(b"dec_persist") == 0
    assert torch.isfinite(losses[:, 0]).all() and float(step.opt.ctl[2]) == 1.0 and not torch.equal(step.flat, before)


def test_greedy_decode_abort_raises(dev, hooks):
    from piano_a2s_amd import engine, hip, spec, synthetic
    from tests.test_gpu_dec_persist import _cfg
    cfg = _cfg()
    st = spec.procedural_state(cfg, 77, eos_bias=2.0, lively="token")
    S = {k: v.to(dev) for k, v in st.items()}
    batch = synthetic.make_batch(2, cfg, 5, frames=97, upper_range=(5, 30), lower_range=(3, 18), full_tail=0.0, spectrogram="ridges")
    hooks(b"gru_persist", 0)
    hooks(b"persist_inject_abort", 1)
    with pytest.raises(hip.A2SError, match="persistent"):
        engine.Engine(cfg).forward(S, batch[0].to(dev), inference=True)
    torch.cuda.synchronize()
