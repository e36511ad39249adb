"""The persistent few-clip note decoder (csrc/a2s_dec_persist.hip: one launch per NoteDecoder.decode_notes call, one clip per XCD) against
the launch-per-step kernels it replaces: same inputs through both, every output and every per-step tensor the backward pass reads.
(Against the reference itself the path is covered by the full-size golden tests -- g2 / g3 run B <= 4 clips, i.e. through this path.)"""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _cfg():
    from piano_a2s_amd import spec
    return spec.default_cfg(freq_bins=48, max_length=(40, 24))          # the model's widths (hidden 256, note embedding 16), short bars


def _forward(cfg, S, batch, dev, persist, tf, seed, training=True):
    import os
    from piano_a2s_amd import engine, hip
    os.environ["A2S_DEC_PERSIST"] = "1" if persist else "0"
    hip.check(hip.lib().a2s_debug_set(b"dec_persist", 1 if persist else 0), "debug_set")
    eng = engine.Engine(cfg)
    gt = [b.to(dev) for b in batch[1:7]]
    outs = eng.forward(S, batch[0].to(dev), inference=False, ground_truth=gt, teacher_forcing_ratio=tf, training=training, dropout=False,
                       rng=random.Random(seed))
    torch.cuda.synchronize()
    calls = []
    for seg in eng.saved["segments"]:
        for k in ("up", "lo"):
            ids, lengths, sv = seg["staff"][k]
            calls.append(dict(ids=ids.clone(), lengths=lengths.clone(), steps=sv["steps"], used=sv.get("persist_ws") is not None,
                              **{n: sv[n].clone() for n in ("h", "x", "q", "o", "gates", "attw") if sv.get(n) is not None}))
    return [o.clone() for o in outs], calls, eng


@pytest.mark.parametrize("B,frames,tf", [(3, 97, 1.0), (8, 1201, 0.6), (1, 40, 0.0), (5, 301, 0.7)])
def test_persistent_decoder_equals_stepwise(dev, B, frames, tf):
    from piano_a2s_amd import spec, synthetic
    cfg = _cfg()
    st = spec.procedural_state(cfg, 40 + B, eos_bias=2.0, lively="token")
    S = {k: v.to(dev) for k, v in st.items()}
    batch = synthetic.make_batch(B, cfg, 7 + B, frames=frames, upper_range=(5, 30), lower_range=(3, 18), full_tail=0.15, spectrogram="ridges")
    S0 = {k: v.clone() for k, v in S.items()}
    o0, c0, _ = _forward(cfg, S0, batch, dev, False, tf, 3)
    S1 = {k: v.clone() for k, v in S.items()}
    o1, c1, _ = _forward(cfg, S1, batch, dev, True, tf, 3)
    assert all(not c["used"] for c in c0) and all(c["used"] for c in c1), "the switch did not select the intended path"
    assert len(c0) == len(c1)
    for a, b in zip(c0, c1):
        assert a["steps"] == b["steps"]
        assert torch.equal(a["ids"], b["ids"]), "fed-back / reported token ids"
        assert torch.equal(a["lengths"], b["lengths"])
        n = a["steps"]
        # (slot n of x holds only the next token's embedding: its context columns are never written by either path)
        assert torch.equal(a["x"][n][:, :16], b["x"][n][:, :16]), "embedding of the token after the last step"
        for name in ("h", "x", "q", "o", "gates", "attw"):
            ta, tb = a[name][:n + 1 if name == "h" else n], b[name][:n + 1 if name == "h" else n]
            assert torch.isfinite(tb).all(), f"{name}: non-finite (a wait timed out?)"
            err = float((ta - tb).abs().max()) / max(1.0, float(ta.abs().max()))
            assert err < 2e-5, f"{name}: {err:.3e}"
    for name, a, b in zip(("ts", "key", "up", "lo"), o0, o1):
        assert float((a - b).abs().max()) < 2e-5, name


def test_fused_training_step_with_the_persistent_long_clip_group(dev):
    """The fused step (finished rows skipped, bars fused, clip groups, pipelined backward) on a minibatch whose long clips form a group of
    at most 8: loss terms, gradient norm and updated parameters with the persistent path equal those with the launch-per-step path."""
    import os
    import models
    from piano_a2s_amd import engine, hip, synthetic, train
    cfg = _cfg()
    B = 24
    # clips 20..23 hold full-length bars -> the planner's long group
    full = [(b, k, s) for b in (20, 21, 22, 23) for k, s in ((1, "up"), (3, "lo"))]
    batch = synthetic.make_batch(B, cfg, 31, frames=241, upper_range=(4, 12), lower_range=(3, 9), full_tail=0.0, full_rows=full)
    dbatch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    torch.manual_seed(5)
    init = models.ScoreTranscription(**cfg).state_dict()
    res = []
    engine._PERSIST_BESIDE = True          # (off by default: it stops the other group while resident; correctness is what is tested here)
    for persist in (False, True):
        os.environ["A2S_DEC_PERSIST"] = "1" if persist else "0"
        hip.check(hip.lib().a2s_debug_set(b"dec_persist", 1 if persist else 0), "debug_set")
        m = models.ScoreTranscription(**cfg)
        m.load_state_dict(init)
        m = m.to(dev).train()
        step = train.TrainStep(m, dropout=False, clip_groups=[(0, 20), (20, 24)])      # (explicit groups: no planner, no permutation)
        losses = step(dbatch, 0.7, rng=random.Random(11))
        torch.cuda.synchronize()
        res.append((losses[:, 0].double().cpu(), step.opt.ctl.double().cpu(), step.flat.double().cpu(), step._last[2]))
        del step, m
    os.environ["A2S_DEC_PERSIST"] = "1"
    engine._PERSIST_BESIDE = False
    hip.check(hip.lib().a2s_debug_set(b"dec_persist", 1), "debug_set")
    (l0, c0, p0, g0), (l1, c1, p1, g1) = res
    assert g1 is not None and len(g1) == 2 and g1[1][1] - g1[1][0] <= 8, f"expected a long-clip group of at most 8 clips, got {g1}"
    assert torch.isfinite(l1).all() and float(c1[2]) == 1.0
    assert torch.allclose(l0, l1, rtol=2e-5, atol=0), (l0, l1)
    assert abs(float(c0[0]) - float(c1[0])) <= 1e-4 * float(c0[0])
    assert float((p0 - p1).abs().max()) <= 2e-5 * float(p0.abs().max())


@pytest.mark.parametrize("B,frames,tf", [(3, 97, 1.0), (8, 1201, 0.6), (5, 301, 0.0)])
def test_persistent_decoder_backward_equals_stepwise(dev, B, frames, tf):
    """Same forward (persistent), then the reverse loops once as launches per step and once as persistent launches: every parameter
    gradient must agree (the two paths sum in different orders: fp32 round-off only)."""
    import os
    from piano_a2s_amd import engine_bwd, hip, spec, synthetic
    from tests.test_gpu_backward import _loss_grads
    cfg = _cfg()
    st = spec.procedural_state(cfg, 60 + B, eos_bias=2.0, lively="token")
    S = {k: v.to(dev) for k, v in st.items()}
    batch = synthetic.make_batch(B, cfg, 17 + B, frames=frames, upper_range=(5, 30), lower_range=(3, 18), full_tail=0.15, spectrogram="ridges")
    grads = []
    for persist_bwd in (False, True):
        Sx = {k: v.clone() for k, v in S.items()}
        outs, calls, eng = _forward(cfg, Sx, batch, dev, True, tf, 4)
        assert all(c["used"] for c in calls)
        _, gouts = _loss_grads(outs, batch, dev)
        hip.check(hip.lib().a2s_debug_set(b"dec_persist", 1 if persist_bwd else 0), "debug_set")
        G = engine_bwd.backward(eng, Sx, gouts)
        torch.cuda.synchronize()
        grads.append({k: v.clone() for k, v in G.items() if isinstance(k, str) and not k.startswith("__")})
    hip.check(hip.lib().a2s_debug_set(b"dec_persist", 1), "debug_set")
    os.environ["A2S_DEC_PERSIST"] = "1"
    bad = []
    for k in grads[0]:
        a, b = grads[0][k].double(), grads[1][k].double()
        assert torch.isfinite(b).all(), f"{k}: non-finite (a wait timed out?)"
        err = float((a - b).abs().max()) / max(float(a.abs().max()), 1e-12)
        if err > 5e-5:
            bad.append((k, err))
    assert not bad, bad[:8]
