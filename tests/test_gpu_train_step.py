"""The drop-in module and the fused training step on the MI355X vs the reference (golden fixtures):
  * models.ScoreTranscription through plain torch autograd (loss.backward()) -> parameter .grad == reference gradients;
  * piano_a2s_amd.train.TrainStep (HIP loss, backward, clip + Adadelta) -> parameters after ONE step == parameters after
    the reference's step (torch.nn.utils.clip_grad_norm_(5.0) + torch.optim.Adadelta run on the reference model);
  * non-finite loss skips the update (SpeechBrain check_gradients semantics)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SMALL_BATCH = dict(frames=41, upper_range=(3, 10), lower_range=(2, 7), full_tail=0.1)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def case(golden_dir):
    from piano_a2s_amd import spec, synthetic
    data = np.load(os.path.join(golden_dir, "g1_small.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g1_small.json")))
    cfg = spec.default_cfg(**meta["cfg"])
    batch = synthetic.make_batch(3, cfg, meta["batch_seed"], **SMALL_BATCH)
    st = spec.procedural_state(cfg, 11, eos_bias=3.0, lively=True)
    return data, cfg, batch, st


def _model(cfg, st, dev):
    import models
    m = models.ScoreTranscription(**cfg)
    m.load_state_dict(st)
    return m.to(dev)


def test_module_state_dict_and_autograd(case, dev, monkeypatch):
    import torch.nn.functional as F
    from piano_a2s_amd import spec
    data, cfg, batch, st = case
    m = _model(cfg, st, dev)
    assert list(m.state_dict().keys()) == list(spec.state_spec(cfg).keys())
    m.train()
    # dropout off on the product side for this comparison: the engine draws masks itself, so patch its switch
    from piano_a2s_amd import engine
    orig = engine.Engine.forward
    monkeypatch.setattr(engine.Engine, "forward", lambda self, *a, **k: orig(self, *a, **{**k, "dropout": False}))
    gt = [b.to(dev) for b in batch[1:7]]
    ts, key, up, lo = m(spectrogram=batch[0].to(dev), inference=False, ground_truth=gt, teacher_forcing_ratio=1.0, device=dev)
    nll, nll_pad = torch.nn.NLLLoss(), torch.nn.NLLLoss(ignore_index=147)
    loss = (nll(ts.permute(0, 2, 1), gt[0]) + nll(key.permute(0, 2, 1), gt[1])
            + nll_pad(up.view(-1, up.shape[2], up.shape[3]).permute(0, 2, 1), gt[2].view(-1, gt[2].shape[2]))
            + nll_pad(lo.view(-1, lo.shape[2], lo.shape[3]).permute(0, 2, 1), gt[4].view(-1, gt[4].shape[2])))
    assert abs(float(loss) - data["train_tf1.losses"][0]) <= 1e-4 * data["train_tf1.losses"][0]
    loss.backward()
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        ref = data[f"train_tf1.grad.{k}"]
        err = np.abs(p.grad.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-12)
        assert err <= 2e-4, f"{k}: {err:.3e}"


def test_fused_step_matches_reference_step(case, dev):
    from piano_a2s_amd import train
    data, cfg, batch, st = case
    m = _model(cfg, st, dev)
    m.train()
    step = train.TrainStep(m, lr=1.0, rho=0.95, eps=1e-8, max_grad_norm=5.0, dropout=False)
    dbatch = [b.to(dev) if torch.is_tensor(b) else b for b in batch]
    losses = step(dbatch, teacher_forcing_ratio=1.0)
    torch.cuda.synchronize()
    got = losses[:, 0].cpu().numpy()
    ref = data["train_tf1.losses"]
    for i in range(4):
        assert abs(got[i] - ref[i + 1]) <= 1e-4 * abs(ref[i + 1]), f"loss term {i}: {got[i]} vs {ref[i + 1]}"
    ctl = step.opt.ctl.cpu().numpy()
    assert abs(ctl[0] - float(data["step.total_norm"])) <= 2e-4 * float(data["step.total_norm"]), f"grad norm {ctl[0]}"
    assert ctl[2] == 1.0
    worst = 0.0
    for k, p in m.named_parameters():
        ref_p = data[f"step.param.{k}"]
        err = np.abs(p.detach().cpu().numpy() - ref_p).max() / max(np.abs(ref_p).max(), 1e-12)
        worst = max(worst, err)
        assert err <= 1e-4, f"updated {k}: {err:.3e}"
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/grad_errors.txt", "a") as f:
        f.write(f"fused step: worst updated-parameter error {worst:.3e}, grad norm {ctl[0]} vs {float(data['step.total_norm'])}\n")


def test_nonfinite_loss_skips_update(case, dev):
    from piano_a2s_amd import train
    data, cfg, batch, st = case
    m = _model(cfg, st, dev)
    flat = m.flatten_()
    opt = train.FusedAdadelta(flat)
    before = flat.clone()
    g = torch.randn_like(flat)
    bad = torch.tensor([float("nan")], device=dev)
    opt.step(g, bad, zero_grad=True)
    torch.cuda.synchronize()
    assert torch.equal(flat, before) and float(opt.ctl[2]) == 0.0 and float(g.abs().sum()) == 0.0
    good = torch.tensor([1.0], device=dev)
    g = torch.randn_like(flat)
    opt.step(g, good, zero_grad=False)
    torch.cuda.synchronize()
    assert not torch.equal(flat, before) and float(opt.ctl[2]) == 1.0


def test_sync_bn_single_rank_equals_local_bn(case, dev):
    """Synchronised-BatchNorm code path (all-reduced statistics, split backward) with a 1-rank RCCL group must reproduce the per-rank
    path: same loss, same updated parameters.  (With N ranks it makes the N-GPU step equal to one GPU on the N-fold batch.)"""
    import torch.distributed as dist
    from piano_a2s_amd import train
    data, cfg, batch, st = case
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        results = []
        for sync in (False, True):
            m = _model(cfg, st, dev)
            m.train()
            step = train.TrainStep(m, dropout=False, sync_bn=sync)
            dbatch = [b.to(dev) if torch.is_tensor(b) else b for b in batch]
            losses = step(dbatch, teacher_forcing_ratio=1.0)
            torch.cuda.synchronize()
            results.append((losses[:, 0].cpu().clone(), m.flatten_().cpu().clone(), {k: v.cpu().clone() for k, v in m.named_buffers()}))
        (l0, p0, b0), (l1, p1, b1) = results
        assert torch.allclose(l0, l1, rtol=1e-5, atol=1e-6), (l0, l1)
        assert float((p0 - p1).abs().max()) <= 1e-5 * float(p0.abs().max())
        for k in b0:
            assert torch.allclose(b0[k].float(), b1[k].float(), rtol=1e-5, atol=1e-6), k
    finally:
        dist.destroy_process_group()


def test_full_width_step_with_flat_parameters(dev):
    """The fused training step on the 256-wide model whose parameters are views of ONE flat buffer (spec.flat_layout puts every tensor
    on a 16-byte boundary; the odd-sized ones -- 173-row vocabulary matrices, biases -- are what used to land on 4-byte offsets): the
    16-byte-load kernels run on the real parameter layout (separately allocated test tensors are always aligned)."""
    import models
    from piano_a2s_amd import spec, synthetic, train
    cfg = spec.default_cfg(max_length=(12, 8), max_bars=2)
    m = models.ScoreTranscription(**cfg).to(dev)
    m.train()
    step = train.TrainStep(m, dropout=True)
    batch = synthetic.make_batch(3, cfg, 9, frames=57, upper_range=(3, 10), lower_range=(2, 7), full_tail=0.2)
    dbatch = [b.to(dev) if torch.is_tensor(b) else b for b in batch]
    before = step.flat.clone()
    losses = step(dbatch, teacher_forcing_ratio=0.7)
    torch.cuda.synchronize()
    assert torch.isfinite(losses).all() and float(step.opt.ctl[2]) == 1.0
    assert not torch.equal(before, step.flat)


@pytest.mark.parametrize("tf_ratio,seed", [(0.6, 4), (1.0, 1), (0.85, 7), (0.0, 2)])
def test_skipping_rows_and_fusing_bars_do_not_change_the_step(dev, tf_ratio, seed):
    """TrainStep's two training-only shortcuts vs the plain step on the 256-wide model (the split-T kernels):
      skip_finished_rows -- attention skipped for rows whose remaining targets are all <pad>;
      fuse_bars          -- consecutive teacher-forced bars decoded in one call (multi-row-per-clip attention kernels).
    Identical loss terms, gradient norm and updated parameters; ragged lengths, mixed teacher forcing (tf 1.0: all 5 bars in one call,
    0.0: never fused), full-length rows without <eos> included."""
    import random
    import models
    from piano_a2s_amd import spec, synthetic, train
    cfg = spec.default_cfg(max_length=(14, 9), max_bars=5)
    st = spec.procedural_state(cfg, 5, eos_bias=1.0, lively="token")
    batch = synthetic.make_batch(5, cfg, 21, frames=61, upper_range=(1, 14), lower_range=(1, 9), full_tail=0.2, spectrogram="ridges")
    dbatch = [b.to(dev) if torch.is_tensor(b) else b for b in batch]
    res = []
    # ... and clip groups: the clips decode as two concurrent groups with their own step counts (explicit ranges, then the planner's
    # own cut with its permutation of the minibatch)
    for skip, fuse, groups in ((False, False, False), (True, False, False), (True, True, False), (True, True, [(0, 2), (2, 5)]), (True, True, "plan")):
        m = models.ScoreTranscription(**cfg)
        m.load_state_dict(st)
        m = m.to(dev)
        m.train()
        step = train.TrainStep(m, dropout=False, skip_finished_rows=skip, fuse_bars=fuse, clip_groups=groups if groups != "plan" else True)
        if groups == "plan":
            step_plan = train.plan_clip_groups
            train.plan_clip_groups = lambda up, lo, **kw: step_plan(up, lo, min_gain=-1e9, jump=1.0)       # always cut at the largest jump
        try:
            losses = step(dbatch, teacher_forcing_ratio=tf_ratio, rng=random.Random(seed))
        finally:
            if groups == "plan":
                train.plan_clip_groups = step_plan
        torch.cuda.synchronize()
        if groups:
            assert len(step._last[2]) >= 2, "the step did not run as clip groups"
            assert (step._last[3] is not None) == (groups == "plan")
        res.append((losses[:, 0].cpu().clone(), step.opt.ctl.cpu().clone(), step.flat.cpu().clone(), [o.cpu() for o in step.last_outputs]))
    l0, c0, p0, o0 = res[0]
    assert torch.isfinite(l0).all() and float(c0[2]) == 1.0
    for variant, (l1, c1, p1, o1) in enumerate(res[1:], start=1):
        assert float(c1[2]) == 1.0
        assert torch.allclose(l0, l1, rtol=2e-6, atol=0), (variant, l0, l1)
        assert abs(float(c0[0]) - float(c1[0])) <= 1e-5 * float(c0[0]), variant
        assert float((p0 - p1).abs().max()) <= 2e-6 * float(p0.abs().max()), variant
        # positions that count (target != <pad>) are untouched; the skipping really happened (some padded position differs)
        for out0, out1, gt in ((o0[2], o1[2], batch[3]), (o0[3], o1[3], batch[5])):
            assert out0.shape == out1.shape
            keep = gt != 147
            err = (out0[keep] - out1[keep]).abs().amax(-1)
            assert float(err.max()) <= 5e-6, (variant, float(err.max()), keep.nonzero()[err > 2e-6][:6].tolist())
        assert torch.allclose(o0[0], o1[0], atol=2e-6) and torch.allclose(o0[1], o1[1], atol=2e-6), variant
        assert not torch.equal(o0[2], o1[2])


@pytest.mark.parametrize("B,T,maxlen,bars,tf", [(1, 37, (9, 5), 2, 0.7), (5, 64, (3, 2), 4, 0.0), (7, 203, (21, 33), 3, 0.5)])
def test_odd_shapes_through_the_fused_step_and_the_greedy_decoder(dev, B, T, maxlen, bars, tf):
    """Full-width model on shapes that are not multiples of any tile size (tools/robustness_sweep.py has the longer list): a single clip of
    37 frames (fewer rows than one GEMM tile -- the BatchNorm-statistics epilogue of the Linear's data gradient must step aside), bars of 2-3
    tokens, no teacher forcing.  Finite losses, an applied update, finite greedy outputs."""
    import random

    import models
    from piano_a2s_amd import spec, synthetic, train
    cfg = spec.default_cfg(max_length=maxlen, max_bars=bars)
    torch.manual_seed(B)
    m = models.ScoreTranscription(**cfg).to(dev)
    m.train()
    step = train.TrainStep(m)
    batch = synthetic.make_batch(B, cfg, 3 + B, frames=T, upper_range=(1, maxlen[0]), lower_range=(1, maxlen[1]), full_tail=0.15)
    batch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    for k in range(2):
        losses = step(batch, tf, rng=random.Random(k))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(losses).all()) and float(step.opt.ctl[2]) == 1.0
    m.eval()
    with torch.no_grad():
        outs = m(spectrogram=batch[0], inference=True, device=dev)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(o).all()) for o in outs)


def test_fused_few_row_decoder_step_matches_the_library_path(dev):
    """csrc/a2s_step.hip (4 launches per decode step forward, 5 backward, used for calls of few rows) against the GEMM-by-GEMM step
    (a2s_debug_set('dec_fused', 0)): the same training step -- losses, gradient norm, updated parameters, log-probabilities -- and
    the same greedy decode (identical token ids), on the full-width model with ragged rows, mixed teacher forcing and dropout off."""
    import random
    import models
    from piano_a2s_amd import hip, spec, synthetic, train
    L = hip.lib()
    cfg = spec.default_cfg(max_length=(14, 9), max_bars=3)
    st = spec.procedural_state(cfg, 5, eos_bias=1.0, lively="token")
    batch = synthetic.make_batch(5, cfg, 21, frames=61, upper_range=(1, 14), lower_range=(1, 9), full_tail=0.2, spectrogram="ridges")
    dbatch = [b.to(dev) if torch.is_tensor(b) else b for b in batch]
    assert L.a2s_debug_get(b"dec_fused") == 1 and L.a2s_debug_get(b"dec_fused_max_rows") >= 15
    res = {}
    try:
        for fused in (0, 1):
            L.a2s_debug_set(b"dec_fused", fused)
            m = models.ScoreTranscription(**cfg)
            m.load_state_dict(st)
            m = m.to(dev)
            m.eval()
            with torch.no_grad():
                greedy = [o.cpu() for o in m(spectrogram=dbatch[0], inference=True, device=dev)]
            m.train()
            step = train.TrainStep(m, dropout=False)
            losses = step(dbatch, teacher_forcing_ratio=0.6, rng=random.Random(4))
            torch.cuda.synchronize()
            res[fused] = (losses[:, 0].cpu().clone(), step.opt.ctl.cpu().clone(), step.flat.cpu().clone(), [o.cpu() for o in step.last_outputs], greedy)
    finally:
        L.a2s_debug_set(b"dec_fused", 1)
    (l0, c0, p0, o0, g0), (l1, c1, p1, o1, g1) = res[0], res[1]
    assert torch.isfinite(l0).all() and float(c0[2]) == 1.0 and float(c1[2]) == 1.0
    assert torch.allclose(l0, l1, rtol=5e-6, atol=0), (l0, l1)
    assert abs(float(c0[0]) - float(c1[0])) <= 2e-5 * float(c0[0])
    assert float((p0 - p1).abs().max()) <= 5e-6 * float(p0.abs().max())
    for a, b, gt in ((o0[2], o1[2], batch[3]), (o0[3], o1[3], batch[5])):
        keep = gt != 147
        assert float((a[keep] - b[keep]).abs().max()) <= 2e-5
    for a, b in zip(g0, g1):
        assert torch.equal(a.argmax(-1), b.argmax(-1)) and float((a - b).abs().max()) <= 2e-5
    assert torch.equal((g0[2].abs().sum(-1) > 0), (g1[2].abs().sum(-1) > 0)), "executed greedy steps differ"
