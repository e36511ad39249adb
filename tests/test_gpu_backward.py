"""Full training-step gradient parity on the MI355X: HIP forward + manual HIP backward vs the REFERENCE's gradients
(fixtures from tests/golden/make_golden.py: all 83 parameter gradients at reduced size, norms + samples at full size)."""
import json
import os
import random

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


def _loss_grads(outs, batch, dev):
    """d(total loss)/d(outputs) with the reference's objective (pretrain.py:72-88), evaluated by torch on the device."""
    leaves = [o.detach().clone().requires_grad_(True) for o in outs]
    nll, nll_pad = torch.nn.NLLLoss(), torch.nn.NLLLoss(ignore_index=147)
    ts_t, key_t, up_t, lo_t = batch[1].to(dev), batch[2].to(dev), batch[3].to(dev), batch[5].to(dev)
    ts_o, key_o, up_o, lo_o = leaves
    terms = [nll(ts_o.permute(0, 2, 1), ts_t), nll(key_o.permute(0, 2, 1), key_t),
             nll_pad(up_o.view(-1, up_o.shape[2], up_o.shape[3]).permute(0, 2, 1), up_t.view(-1, up_t.shape[2])),
             nll_pad(lo_o.view(-1, lo_o.shape[2], lo_o.shape[3]).permute(0, 2, 1), lo_t.view(-1, lo_t.shape[2]))]
    total = sum(terms)
    total.backward()
    return [float(total)] + [float(t) for t in terms], [l.grad for l in leaves]


def _log(line):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/grad_errors.txt", "a") as f:
        f.write(line + "\n")


@pytest.mark.parametrize("name", ["train_tf1", "train_tf05"])
def test_small_model_all_gradients(golden_dir, dev, name):
    from piano_a2s_amd import engine, engine_bwd, spec, synthetic
    data = np.load(os.path.join(golden_dir, "g1_small.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g1_small.json")))
    cfg = spec.default_cfg(**meta["cfg"])
    case = meta["cases"][name]
    batch = synthetic.make_batch(3, cfg, meta["batch_seed"], **SMALL_BATCH)
    S = {k: v.to(dev) for k, v in spec.procedural_state(cfg, case["weights_seed"], eos_bias=case["eos_bias"], lively=True).items()}
    gt = [b.to(dev) for b in batch[1:7]]
    rng = random.Random()
    if case["random_seed"] is not None:
        rng.seed(case["random_seed"])
    eng = engine.Engine(cfg)
    outs = eng.forward(S, batch[0].to(dev), inference=False, ground_truth=gt, teacher_forcing_ratio=case["tf"], training=True, rng=rng, dropout=False)
    losses, gouts = _loss_grads(outs, batch, dev)
    ref_losses = data[f"{name}.losses"]
    for i, (l, r) in enumerate(zip(losses, ref_losses)):
        assert abs(l - r) <= 1e-4 * abs(r), f"loss term {i}: {l} vs reference {r}"
    G = engine_bwd.backward(eng, S, gouts)
    torch.cuda.synchronize()
    worst = 0.0
    failures = []
    for k in S:
        if spec.is_buffer(k):
            continue
        g = G[k].cpu().numpy().astype(np.float64)
        if name == "train_tf1":
            ref = data[f"{name}.grad.{k}"].astype(np.float64)
            err = np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-12)
        else:
            ref = float(data[f"{name}.gradnorm.{k}"])
            err = abs(np.linalg.norm(g) - ref) / max(ref, 1e-12)
        _log(f"{name} {k}: {err:.3e}")
        worst = max(worst, err)
        if err > 2e-4:
            failures.append((k, err))
    assert not failures, f"{len(failures)} parameter gradients off: {failures[:8]}"


def test_full_size_gradient_norms(golden_dir, dev):
    from piano_a2s_amd import engine, engine_bwd, spec, synthetic
    data = np.load(os.path.join(golden_dir, "g2_full.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g2_full.json")))
    cfg = spec.default_cfg()
    st = spec.procedural_state(cfg, meta["weights_seed"], eos_bias=meta["eos_bias"], lively=meta["lively"])
    kw = dict(meta["batch_kwargs"])
    kw["upper_range"], kw["lower_range"] = tuple(kw["upper_range"]), tuple(kw["lower_range"])
    batch = synthetic.make_batch(2, cfg, meta["batch_seed"], **kw)
    S = {k: v.to(dev) for k, v in st.items()}
    gt = [b.to(dev) for b in batch[1:7]]
    eng = engine.Engine(cfg)
    outs = eng.forward(S, batch[0].to(dev), inference=False, ground_truth=gt, teacher_forcing_ratio=1.0, training=True, dropout=False)
    losses, gouts = _loss_grads(outs, batch, dev)
    for i, (l, r) in enumerate(zip(losses, data["train_tf1.losses"])):
        _log(f"full loss term {i}: {l} vs {r} rel {abs(l - r) / abs(r):.3e}")
        assert abs(l - r) <= 1e-4 * abs(r), f"loss term {i}: {l} vs reference {r}"     # north-star: loss within 1e-4 rel
    G = engine_bwd.backward(eng, S, gouts)
    torch.cuda.synchronize()
    failures = []
    for k, rn in zip(meta["grad_names"], data["train_tf1.gradnorms"]):
        g = G[k]
        gn = float(g.double().norm())
        idx = torch.from_numpy(data[f"train_tf1.gsample_idx.{k}"]).to(dev)
        got = g.flatten()[idx].cpu().numpy()
        rs = data[f"train_tf1.gsample.{k}"]
        e_norm = abs(gn - rn) / max(rn, 1e-12)
        e_samp = np.abs(got - rs).max() / max(np.abs(rs).max(), rn / np.sqrt(g.numel()))
        _log(f"full {k}: norm {e_norm:.3e} samples {e_samp:.3e}")
        # Bars.  The ConvStack BN/conv gradients are sums over 2 x 1201 x 480 positions with heavy cancellation after ~2900 decoder steps
        # and 4 x 1201 GRU steps of back-propagation, and they are ILL-CONDITIONED at the fp32 level: perturbing the input spectrogram
        # by ONE unit in the last place moves these norms by up to 3.0e-4 from the fixture with the arithmetic otherwise unchanged
        # (tools/grad_conditioning.py, profiles/r01_grad_conditioning.txt: 1.6e-4 unperturbed; 3.0e-4 / 2.6e-4 / 1.1e-4 / 1.5e-4 for four
        # 1-ulp perturbations; 2.6e-4 with the split-operand convolutions, whose per-element error is at or below the fp32-input MFMA
        # kernel's).  Norm bar 5e-4 = 1.7x the worst 1-ulp deviation; everything outside the ConvStack is <= 1e-5.  Sampled elements 3e-3
        # of the tensor's max: two fp32 evaluations that only differ in summation grouping already disagree by 4.5e-4 there (oracle vs
        # reference, tests/test_oracle_full.py) and this path by up to 1.2e-3 on one element of bn4.bias.  The tight element-wise gate is
        # the reduced-size test above (all 83 tensors, 2e-4, measured 1.2e-5), where round-off does not accumulate over 1.15 M positions.
        if e_norm > (5e-4 if k.startswith("convstack.") else 2e-4) or e_samp > 3e-3:
            failures.append((k, e_norm, e_samp))
    assert not failures, f"{len(failures)} gradients off: {failures[:8]}"


def test_full_size_seeded_teacher_forcing(golden_dir, dev):
    """BASELINE.md section 3 parity gate, second case: the FULL model in train mode with SEEDED teacher forcing tf = 0.7 (hparams/pretrain.yaml's
    epoch-0 ratio) against the reference's fixture (tests/golden/make_golden.py g2tf): Python-random draw count, executed steps per (clip, bar),
    every fed-back token id, the four loss terms + total within 1e-4 rel (north-star bar), gradient norms."""
    import random
    from piano_a2s_amd import engine, engine_bwd, spec, synthetic
    data = np.load(os.path.join(golden_dir, "g2_full_tf07.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g2_full_tf07.json")))
    g2 = json.load(open(os.path.join(golden_dir, "g2_full.json")))
    assert min(meta["min_margin"].values()) >= 1e-3, "fixture precondition: no near-tie among the fed-back argmax decisions"
    cfg = spec.default_cfg()
    st = spec.procedural_state(cfg, meta["weights_seed"], eos_bias=meta["eos_bias"], lively=meta["lively"])
    kw = dict(g2["batch_kwargs"])
    kw["upper_range"], kw["lower_range"] = tuple(kw["upper_range"]), tuple(kw["lower_range"])
    batch = synthetic.make_batch(2, cfg, meta["batch_seed"], **kw)
    S = {k: v.to(dev) for k, v in st.items()}
    gt = [b.to(dev) for b in batch[1:7]]
    rng = random.Random(meta["random_seed"])
    draws = {"n": 0}

    class Counting:
        def random(self):
            draws["n"] += 1
            return rng.random()
    eng = engine.Engine(cfg)
    outs = eng.forward(S, batch[0].to(dev), inference=False, ground_truth=gt, teacher_forcing_ratio=meta["tf"], training=True, dropout=False, rng=Counting())
    torch.cuda.synchronize()
    assert draws["n"] == meta["draws"], "the forward must consume the reference's number of Python-random draws"
    up, lo = outs[2].cpu(), outs[3].cpu()
    assert np.array_equal((up.abs().sum(-1) > 0).sum(-1).numpy(), data["up_rows"]) and np.array_equal((lo.abs().sum(-1) > 0).sum(-1).numpy(), data["lo_rows"])
    assert np.array_equal(up.argmax(-1).numpy(), data["up_ids"]) and np.array_equal(lo.argmax(-1).numpy(), data["lo_ids"]), "token ids (incl. every fed-back argmax)"
    for nm, o in (("up", up), ("lo", lo)):
        err = np.abs(o.flatten()[torch.from_numpy(data[f"{nm}_sample_idx"])].numpy() - data[f"{nm}_sample"]).max()
        assert err <= 1e-4, f"{nm} log-probabilities differ by {err:.3e}"
    assert np.abs(outs[0].cpu().numpy() - data["ts"]).max() <= 1e-4 and np.abs(outs[1].cpu().numpy() - data["key"]).max() <= 1e-4
    losses, gouts = _loss_grads(outs, batch, dev)
    for i, (l, r) in enumerate(zip(losses, data["losses"])):
        _log(f"full tf0.7 loss term {i}: {l} vs {r} rel {abs(l - r) / abs(r):.3e}")
        assert abs(l - r) <= 1e-4 * abs(r), f"loss term {i}: {l} vs reference {r}"
    G = engine_bwd.backward(eng, S, gouts)
    torch.cuda.synchronize()
    failures = []
    for k, rn in zip(meta["grad_names"], data["gradnorms"]):
        e = abs(float(G[k].double().norm()) - rn) / max(rn, 1e-12)
        _log(f"full tf0.7 {k}: norm {e:.3e}")
        if e > (5e-4 if k.startswith("convstack.") else 2e-4):       # same bars as test_full_size_gradient_norms (see the note there)
            failures.append((k, e))
    assert not failures, f"{len(failures)} gradient norms off: {failures[:8]}"


@pytest.mark.parametrize("hidden", [64, 128, 96, 40, 300])
def test_other_hidden_sizes_forward_and_gradients(dev, hidden):
    """hidden_size 64 / 128 and (round 5) widths that are no power of two -- 96, 40 (not a multiple of 16 or 32: the generic step path) and 300 (> 256) -- the
    reference constructor takes any width, the one-workgroup-per-clip attention kernels take the width as a run-time argument:
    forward log-probabilities, loss and all 83 gradients against the oracle on CPU, train mode, teacher forcing 0.6 (seeded)."""
    import random
    from oracle import model_ref, recipe_ref
    from piano_a2s_amd import engine, engine_bwd, spec, synthetic
    cfg = spec.default_cfg(freq_bins=24, conv_feature_size=48, hidden_size=hidden, max_length=(12, 8))
    st = spec.procedural_state(cfg, 31, eos_bias=2.0, lively=True)
    batch = synthetic.make_batch(3, cfg, 6, frames=37, upper_range=(3, 10), lower_range=(2, 7), full_tail=0.1)
    P, Bf = spec.split_state(st)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = model_ref.forward(P, {k: v.clone() for k, v in Bf.items()}, cfg, batch[0], inference=False, ground_truth=[batch[i] for i in range(1, 7)],
                            teacher_forcing_ratio=0.6, training=True, rng=random.Random(5), dropout=False)
    rl = recipe_ref.objectives(ref, (batch[1], batch[2], batch[3], batch[5]))
    rl[0].backward()
    S = {k: v.to(dev) for k, v in st.items()}
    eng = engine.Engine(cfg)
    outs = eng.forward(S, batch[0].to(dev), inference=False, ground_truth=[b.to(dev) for b in batch[1:7]], teacher_forcing_ratio=0.6, training=True,
                       dropout=False, rng=random.Random(5))
    for name, o, r in zip(("ts", "key", "up", "lo"), outs, ref):
        assert float((o.cpu() - r.detach()).abs().max()) <= 1e-4, name
    losses, gouts = _loss_grads(outs, batch, dev)
    assert abs(losses[0] - float(rl[0])) <= 1e-4 * abs(float(rl[0]))
    G = engine_bwd.backward(eng, S, gouts)
    torch.cuda.synchronize()
    bad = []
    for k, p in P.items():
        rg = p.grad.double()
        err = float((G[k].cpu().double() - rg).abs().max()) / max(float(rg.abs().max()), 1e-12)
        if err > 2e-4:
            bad.append((k, err))
    assert not bad, bad[:6]
