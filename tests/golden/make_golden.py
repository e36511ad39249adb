#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE (build container only).

Imports /root/reference/models.py read-only (``music21`` is stubbed: humdrum.py imports it at module
top but the model only needs the pure-Python LabelsMultiple), drives it on CPU with this repo's
procedural weights (piano_a2s_amd.spec.procedural_state) and synthetic batches
(piano_a2s_amd.synthetic.make_batch), and stores inputs' checksums + the reference's outputs.
Nothing of the reference's text is stored: fixtures are numbers.

Usage:  python tests/golden/make_golden.py [g1] [g2] [g2tf] [g3] [g4] [g4b] [tok]
The fixtures are committed; this script documents how they were made and can regenerate them.
"""
import hashlib
import json
import os
import random
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import numpy as np
import torch

from piano_a2s_amd import spec, synthetic  # noqa: E402

SMALL = dict(freq_bins=24, conv_feature_size=32, hidden_size=32, max_length=(12, 8))
SMALL_BATCH = dict(frames=41, upper_range=(3, 10), lower_range=(2, 7), full_tail=0.1)


def load_reference():
    sys.modules.setdefault("music21", types.ModuleType("music21"))
    if "/root/reference" not in sys.path:
        sys.path.insert(1, "/root/reference")
    saved = sys.modules.pop("data_processing", None), sys.modules.pop("data_processing.humdrum", None)
    # make sure the REFERENCE's data_processing package is the one models.py sees
    sys.path.remove(REPO)
    try:
        import models as ref_models
        from data_processing.humdrum import LabelsMultiple as RefLabels
    finally:
        sys.path.insert(0, REPO)
    for k in [k for k in sys.modules if k == "data_processing" or k.startswith("data_processing.")]:
        del sys.modules[k]
    return ref_models, RefLabels


def digest(tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def no_dropout():
    """Neutralise dropout in THIS process only: models.py calls F.dropout(...)."""
    import torch.nn.functional as F
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x


def ref_losses(outs, batch):
    ts_o, key_o, up_o, lo_o = outs
    _, ts_t, key_t, up_t, _, lo_t, _, _, _ = batch
    nll, nll_pad = torch.nn.NLLLoss(), torch.nn.NLLLoss(ignore_index=147)
    # reference pretrain.py:72-88
    time_loss = nll(ts_o.permute(0, 2, 1), ts_t)
    key_loss = nll(key_o.permute(0, 2, 1), key_t)
    up = nll_pad(up_o.view(up_o.shape[0] * up_o.shape[1], -1, up_o.shape[3]).permute(0, 2, 1),
                 up_t.view(up_t.shape[0] * up_t.shape[1], -1))
    lo = nll_pad(lo_o.view(lo_o.shape[0] * lo_o.shape[1], -1, lo_o.shape[3]).permute(0, 2, 1),
                 lo_t.view(lo_t.shape[0] * lo_t.shape[1], -1))
    return time_loss + key_loss + up + lo, time_loss, key_loss, up, lo


def gt_of(batch):
    return [batch[1], batch[2], batch[3], batch[4], batch[5], batch[6]]


def make_g1(ref_models):
    cfg = spec.default_cfg(**SMALL)
    out = {}
    meta = {"cfg": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, "cases": {}}
    batch = synthetic.make_batch(3, cfg, 5, **SMALL_BATCH)
    meta["batch_seed"] = 5
    meta["batch_sha256"] = digest([batch[0], batch[1], batch[2], batch[3], batch[4], batch[5], batch[6]])

    def model_for(seed, eb):
        st = spec.procedural_state(cfg, seed, eos_bias=eb, lively=True)
        m = ref_models.ScoreTranscription(**SMALL)
        m.load_state_dict(st)
        return m, st

    # --- greedy, eval mode (two weight sets with mid-way early breaks and repeated EOS)
    for seed, eb in ((11, 3.0), (18, 3.0)):
        m, st = model_for(seed, eb)
        m.eval()
        with torch.no_grad():
            outs = m(spectrogram=batch[0], inference=True, ground_truth=None, teacher_forcing_ratio=0., device="cpu")
        name = f"greedy_s{seed}"
        for n, o in zip(("ts", "key", "up", "lo"), outs):
            out[f"{name}.{n}"] = o.numpy()
        meta["cases"][name] = {"weights_seed": seed, "eos_bias": eb, "state_sha256": digest(st.values())}

    # --- eval mode, teacher forced with ground truth (running-stat BN, tf=1)
    m, st = model_for(11, 3.0)
    m.eval()
    with torch.no_grad():
        outs = m(spectrogram=batch[0], inference=False, ground_truth=gt_of(batch), teacher_forcing_ratio=1.0, device="cpu")
    for n, o in zip(("ts", "key", "up", "lo"), outs):
        out[f"eval_tf1.{n}"] = o.numpy()
    meta["cases"]["eval_tf1"] = {"weights_seed": 11, "eos_bias": 3.0}

    # --- train mode (batch-stat BN), dropout neutralised, tf = 1: outputs, losses, all grads, BN buffers
    no_dropout()
    for name, tf, rseed in (("train_tf1", 1.0, None), ("train_tf05", 0.5, 7)):
        m, st = model_for(11, 3.0)
        m.train()
        if rseed is not None:
            random.seed(rseed)
        state0 = random.getstate()
        outs = m(spectrogram=batch[0], inference=False, ground_truth=gt_of(batch), teacher_forcing_ratio=tf, device="cpu")
        # count python-random draws the forward consumed (a-12): replay from state0 until states match
        state1 = random.getstate()
        random.setstate(state0)
        draws = 0
        while random.getstate() != state1 and draws < 100000:
            random.random()
            draws += 1
        losses = ref_losses(outs, batch)
        losses[0].backward()
        for n, o in zip(("ts", "key", "up", "lo"), outs):
            out[f"{name}.{n}"] = o.detach().numpy()
        out[f"{name}.losses"] = np.array([float(l) for l in losses], dtype=np.float64)
        sd = m.state_dict()
        for k, p in m.named_parameters():
            if name == "train_tf1":
                out[f"{name}.grad.{k}"] = p.grad.numpy().copy()
            else:
                out[f"{name}.gradnorm.{k}"] = np.array(float(p.grad.double().norm()))
        for k in sd:
            if spec.is_buffer(k):
                out[f"{name}.buf.{k}"] = sd[k].numpy().copy()
        meta["cases"][name] = {"weights_seed": 11, "eos_bias": 3.0, "tf": tf, "random_seed": rseed, "draws": draws}

        if name == "train_tf1":
            # --- one clip + Adadelta step with the very torch objects the reference recipe instantiates
            params = [p for p in m.parameters()]
            total = torch.nn.utils.clip_grad_norm_(params, 5.0)
            opt = torch.optim.Adadelta(params, lr=1.0, rho=0.95, eps=1e-8)
            opt.step()
            out["step.total_norm"] = np.array(float(total))
            for k, p in m.named_parameters():
                out[f"step.param.{k}"] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g1_small.npz"), **out)
    with open(os.path.join(HERE, "g1_small.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("g1 written:", len(out), "arrays")


FULL_BATCH = dict(frames=1201, upper_range=(6, 24), lower_range=(4, 16), full_tail=0.0, spectrogram="ridges")


def make_g2(ref_models, seed=2032, eb=None):
    """Full-size model (16.36 M parameters, hparams/pretrain.yaml dims), procedural weights, B=2.

    (seed, eos_bias) were chosen by scanning a handful of seeds for a decode that (i) breaks early at
    data-dependent steps in some bars and runs to the 398/189 caps in others, (ii) differs between the two
    clips, and (iii) keeps every argmax decision at least 1e-3 away from a tie -- three orders of magnitude
    above fp32 re-association noise -- so that demanding bit-exact token ids from an implementation that
    sums in a different order is a fair test.  The margins are stored in the fixture and the parity test
    asserts that precondition instead of assuming it."""
    cfg = spec.default_cfg()
    eb = 2.5 if eb is None else eb
    st = spec.procedural_state(cfg, seed, eos_bias=eb, lively="token")
    batch = synthetic.make_batch(2, cfg, 77, **FULL_BATCH)
    out, meta = {}, {"weights_seed": seed, "eos_bias": eb, "batch_seed": 77, "batch_kwargs": {k: list(v) if isinstance(v, tuple) else v for k, v in FULL_BATCH.items()},
                     "state_sha256": digest(st.values()),
                     "batch_sha256": digest([batch[0], batch[1], batch[2], batch[3], batch[4], batch[5], batch[6]])}
    m = ref_models.ScoreTranscription(**cfg)      # NB the class defaults (437,129) differ from the yaml's (398,189)
    m.load_state_dict(st)
    m.eval()
    with torch.no_grad():
        ts, key, up, lo = m(spectrogram=batch[0], inference=True, ground_truth=None, teacher_forcing_ratio=0., device="cpu")
    out["greedy.ts"], out["greedy.key"] = ts.numpy(), key.numpy()
    out["greedy.up_ids"] = up.argmax(-1).numpy().astype(np.int16)
    out["greedy.lo_ids"] = lo.argmax(-1).numpy().astype(np.int16)
    out["greedy.up_rows"] = (up.abs().sum(-1) > 0).sum(-1).numpy().astype(np.int16)   # executed steps per (b,bar)
    out["greedy.lo_rows"] = (lo.abs().sum(-1) > 0).sum(-1).numpy().astype(np.int16)
    g = np.random.default_rng(0)
    idx = g.integers(0, up.numel(), size=2000)
    out["greedy.up_sample_idx"], out["greedy.up_sample"] = idx, up.flatten()[idx].numpy()
    idx = g.integers(0, lo.numel(), size=2000)
    out["greedy.lo_sample_idx"], out["greedy.lo_sample"] = idx, lo.flatten()[idx].numpy()
    # top-2 margin of every decoded row: how far the argmax is from flipping
    top2 = up.topk(2, dim=-1).values
    out["greedy.up_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
    top2 = lo.topk(2, dim=-1).values
    out["greedy.lo_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
    for nm, x in (("ts", ts), ("key", key)):
        top2 = x.topk(2, dim=-1).values
        out[f"greedy.{nm}_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
    decoded_up, decoded_lo = up.abs().sum(-1) > 0, lo.abs().sum(-1) > 0
    meta["min_margin"] = {"up": float(torch.from_numpy(out["greedy.up_margin"])[decoded_up].min()),
                          "lo": float(torch.from_numpy(out["greedy.lo_margin"])[decoded_lo].min()),
                          "ts": float(out["greedy.ts_margin"].min()), "key": float(out["greedy.key_margin"].min())}
    meta["lively"] = "token"
    print("g2 greedy rows", out["greedy.up_rows"].tolist(), out["greedy.lo_rows"].tolist())

    no_dropout()
    m.train()
    outs = m(spectrogram=batch[0], inference=False, ground_truth=gt_of(batch), teacher_forcing_ratio=1.0, device="cpu")
    losses = ref_losses(outs, batch)
    losses[0].backward()
    out["train_tf1.losses"] = np.array([float(l) for l in losses], dtype=np.float64)
    out["train_tf1.ts"], out["train_tf1.key"] = outs[0].detach().numpy(), outs[1].detach().numpy()
    idx = g.integers(0, outs[2].numel(), size=2000)
    out["train_tf1.up_sample_idx"], out["train_tf1.up_sample"] = idx, outs[2].detach().flatten()[idx].numpy()
    names = []
    norms = []
    for k, p in m.named_parameters():
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        flat = p.grad.flatten()
        sidx = g.integers(0, flat.numel(), size=min(64, flat.numel()))
        out[f"train_tf1.gsample_idx.{k}"] = sidx
        out[f"train_tf1.gsample.{k}"] = flat[sidx].numpy()
    out["train_tf1.gradnorms"] = np.array(norms)
    meta["grad_names"] = names
    sd = m.state_dict()
    for k in sd:
        if spec.is_buffer(k):
            out[f"train_tf1.buf.{k}"] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g2_full.npz"), **out)
    with open(os.path.join(HERE, "g2_full.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("g2 written")


def make_g2_tf(ref_models, seed=2032, eb=2.5, tf=0.7, rseeds=(3, 5, 9, 12)):
    """Full-size model, train mode (batch-statistics BatchNorm, dropout neutralised), SEEDED teacher forcing tf = 0.7 -- the epoch-0
    ratio of hparams/pretrain.yaml and the second case of BASELINE.md section 3's parity gate.  Same weights and batch as g2.  With
    0 < tf < 1 the steps whose coin says "no" feed the model's OWN argmax back, so a near-tie in any such decision would let two correct
    fp32 implementations diverge; the Python-random seed is picked among `rseeds` as the one whose decisions have the largest minimum
    top-2 margin, the margin is stored, and the parity test asserts that precondition (as g2 does for greedy decoding)."""
    cfg = spec.default_cfg()
    st = spec.procedural_state(cfg, seed, eos_bias=eb, lively="token")
    batch = synthetic.make_batch(2, cfg, 77, **FULL_BATCH)
    no_dropout()
    best = None
    for rseed in rseeds:
        m = ref_models.ScoreTranscription(**cfg)
        m.load_state_dict(st)
        m.train()
        random.seed(rseed)
        state0 = random.getstate()
        outs = m(spectrogram=batch[0], inference=False, ground_truth=gt_of(batch), teacher_forcing_ratio=tf, device="cpu")
        state1 = random.getstate()
        random.setstate(state0)
        draws = 0
        while random.getstate() != state1 and draws < 100000:
            random.random()
            draws += 1
        margins = []
        for o in outs[2:]:
            decoded = o.detach().abs().sum(-1) > 0
            top2 = o.detach().topk(2, dim=-1).values
            margins.append(float((top2[..., 0] - top2[..., 1])[decoded].min()))
        for o in outs[:2]:
            top2 = o.detach().topk(2, dim=-1).values
            margins.append(float((top2[..., 0] - top2[..., 1]).min()))
        print("g2_tf: python-random seed", rseed, "draws", draws, "min margins (up, lo, ts, key)", margins, flush=True)
        if best is None or min(margins) > min(best[3]):
            best = (rseed, draws, outs, margins, m)
    rseed, draws, outs, margins, m = best
    losses = ref_losses(outs, batch)
    losses[0].backward()
    g = np.random.default_rng(1)
    out = {"losses": np.array([float(l) for l in losses], dtype=np.float64), "ts": outs[0].detach().numpy(), "key": outs[1].detach().numpy(),
           "up_rows": (outs[2].detach().abs().sum(-1) > 0).sum(-1).numpy().astype(np.int16),
           "lo_rows": (outs[3].detach().abs().sum(-1) > 0).sum(-1).numpy().astype(np.int16),
           "up_ids": outs[2].detach().argmax(-1).numpy().astype(np.int16), "lo_ids": outs[3].detach().argmax(-1).numpy().astype(np.int16)}
    for nm, o in (("up", outs[2]), ("lo", outs[3])):
        idx = g.integers(0, o.numel(), size=2000)
        out[f"{nm}_sample_idx"], out[f"{nm}_sample"] = idx, o.detach().flatten()[idx].numpy()
    names, norms = [], []
    for k, p in m.named_parameters():
        names.append(k)
        norms.append(float(p.grad.double().norm()))
    out["gradnorms"] = np.array(norms)
    meta = {"weights_seed": seed, "eos_bias": eb, "lively": "token", "batch_seed": 77, "tf": tf, "random_seed": rseed, "draws": draws,
            "min_margin": dict(zip(("up", "lo", "ts", "key"), margins)), "grad_names": names, "state_sha256": digest(st.values()),
            "batch_sha256": digest([batch[0], batch[1], batch[2], batch[3], batch[4], batch[5], batch[6]])}
    np.savez_compressed(os.path.join(HERE, "g2_full_tf07.npz"), **out)
    with open(os.path.join(HERE, "g2_full_tf07.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("g2_tf written: seed", rseed, "losses", out["losses"].tolist())


def margin_summary(o):
    """Top-2 margins of the decoded rows of a staff output: histogram over decades + the smallest ones (how far every argmax is from a tie)."""
    o = o.detach()
    decoded = o.abs().sum(-1) > 0
    top2 = o.topk(2, dim=-1).values
    mg = (top2[..., 0] - top2[..., 1])
    vals = mg[decoded].double().numpy()
    edges = [0.0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0, float("inf")]
    hist = np.histogram(vals, bins=edges)[0].tolist() if vals.size else [0] * 8
    return mg.numpy().astype(np.float32), {"decisions": int(vals.size), "hist_edges": edges[:-1] + ["inf"], "hist": hist,
                                           "below_1e-3": int((vals < 1e-3).sum()), "min": float(vals.min()) if vals.size else None}


G3_BATCH = dict(frames=1201, upper_range=(6, 24), lower_range=(4, 16), full_tail=0.0, spectrogram="ridges")
G3_CASES = {
    # name: (weights seed, eos bias, batch size, batch seed, rows forced to full length, mode, tf, python-random seed)
    "s2041_greedy": (2041, 2.5, 2, 78, (), "greedy", 0.0, None),
    "s2057_greedy": (2057, 2.5, 2, 79, (), "greedy", 0.0, None),
    "b4_greedy": (2032, 2.5, 4, 80, (), "greedy", 0.0, None),
    "tail_tf1": (2041, 2.5, 2, 81, ((0, 1, "up"), (1, 3, "lo")), "train", 1.0, None),        # full-length bars WITHOUT <eos>, teacher forced
    "tf06": (2057, 2.5, 2, 78, (), "train", 0.6, 4),                                       # reference finetune.py:44 ratio
}


def make_g3(ref_models, only=None):
    """Round 4: more weight seeds, a B = 4 batch, a train-mode case with full-length bars that never show <eos>, and the
    finetune teacher-forcing ratio.  Seeds are NOT screened for margins: the margin histogram of every case is stored and the parity test
    reports how many decisions lie below 1e-3 (two correct fp32 implementations may legitimately part ways at such a decision)."""
    cfg = spec.default_cfg()
    meta_path = os.path.join(HERE, "g3_full.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {"batch_kwargs": {k: list(v) if isinstance(v, tuple) else v for k, v in G3_BATCH.items()},
                                                                         "cases": {}}
    no_dropout()
    for name, (wseed, eb, B, bseed, full_rows, mode, tf, rseed) in G3_CASES.items():
        if only and name not in only:
            continue
        st = spec.procedural_state(cfg, wseed, eos_bias=eb, lively="token")
        batch = synthetic.make_batch(B, cfg, bseed, full_rows=full_rows, **G3_BATCH)
        m = ref_models.ScoreTranscription(**cfg)
        m.load_state_dict(st)
        out = {}
        cm = {"weights_seed": wseed, "eos_bias": eb, "lively": "token", "batch": B, "batch_seed": bseed, "full_rows": [list(r) for r in full_rows],
              "mode": mode, "tf": tf, "random_seed": rseed, "state_sha256": digest(st.values()),
              "batch_sha256": digest([batch[0], batch[1], batch[2], batch[3], batch[4], batch[5], batch[6]])}
        g = np.random.default_rng(2)
        if mode == "greedy":
            m.eval()
            with torch.no_grad():
                outs = m(spectrogram=batch[0], inference=True, ground_truth=None, teacher_forcing_ratio=0., device="cpu")
        else:
            m.train()
            if rseed is not None:
                random.seed(rseed)
            state0 = random.getstate()
            outs = m(spectrogram=batch[0], inference=False, ground_truth=gt_of(batch), teacher_forcing_ratio=tf, device="cpu")
            state1 = random.getstate()
            random.setstate(state0)
            draws = 0
            while random.getstate() != state1 and draws < 100000:
                random.random()
                draws += 1
            cm["draws"] = draws
            losses = ref_losses(outs, batch)
            losses[0].backward()
            out["losses"] = np.array([float(l) for l in losses], dtype=np.float64)
            names, norms = [], []
            for k, p in m.named_parameters():
                names.append(k)
                norms.append(float(p.grad.double().norm()))
            out["gradnorms"] = np.array(norms)
            cm["grad_names"] = names
        ts, key, up, lo = [o.detach() for o in outs]
        out["ts"], out["key"] = ts.numpy(), key.numpy()
        cm["margins"] = {}
        for nm, o in (("up", up), ("lo", lo)):
            out[f"{nm}_ids"] = o.argmax(-1).numpy().astype(np.int16)
            out[f"{nm}_rows"] = (o.abs().sum(-1) > 0).sum(-1).numpy().astype(np.int16)
            idx = g.integers(0, o.numel(), size=2000)
            out[f"{nm}_sample_idx"], out[f"{nm}_sample"] = idx, o.flatten()[idx].numpy()
            out[f"{nm}_margin"], cm["margins"][nm] = margin_summary(o)
        print("g3", name, "rows", out["up_rows"].tolist(), out["lo_rows"].tolist(), "margins", {k: (v["below_1e-3"], v["min"]) for k, v in cm["margins"].items()},
              "losses", out.get("losses", np.zeros(0)).tolist(), flush=True)
        np.savez_compressed(os.path.join(HERE, f"g3_{name}.npz"), **out)
        meta["cases"][name] = cm
        with open(meta_path, "w") as f:
            json.dump(meta, f, indent=1)
        del m, outs
    print("g3 written")


G4_BATCH = dict(frames=301, upper_range=(10, 60), lower_range=(6, 40), full_tail=0.0, spectrogram="ridges")
G4 = dict(weights_seed=2041, eos_bias=2.5, batch=12, batch_seed=90, full_rows=((3, 1, "up"), (8, 3, "lo")), tf=0.7, rseeds=(4, 6, 11, 15, 21, 28))


G4B_BATCH = dict(frames=301, upper_range=(10, 60), lower_range=(6, 40), full_tail=0.0, spectrogram="ridges")
# round 6: TWO clips (2 and 9) with a full-length UPPER bar, in bars 0 and 3 -- with the planner's 12-clip cost setting (step_cost 4) they
# form the long-clip group, and under a seed whose bar-level coins put bars 0 and 3 into different bar segments the fused step cuts that group in two
# (train.split_long_group): the three-clip-group path the benchmark runs by default.  Seeds under which the host plan does not split are not candidates.
G4B = dict(weights_seed=2041, eos_bias=2.5, batch=12, batch_seed=93, full_rows=((2, 0, "up"), (9, 3, "up")), tf=0.7, rseeds=(2, 4, 15, 21, 28, 33),
           plan_kw={"step_cost": 4.0}, must_split=True)


def host_groups(batch, cfg, rseed, tf, plan_kw):
    """The clip groups train.TrainStep forms on this minibatch under Python-random seed `rseed` (pure host code: piano_a2s_amd.train.plan_step_groups)."""
    from piano_a2s_amd import train
    order, cuts, _ = train.plan_step_groups([batch[3], batch[5], batch[4], batch[6]], cfg["max_bars"], cfg["max_length"], random.Random(rseed), tf, plan_kw, True)
    return order.tolist(), cuts


def make_g4(ref_models, G4=None, G4_BATCH=None, name="g4_step"):
    """Round 5: ONE WHOLE OPTIMIZER STEP of the reference on a minibatch that forces the fused step's planner through its
    control flow -- full widths (H = 256, E = 16, 480 bins), T = 301 frames (so that the as-written reference fits in this container's RAM at
    B = 12), two clips (3 and 8, neither at the end: the planner must permute) holding a full-length bar without <eos>, train mode, dropout
    neutralised, seeded teacher forcing 0.7.  Stored: draw count, executed steps, token ids + top-2 margins, sampled log-probabilities, the loss
    terms, every gradient norm, the clip norm, and -- from the torch objects the reference recipe instantiates (clip_grad_norm_(5.0),
    Adadelta(lr=1, rho=.95, eps=1e-8); reference pretrain.py:121-129, hparams/pretrain.yaml:44-47) -- per parameter the norm of the update, the norm of
    the updated tensor and 64 sampled values of it.  The Python-random seed is the one of `rseeds` with the largest minimum margin (as g2_tf)."""
    G4 = G4 or globals()["G4"]
    G4_BATCH = G4_BATCH or globals()["G4_BATCH"]
    cfg = spec.default_cfg()
    st = spec.procedural_state(cfg, G4["weights_seed"], eos_bias=G4["eos_bias"], lively="token")
    batch = synthetic.make_batch(G4["batch"], cfg, G4["batch_seed"], full_rows=G4["full_rows"], **G4_BATCH)
    no_dropout()
    rseeds = list(G4["rseeds"])
    if G4.get("must_split"):
        plans = {r: host_groups(batch, cfg, r, G4["tf"], G4["plan_kw"]) for r in rseeds}
        for r, (order, cuts) in plans.items():
            print(name, ": seed", r, "-> clip groups", cuts, "order", order, flush=True)
        rseeds = [r for r in rseeds if len(plans[r][1]) == 3]
        assert rseeds, "no candidate seed cuts the long-clip group in two"
    live = [(batch[3] != 147), (batch[5] != 147)]          # decisions that reach the loss or the next bar's staff token: targets that are not <pad>

    def run(rseed, grad):
        m = ref_models.ScoreTranscription(**cfg)
        m.load_state_dict(st)
        m.train()
        random.seed(rseed)
        state0 = random.getstate()
        with torch.enable_grad() if grad else torch.no_grad():
            outs = m(spectrogram=batch[0], inference=False, ground_truth=gt_of(batch), teacher_forcing_ratio=G4["tf"], device="cpu")
        state1 = random.getstate()
        random.setstate(state0)
        draws = 0
        while random.getstate() != state1 and draws < 100000:
            random.random()
            draws += 1
        mins = []
        for o, lv in zip(outs[2:], live):
            top2 = o.detach().topk(2, dim=-1).values
            mins.append(float((top2[..., 0] - top2[..., 1])[lv].min()))
        return m, outs, draws, mins

    best = None
    for rseed in rseeds:
        m, outs, draws, mins = run(rseed, False)
        print("g4: python-random seed", rseed, "draws", draws, "min margins at non-pad targets (up, lo)", mins, flush=True)
        if best is None or min(mins) > min(best[1]):
            best = (rseed, mins)
        del m, outs
    rseed = best[0]
    m, outs, draws, mins = run(rseed, True)
    cm_live = dict(zip(("up", "lo"), mins))
    losses = ref_losses(outs, batch)
    losses[0].backward()
    g = np.random.default_rng(4)
    out = {"losses": np.array([float(l) for l in losses], dtype=np.float64)}
    cm = {"weights_seed": G4["weights_seed"], "eos_bias": G4["eos_bias"], "lively": "token", "batch": G4["batch"], "batch_seed": G4["batch_seed"],
          "full_rows": [list(r) for r in G4["full_rows"]], "tf": G4["tf"], "random_seed": rseed, "draws": draws, "margins": {}, "min_margin_at_targets": cm_live,
          "batch_kwargs": {k: list(v) if isinstance(v, tuple) else v for k, v in G4_BATCH.items()}, "state_sha256": digest(st.values()),
          "batch_sha256": digest([batch[0], batch[1], batch[2], batch[3], batch[4], batch[5], batch[6]])}
    if G4.get("must_split"):
        order, cuts = host_groups(batch, cfg, rseed, G4["tf"], G4["plan_kw"])
        assert len(cuts) == 3
        cm["plan_kw"], cm["clip_groups"], cm["clip_order"] = G4["plan_kw"], [list(c) for c in cuts], order
    ts, key, up, lo = [o.detach() for o in outs]
    out["ts"], out["key"] = ts.numpy(), key.numpy()
    for nm, o in (("up", up), ("lo", lo)):
        out[f"{nm}_ids"] = o.argmax(-1).numpy().astype(np.int16)
        out[f"{nm}_rows"] = (o.abs().sum(-1) > 0).sum(-1).numpy().astype(np.int16)
        idx = g.integers(0, o.numel(), size=4000)
        out[f"{nm}_sample_idx"], out[f"{nm}_sample"] = idx, o.flatten()[idx].numpy()
        out[f"{nm}_margin"], cm["margins"][nm] = margin_summary(o)
    names, norms = [], []
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    for k, p in m.named_parameters():
        names.append(k)
        norms.append(float(p.grad.double().norm()))
    out["gradnorms"] = np.array(norms)
    cm["grad_names"] = names
    params = [p for p in m.parameters()]
    total = torch.nn.utils.clip_grad_norm_(params, 5.0)
    opt = torch.optim.Adadelta(params, lr=1.0, rho=0.95, eps=1e-8)
    opt.step()
    out["step_total_norm"] = np.array(float(total))
    upd, after = [], []
    for k, p in m.named_parameters():
        d = p.detach()
        upd.append(float((d - before[k]).double().norm()))
        after.append(float(d.double().norm()))
        flat = d.flatten()
        sidx = g.integers(0, flat.numel(), size=min(64, flat.numel()))
        out[f"step_sample_idx.{k}"], out[f"step_sample.{k}"] = sidx, flat[sidx].numpy()
    out["step_update_norms"], out["step_param_norms"] = np.array(upd), np.array(after)
    sd = m.state_dict()
    for k in sd:
        if spec.is_buffer(k):
            out[f"buf.{k}"] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(cm, f, indent=1)
    print(name, "written: seed", rseed, "rows", out["up_rows"].tolist(), out["lo_rows"].tolist(), "losses", out["losses"].tolist(), "clip norm", float(total))


def make_tok(RefLabels):
    lab = RefLabels(extended=True)
    cases = ["4c", "4c\t8e 8g\n4r", "[2.CC#_ 4ee-;]\t.\n16ffff", "8.r\t4c 4e 4g", "16.BBB#]\t[8cccc-",
             "2r\t.\n.\t4GG# 4d-\n=\t=", "128CCC\t176BBB-_ 112ee#;", "4c 4e 4g\t4C 4E 4G\n.\t8r\n*\t*"]
    kats = []
    for s in cases:
        try:
            ids = lab.encode(s)
            kats.append({"text": s, "ids": ids, "decoded": lab.decode(ids)})
        except Exception as e:  # noqa: BLE001
            kats.append({"text": s, "raises": type(e).__name__})
    for s in ["4h", "4c  4e", ""]:
        try:
            lab.encode(s)
            kats.append({"text": s, "ids": lab.encode(s)})
        except Exception as e:  # noqa: BLE001
            kats.append({"text": s, "raises": type(e).__name__})
    data = {"labels_extended": lab.labels, "n_base": len(RefLabels(extended=False).labels), "kats": kats}
    with open(os.path.join(HERE, "tokenizer_kat.json"), "w") as f:
        json.dump(data, f, indent=1)
    print("tokenizer KATs written:", len(kats))


if __name__ == "__main__":
    what = sys.argv[1:] or ["tok", "g1", "g2"]
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, RefLabels = load_reference()
    if "tok" in what:
        make_tok(RefLabels)
    if "g1" in what:
        make_g1(ref_models)
    if "g2" in what:
        make_g2(ref_models)
    if "g2tf" in what:
        make_g2_tf(ref_models)
    if "g3" in what or any(w.startswith("g3:") for w in what):
        only = [w[3:] for w in what if w.startswith("g3:")]
        make_g3(ref_models, only or None)
    if "g4" in what:
        make_g4(ref_models)
    if "g4b" in what:
        make_g4(ref_models, G4B, G4B_BATCH, "g4b_step")
