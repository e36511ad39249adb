"""End-to-end forward parity of the HIP path (piano_a2s_amd.engine.Engine -> liba2s_hip.so) on the MI355X against
the reference's own outputs (golden fixtures made by tests/golden/make_golden.py).  Bars: token ids bit-exact;
fp32 log-probs within 1e-4 (north-star tolerance), relative to max(1, |ref|max)."""
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4
SMALL_BATCH = dict(frames=41, upper_range=(3, 10), lower_range=(2, 7), full_tail=0.1)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def g1(golden_dir):
    from piano_a2s_amd import spec, synthetic
    data = np.load(os.path.join(golden_dir, "g1_small.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g1_small.json")))
    cfg = spec.default_cfg(**meta["cfg"])
    batch = synthetic.make_batch(3, cfg, meta["batch_seed"], **SMALL_BATCH)
    return data, meta, cfg, batch


def _state(cfg, case, dev):
    from piano_a2s_amd import spec
    st = spec.procedural_state(cfg, case["weights_seed"], eos_bias=case["eos_bias"], lively=True)
    return {k: v.to(dev) for k, v in st.items()}


def _check(outs, data, prefix, tol=TOL):
    worst = 0.0
    for n, o in zip(("ts", "key", "up", "lo"), outs):
        ref = data[f"{prefix}.{n}"]
        err = float(np.abs(o.detach().cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
        worst = max(worst, err)
        assert err <= tol, f"{prefix}.{n}: {err:.3e} > {tol}"
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/op_errors.txt", "a") as f:
        f.write(f"forward {prefix}: {worst:.3e}\n")


@pytest.mark.parametrize("seed", [11, 18])
def test_small_greedy_ids_exact(g1, dev, seed):
    from piano_a2s_amd import engine
    data, meta, cfg, batch = g1
    S = _state(cfg, meta["cases"][f"greedy_s{seed}"], dev)
    outs = engine.Engine(cfg).forward(S, batch[0].to(dev), inference=True)
    torch.cuda.synchronize()
    _check(outs, data, f"greedy_s{seed}")
    for n, o in (("up", outs[2]), ("lo", outs[3])):
        ref = data[f"greedy_s{seed}.{n}"]
        assert np.array_equal(o.argmax(-1).cpu().numpy(), ref.argmax(-1)), f"{n} ids"
        # rows the reference never decoded stay exactly zero
        assert np.array_equal((o.abs().sum(-1) == 0).cpu().numpy(), np.abs(ref).sum(-1) == 0), f"{n} untouched rows"


def test_small_eval_teacher_forced(g1, dev):
    from piano_a2s_amd import engine
    data, meta, cfg, batch = g1
    S = _state(cfg, meta["cases"]["eval_tf1"], dev)
    gt = [b.to(dev) for b in batch[1:7]]
    outs = engine.Engine(cfg).forward(S, batch[0].to(dev), inference=False, ground_truth=gt, teacher_forcing_ratio=1.0, training=False)
    torch.cuda.synchronize()
    _check(outs, data, "eval_tf1")


@pytest.mark.parametrize("name", ["train_tf1", "train_tf05"])
def test_small_train_forward_and_buffers(g1, dev, name):
    from piano_a2s_amd import engine, spec
    data, meta, cfg, batch = g1
    case = meta["cases"][name]
    S = _state(cfg, case, dev)
    gt = [b.to(dev) for b in batch[1:7]]

    class CountingRandom(random.Random):
        n = 0

        def random(self):
            CountingRandom.n += 1
            return super().random()

    rng = CountingRandom()
    if case["random_seed"] is not None:
        rng.seed(case["random_seed"])
    outs = engine.Engine(cfg).forward(S, batch[0].to(dev), inference=False, ground_truth=gt, teacher_forcing_ratio=case["tf"],
                                      training=True, rng=rng, dropout=False)
    torch.cuda.synchronize()
    assert CountingRandom.n == case["draws"], f"python-random draws: {CountingRandom.n} vs reference {case['draws']}"
    _check(outs, data, name)
    for k in S:
        if spec.is_buffer(k):
            ref = data[f"{name}.buf.{k}"]
            err = float(np.abs(S[k].cpu().numpy().astype(np.float64) - ref).max()) / max(1.0, float(np.abs(ref).max()))
            assert err <= 1e-5, f"BN buffer {k}: {err:.3e}"


def test_full_size_greedy_ids_exact(golden_dir, dev, decoder_path):
    """16.36 M-parameter model, 1201 frames, 5 bars x (398 + 189) steps: reference ids must be reproduced exactly.
    Fair only because the fixture has no near-tie (asserted); on a mismatch the margin at the first differing
    decision is reported so a tie flip can be told from a real defect."""
    from piano_a2s_amd import engine, spec, synthetic
    data = np.load(os.path.join(golden_dir, "g2_full.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g2_full.json")))
    assert min(meta["min_margin"].values()) >= 1e-3, "fixture precondition: no near-tie argmax"
    cfg = spec.default_cfg()
    st = spec.procedural_state(cfg, meta["weights_seed"], eos_bias=meta["eos_bias"], lively=meta["lively"])
    kw = dict(meta["batch_kwargs"])
    kw["upper_range"], kw["lower_range"] = tuple(kw["upper_range"]), tuple(kw["lower_range"])
    batch = synthetic.make_batch(2, cfg, meta["batch_seed"], **kw)
    S = {k: v.to(dev) for k, v in st.items()}
    eng = engine.Engine(cfg)
    ts, key, up, lo = eng.forward(S, batch[0].to(dev), inference=True)
    torch.cuda.synchronize()
    decoder_path(eng)
    for nm, t in (("up", up), ("lo", lo)):
        ids = t.argmax(-1).cpu().numpy()
        ref = data[f"greedy.{nm}_ids"]
        if not np.array_equal(ids, ref):
            bad = np.argwhere(ids != ref)[0]
            margin = float(data[f"greedy.{nm}_margin"][tuple(bad)])
            raise AssertionError(f"{nm} ids differ first at (clip,bar,step)={tuple(bad)}: got {ids[tuple(bad)]} ref {ref[tuple(bad)]}; "
                                 f"reference top-2 margin there = {margin:.3e}; {int((ids != ref).sum())} of {ref.size} differ")
        rows = (t.abs().sum(-1) > 0).sum(-1).cpu().numpy()
        assert np.array_equal(rows, data[f"greedy.{nm}_rows"]), f"{nm} executed steps {rows.tolist()}"
        got = t.flatten()[torch.from_numpy(data[f"greedy.{nm}_sample_idx"]).to(t.device)].cpu().numpy()
        rs = data[f"greedy.{nm}_sample"]
        assert np.abs(got - rs).max() <= TOL * max(1.0, np.abs(rs).max()), f"{nm} log-probs {np.abs(got - rs).max():.3e}"
    assert np.abs(ts.cpu().numpy() - data["greedy.ts"]).max() <= TOL
    assert np.abs(key.cpu().numpy() - data["greedy.key"]).max() <= TOL


def test_greedy_graph_replay_matches_eager(g1, dev):
    """The hipGraph-replayed greedy decoder (opt-in) must give the same token ids, log-probs and executed-step counts as eager launches."""
    from piano_a2s_amd import engine
    data, meta, cfg, batch = g1
    S = _state(cfg, meta["cases"]["greedy_s11"], dev)
    outs = []
    for graph in (False, True):
        eng = engine.Engine(cfg)
        eng.greedy_graph = graph
        o = eng.forward(S, batch[0].to(dev), inference=True)
        torch.cuda.synchronize()
        outs.append(([t.cpu() for t in o], [b["staff"][k][2]["steps"] for b in eng.saved["bars"] for k in ("up", "lo")]))
    assert outs[0][1] == outs[1][1]
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a, b)
    _check(outs[1][0], data, "greedy_s11")
