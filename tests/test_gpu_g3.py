"""Full-size parity against the reference on UNSCREENED cases (tests/golden/make_golden.py g3): two more weight
seeds, a B = 4 batch, a train-mode batch with full-length bars that never show <eos> (398 / 189 teacher-forced steps), and the
finetune teacher-forcing ratio 0.6 (reference finetune.py:44).

The seeds of these fixtures were NOT chosen for wide top-2 margins.  Policy for token ids: they must equal the reference's everywhere,
EXCEPT that a clip may part ways at a decision whose reference margin is below 1e-3 (two correct fp32 implementations that sum in a
different order can legitimately flip such an argmax; in free-running decoding everything that clip decodes afterwards then differs).
Every such event is written to gpurun_out/g3_parity_report.txt together with the margin histogram of the case -- reported, not hidden.
"""
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4
NEAR_TIE = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def g3(golden_dir):
    return json.load(open(os.path.join(golden_dir, "g3_full.json")))


def _report(line):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/g3_parity_report.txt", "a") as f:
        f.write(line + "\n")


def _setup(g3, golden_dir, name, dev):
    from piano_a2s_amd import spec, synthetic
    case = g3["cases"][name]
    data = np.load(os.path.join(golden_dir, f"g3_{name}.npz"))
    cfg = spec.default_cfg()
    st = spec.procedural_state(cfg, case["weights_seed"], eos_bias=case["eos_bias"], lively=case["lively"])
    kw = dict(g3["batch_kwargs"])
    kw["upper_range"], kw["lower_range"] = tuple(kw["upper_range"]), tuple(kw["lower_range"])
    batch = synthetic.make_batch(case["batch"], cfg, case["batch_seed"], full_rows=[tuple(r) for r in case["full_rows"]], **kw)
    S = {k: v.to(dev) for k, v in st.items()}
    return case, data, cfg, batch, S


def _compare_ids(name, case, data, up, lo, free_running):
    """Returns the set of clips that left the reference's path at a near-tie (their later outputs are not comparable)."""
    diverged = {}
    ids = {"up": up.argmax(-1).cpu().numpy(), "lo": lo.argmax(-1).cpu().numpy()}
    B, bars = ids["up"].shape[:2]
    flips = 0
    for b in range(B):
        for bar in range(bars):
            if b in diverged and free_running:
                break
            for nm in ("up", "lo"):
                ref, got = data[f"{nm}_ids"][b, bar], ids[nm][b, bar]
                if np.array_equal(ref, got):
                    continue
                t = int(np.argwhere(ref != got)[0][0])
                margin = float(data[f"{nm}_margin"][b, bar, t])
                msg = f"{name}: {nm} ids of clip {b} bar {bar} differ first at step {t}: got {got[t]} ref {ref[t]}, reference top-2 margin {margin:.3e}"
                assert margin < NEAR_TIE, msg + " -- not a near-tie: a real defect"
                _report(msg + " (near-tie flip, tolerated)")
                flips += 1
                if free_running:
                    diverged[b] = (bar, nm, t)
    for nm in ("up", "lo"):
        m = case["margins"][nm]
        _report(f"{name}: {nm} decisions {m['decisions']}, below 1e-3: {m['below_1e-3']}, min margin {m['min']:.3e}, histogram over decades {m['hist']}")
    _report(f"{name}: near-tie flips on the MI355X: {flips}; clips that left the reference's path: {sorted(diverged)}")
    return diverged


@pytest.mark.parametrize("name", ["s2041_greedy", "s2057_greedy", "b4_greedy"])
def test_g3_greedy(g3, golden_dir, dev, name, decoder_path):
    from piano_a2s_amd import engine
    case, data, cfg, batch, S = _setup(g3, golden_dir, name, dev)
    eng = engine.Engine(cfg)
    ts, key, up, lo = eng.forward(S, batch[0].to(dev), inference=True)
    torch.cuda.synchronize()
    decoder_path(eng)
    name = f"{name}[{decoder_path.name}]"
    diverged = _compare_ids(name, case, data, up, lo, free_running=True)
    if not diverged:
        for nm, t in (("up", up), ("lo", lo)):
            rows = (t.abs().sum(-1) > 0).sum(-1).cpu().numpy()
            assert np.array_equal(rows, data[f"{nm}_rows"]), f"{nm} executed steps {rows.tolist()}"
            got = t.flatten()[torch.from_numpy(data[f"{nm}_sample_idx"]).to(t.device)].cpu().numpy()
            rs = data[f"{nm}_sample"]
            assert np.abs(got - rs).max() <= TOL * max(1.0, np.abs(rs).max()), f"{nm} log-probs {np.abs(got - rs).max():.3e}"
        assert np.abs(ts.cpu().numpy() - data["ts"]).max() <= TOL
        assert np.abs(key.cpu().numpy() - data["key"]).max() <= TOL


@pytest.mark.parametrize("name", ["tail_tf1", "tf06"])
def test_g3_train_mode(g3, golden_dir, dev, name, decoder_path):
    """Train mode (batch-statistics BatchNorm, dropout neutralised as in the fixture): draw count, executed steps, ids, log-probabilities,
    the four loss terms + total within 1e-4 relative, all gradient norms.  tail_tf1 holds a 398-step upper bar and a 189-step lower bar
    without <eos> -- the rows the benchmark's 1 % tail consists of."""
    from piano_a2s_amd import engine, engine_bwd
    from tests.test_gpu_backward import _loss_grads
    case, data, cfg, batch, S = _setup(g3, golden_dir, name, dev)
    rng = random.Random(case["random_seed"]) if case["random_seed"] is not None else random.Random(0)
    draws = {"n": 0}

    class Counting:
        def random(self):
            draws["n"] += 1
            return rng.random()
    eng = engine.Engine(cfg)
    gt = [b.to(dev) for b in batch[1:7]]
    outs = eng.forward(S, batch[0].to(dev), inference=False, ground_truth=gt, teacher_forcing_ratio=case["tf"], training=True, dropout=False, rng=Counting())
    torch.cuda.synchronize()
    decoder_path(eng)
    assert draws["n"] == case["draws"], f"python-random draws {draws['n']} vs reference {case['draws']}"
    up, lo = outs[2], outs[3]
    for nm, t in (("up", up), ("lo", lo)):
        rows = (t.abs().sum(-1) > 0).sum(-1).cpu().numpy()
        assert np.array_equal(rows, data[f"{nm}_rows"]), f"{nm} executed steps {rows.tolist()} vs {data[f'{nm}_rows'].tolist()}"
    full_length_case = name == "tail_tf1"
    name = f"{name}[{decoder_path.name}]"
    if full_length_case:
        assert int(data["up_rows"].max()) == cfg["max_length"][0] and int(data["lo_rows"].max()) == cfg["max_length"][1], "fixture: full-length bars"
    diverged = _compare_ids(name, case, data, up, lo, free_running=case["tf"] < 1.0)
    if diverged:
        pytest.skip(f"{name}: a fed-back near-tie decision flipped (reported in gpurun_out/g3_parity_report.txt); losses are not comparable")
    for nm, o in (("up", up), ("lo", lo)):
        got = o.flatten()[torch.from_numpy(data[f"{nm}_sample_idx"]).to(o.device)].cpu().numpy()
        err = np.abs(got - data[f"{nm}_sample"]).max()
        assert err <= TOL, f"{nm} log-probabilities differ by {err:.3e}"
    assert np.abs(outs[0].cpu().numpy() - data["ts"]).max() <= TOL and np.abs(outs[1].cpu().numpy() - data["key"]).max() <= TOL
    losses, gouts = _loss_grads(outs, batch, dev)
    for i, (l, r) in enumerate(zip(losses, data["losses"])):
        _report(f"{name}: loss term {i}: {l} vs reference {r}, rel {abs(l - r) / abs(r):.3e}")
        assert abs(l - r) <= 1e-4 * abs(r), f"loss term {i}: {l} vs reference {r}"
    G = engine_bwd.backward(eng, S, gouts)
    torch.cuda.synchronize()
    failures, worst = [], 0.0
    for k, rn in zip(case["grad_names"], data["gradnorms"]):
        e = abs(float(G[k].double().norm()) - rn) / max(rn, 1e-12)
        worst = max(worst, e)
        if e > (5e-4 if k.startswith("convstack.") else 2e-4):       # same bars as test_full_size_gradient_norms (see the note there)
            failures.append((k, e))
    _report(f"{name}: worst gradient-norm error {worst:.3e} over {len(case['grad_names'])} parameters")
    assert not failures, f"{len(failures)} gradient norms off: {failures[:8]}"
