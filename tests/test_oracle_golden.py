"""Pin the oracle (oracle/model_ref.py, oracle/recipe_ref.py) against fixtures produced by the reference.

Fixtures: tests/golden/g1_small.npz (+ .json), written by tests/golden/make_golden.py from
/root/reference/models.py.  Tolerances: the oracle re-states the same fp32 arithmetic with a different
operation grouping (explicit GRU cells instead of nn.GRU), so agreement is to fp32 round-off.
"""
import copy
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import model_ref, recipe_ref
from piano_a2s_amd import spec, synthetic

SMALL_BATCH = dict(frames=41, upper_range=(3, 10), lower_range=(2, 7), full_tail=0.1)


@pytest.fixture(scope="module")
def g1(golden_dir):
    data = np.load(os.path.join(golden_dir, "g1_small.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g1_small.json")))
    cfg = spec.default_cfg(**{k: v for k, v in meta["cfg"].items()})
    batch = synthetic.make_batch(3, cfg, meta["batch_seed"], **SMALL_BATCH)
    return data, meta, cfg, batch


def _state(cfg, case):
    st = spec.procedural_state(cfg, case["weights_seed"], eos_bias=case["eos_bias"], lively=True)
    P, B = spec.split_state(st)
    return {k: v.clone() for k, v in P.items()}, {k: v.clone() for k, v in B.items()}


def _close(a, b, tol, what):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-30)
    assert err <= tol * max(1.0, scale), f"{what}: max abs err {err:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("seed", [11, 18])
def test_greedy_matches_reference(g1, seed):
    data, meta, cfg, batch = g1
    case = meta["cases"][f"greedy_s{seed}"]
    P, B = _state(cfg, case)
    with torch.no_grad():
        outs = model_ref.forward(P, B, cfg, batch[0], inference=True, training=False)
    for n, o in zip(("ts", "key", "up", "lo"), outs):
        ref = data[f"greedy_s{seed}.{n}"]
        _close(o, ref, 2e-5, f"greedy {n}")
    # token ids bit-exact, untouched rows exactly zero
    assert np.array_equal(outs[2].argmax(-1).numpy(), data[f"greedy_s{seed}.up"].argmax(-1))
    assert np.array_equal(outs[3].argmax(-1).numpy(), data[f"greedy_s{seed}.lo"].argmax(-1))
    assert np.array_equal((outs[2].abs().sum(-1) == 0).numpy(), (np.abs(data[f"greedy_s{seed}.up"]).sum(-1) == 0))


def test_eval_teacher_forced(g1):
    data, meta, cfg, batch = g1
    P, B = _state(cfg, meta["cases"]["eval_tf1"])
    gt = [batch[1], batch[2], batch[3], batch[4], batch[5], batch[6]]
    with torch.no_grad():
        outs = model_ref.forward(P, B, cfg, batch[0], inference=False, ground_truth=gt,
                                 teacher_forcing_ratio=1.0, training=False)
    for n, o in zip(("ts", "key", "up", "lo"), outs):
        _close(o, data[f"eval_tf1.{n}"], 2e-5, f"eval_tf1 {n}")


@pytest.mark.parametrize("name", ["train_tf1", "train_tf05"])
def test_train_forward_loss_grads(g1, name):
    data, meta, cfg, batch = g1
    case = meta["cases"][name]
    P, B = _state(cfg, case)
    for p in P.values():
        p.requires_grad_(True)
    gt = [batch[1], batch[2], batch[3], batch[4], batch[5], batch[6]]

    class CountingRandom(random.Random):
        n = 0

        def random(self):
            CountingRandom.n += 1
            return super().random()

    rng = CountingRandom()
    if case["random_seed"] is not None:
        rng.seed(case["random_seed"])
    outs = model_ref.forward(P, B, cfg, batch[0], inference=False, ground_truth=gt,
                             teacher_forcing_ratio=case["tf"], training=True, rng=rng, dropout=False)
    assert CountingRandom.n == case["draws"], "python-random draw protocol (a-12)"
    for n, o in zip(("ts", "key", "up", "lo"), outs):
        _close(o.detach(), data[f"{name}.{n}"], 5e-5, f"{name} {n}")
    losses = recipe_ref.objectives(outs, (batch[1], batch[2], batch[3], batch[5]))
    ref_losses = data[f"{name}.losses"]
    for i, l in enumerate(losses):
        assert abs(float(l) - ref_losses[i]) <= 1e-5 * abs(ref_losses[i]), f"loss term {i}"
    losses[0].backward()
    for k, p in P.items():
        if name == "train_tf1":
            ref = data[f"{name}.grad.{k}"]
            err = np.abs(p.grad.numpy() - ref).max()
            assert err <= 1e-4 * max(np.abs(ref).max(), 1e-6) + 1e-7, f"grad {k}: {err:.3e} vs max {np.abs(ref).max():.3e}"
        else:
            ref = float(data[f"{name}.gradnorm.{k}"])
            assert abs(float(p.grad.double().norm()) - ref) <= 1e-4 * ref + 1e-7, f"grad norm {k}"
    for k, b in B.items():
        _close(b, data[f"{name}.buf.{k}"], 1e-5, f"buffer {k}")


def test_clip_adadelta_step(g1):
    data, meta, cfg, batch = g1
    P, _ = _state(cfg, meta["cases"]["train_tf1"])
    grads = {k: torch.from_numpy(data[f"train_tf1.grad.{k}"].copy()) for k in P}
    total, _ = recipe_ref.clip_grad_norm(list(grads.values()), 5.0)
    assert abs(float(total) - float(data["step.total_norm"])) <= 1e-5 * float(data["step.total_norm"])
    state = {}
    assert recipe_ref.train_step(P, grads, state, loss_value=1.0)
    for k, p in P.items():
        _close(p, data[f"step.param.{k}"], 1e-6, f"updated {k}")
    # non-finite loss skips the step
    before = {k: v.clone() for k, v in P.items()}
    assert not recipe_ref.train_step(P, grads, state, loss_value=float("nan"))
    assert all(torch.equal(P[k], before[k]) for k in P)


def test_padding_contract():
    row = recipe_ref.pad_measure([5, 6, 7], 6)
    assert row.tolist() == [5, 6, 7, 146, 147, 147]
    assert recipe_ref.pad_measure([1, 2, 3, 4, 5, 6, 7], 6).tolist() == [1, 2, 3, 4, 5, 6]     # truncated, no <eos>
    assert recipe_ref.pad_measure([], 3).tolist() == [146, 147, 147]
    padded, lens = recipe_ref.pad_score([[1, 2], [3] * 9], 4)
    assert lens.tolist() == [2, 4] and padded[1].tolist() == [3, 3, 3, 3]
    assert np.array_equal(synthetic.pad_measure([5, 6, 7], 6), row.numpy())
    s = recipe_ref.pad_spectrogram(np.ones((3, 4), dtype=np.float32), 5)
    assert s.shape == (1, 5, 4) and s[0, 3:].abs().sum() == 0 and s[0, :3].sum() == 12
