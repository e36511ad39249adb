"""Host logic of the fused step's planner (no GPU): the coins of a step drawn by engine.draw_plan in the reference's order (models.py:404,289: one per
executed note step, upper then lower, then one per bar), the bar segments, the clip-group cut and its round-5 refinement (long-clip sub-groups)."""
import random

import numpy as np
import torch

from piano_a2s_amd import engine, spec, synthetic, train
from piano_a2s_amd.spec import EOS, PAD


def _literal_draws(gt_up, gt_lo, maxlen, rng, tf):
    """The reference's loops, literally: per bar, upper then lower -- a draw after every executed step (the loop breaks once every row has shown <eos>) --
    then one draw for the bar."""
    bars = gt_up.shape[1]
    out = []
    for bar in range(bars):
        p = {}
        for gi, g in enumerate((gt_up, gt_lo)):
            B = g.shape[0]
            seen = [False] * B
            flags = []
            for t in range(maxlen[gi]):
                if all(seen):
                    break
                flags.append(rng.random() < tf)
                for b in range(B):
                    if int(g[b, bar, t]) == EOS:
                        seen[b] = True
            p[gi] = (len(flags), flags)
        p["tf"] = rng.random() < tf
        out.append(p)
    return out


def test_draw_plan_is_the_reference_draw_order():
    cfg = spec.default_cfg(max_length=(14, 9))
    batch = synthetic.make_batch(6, cfg, 3, frames=9, upper_range=(1, 14), lower_range=(1, 9), full_tail=0.2)
    gt = (batch[3], batch[5], batch[4], batch[6])
    a = engine.draw_plan(gt, cfg["max_bars"], cfg["max_length"], random.Random(5), 0.7)
    b = _literal_draws(batch[3], batch[5], cfg["max_length"], random.Random(5), 0.7)
    assert a == b
    # the plan does not depend on the order of the clips (the step counts are batch-wide): TrainStep draws it before it permutes them
    perm = torch.randperm(6, generator=torch.Generator().manual_seed(1))
    c = engine.draw_plan(tuple(t[perm] for t in gt), cfg["max_bars"], cfg["max_length"], random.Random(5), 0.7)
    assert a == c


def test_plan_segments():
    plan = [{"tf": True}, {"tf": False}, {"tf": True}, {"tf": True}, {"tf": False}]
    assert engine.plan_segments(plan, 5, True) == [[0, 1], [2, 3, 4]]
    assert engine.plan_segments(plan, 5, False) == [[0], [1], [2], [3], [4]]
    assert engine.plan_segments([{"tf": True}] * 7, 7, True) == [[0, 1, 2, 3, 4], [5, 6]]       # at most 5 bars per call


def _untils(B=256, seed=0, tail=0.01):
    rng = np.random.default_rng(seed)
    up = rng.integers(20, 121, size=(B, 5))
    lo = rng.integers(10, 81, size=(B, 5))
    up[rng.random((B, 5)) < tail] = 398
    lo[rng.random((B, 5)) < tail] = 189
    return up, lo


def test_clip_groups_and_long_subgroups():
    up, lo = _untils()
    order, n_main = train.plan_clip_groups(up, lo)
    assert sorted(order.tolist()) == list(range(256)) and 0 < n_main < 256
    long_ids = order[n_main:].tolist()
    assert all(up[c].max() == 398 for c in long_ids), "the long group holds the clips with a full-length upper bar"

    def chain(ids, segs, sequential):
        return sum((up[np.ix_(ids, s)].max() + lo[np.ix_(ids, s)].max()) if sequential else max(up[np.ix_(ids, s)].max(), lo[np.ix_(ids, s)].max()) for s in segs)
    assert train.split_long_group(long_ids, up, lo, [[0, 1, 2, 3, 4]]) is None, "one bar segment: nothing to gain"
    for segs in ([[0, 1, 2], [3, 4]], [[0], [1], [2], [3, 4]], [[0, 1], [2, 3], [4]]):
        r = train.split_long_group(long_ids, up, lo, segs)
        if r is None:
            continue
        a, b = r
        assert sorted(a + b) == sorted(long_ids) and a and b, "a partition of the long clips"
        assert max(chain(a, segs, True), chain(b, segs, True)) < 0.9 * chain(long_ids, segs, False), "taken only for a >= 10 % shorter chain"
    # at least the two-segment case must split on this data (each long clip holds ONE full-length bar)
    assert train.split_long_group(long_ids, up, lo, [[0, 1, 2], [3, 4]]) is not None
    assert train.split_long_group(long_ids[:1], up, lo, [[0, 1, 2], [3, 4]]) is None


def test_plan_clip_groups_small_batches_stay_whole():
    up, lo = _untils(B=3)
    order, n_main = train.plan_clip_groups(up, lo)
    assert n_main == 3 and order.tolist() == [0, 1, 2]


def test_plan_step_groups_matches_the_g4b_fixture():
    """train.plan_step_groups is the product's whole clip-group decision as a pure function of the targets and the coins: on the g4b minibatch and seed it
    forms the THREE groups the fixture's generator stored (tests/golden/make_golden.py g4b ran the same function next to the reference), draws the coins
    in the reference's order (1170 draws, the reference's count), and without sub-groups the same minibatch is two groups."""
    import json
    import os
    meta = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g4b_step.json")))
    cfg = spec.default_cfg()
    kw = dict(meta["batch_kwargs"])
    kw["upper_range"], kw["lower_range"] = tuple(kw["upper_range"]), tuple(kw["lower_range"])
    batch = synthetic.make_batch(meta["batch"], cfg, meta["batch_seed"], full_rows=[tuple(r) for r in meta["full_rows"]], **kw)
    gt = [batch[3], batch[5], batch[4], batch[6]]

    class Counting(random.Random):
        n = 0

        def random(self):
            Counting.n += 1
            return super().random()
    order, cuts, plan = train.plan_step_groups(gt, cfg["max_bars"], cfg["max_length"], Counting(meta["random_seed"]), meta["tf"], meta["plan_kw"], True)
    assert [list(c) for c in cuts] == meta["clip_groups"] and len(cuts) == 3
    assert order.tolist() == meta["clip_order"]
    assert plan is not None and Counting.n == meta["draws"]
    order2, cuts2, plan2 = train.plan_step_groups(gt, cfg["max_bars"], cfg["max_length"], random.Random(meta["random_seed"]), meta["tf"], meta["plan_kw"], False)
    assert cuts2 == [(0, 10), (10, 12)] and plan2 is None
    assert sorted(order2.tolist()) == list(range(meta["batch"]))
