"""Wall time of the phases of one fused training step (HIP events on the main stream around each phase).
usage: python tools/phase_times.py [--batch 256] [--steps 4]"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--full-tail", type=float, default=0.01)
    ap.add_argument("--segments", action="store_true")
    a = ap.parse_args()
    import models
    from piano_a2s_amd import engine, engine_bwd, spec, synthetic, train
    dev = torch.device("cuda:0")
    cfg = spec.default_cfg()
    torch.manual_seed(1)
    m = models.ScoreTranscription(**cfg).to(dev)
    m.train()
    step = train.TrainStep(m)
    b = synthetic.make_batch(a.batch, cfg, 1234, full_tail=a.full_tail)
    b = [t.to(dev) if torch.is_tensor(t) else t for t in b]
    marks = []
    host_marks = []
    import time

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))

    def wrap(obj, attr, name):
        orig = getattr(obj, attr)

        def f(*args, **kw):
            mark("before " + name)
            host_marks.append((name + " called", time.time()))
            r = orig(*args, **kw)
            host_marks.append((name + " returned", time.time()))
            mark(name)
            return r
        setattr(obj, attr, f)

    wrap(engine.Engine, "convstack", "convstack fwd")
    wrap(engine.Engine, "encoder", "encoder fwd")
    wrap(engine.Engine, "forward", "decoder fwd (rest of forward)")
    wrap(engine_bwd, "_encoder_bwd", "encoder bwd")
    wrap(engine_bwd, "_convstack_bwd", "convstack bwd")
    wrap(engine_bwd, "backward", "backward tail")
    wrap(train.Objective, "__call__", "loss")
    # per clip group: when its decoder forward starts, when its backward starts (= its forward has been issued) and ends, on the group's
    # own stream, relative to the start of the step
    gmarks = []

    def gmark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()                                   # on the calling thread's current stream
        gmarks.append((name, e))

    orig_rcg = engine.run_clip_groups

    def rcg(device, fns):
        def timed(g, fn):
            def f():
                gmark(f"group {g} decoder forward starts")
                r = fn()
                gmark(f"group {g} done")
                return r
            return f
        return orig_rcg(device, [timed(g, fn) for g, fn in enumerate(fns)])
    engine.run_clip_groups = rcg
    orig_dg = engine_bwd.Backward.decoder_group

    def dg(self, gidx, *args):
        gmark(f"group {gidx} decoder backward starts")
        return orig_dg(self, gidx, *args)
    engine_bwd.Backward.decoder_group = dg
    gtot = {}
    # --segments: every note-decoder call (one staff of one segment of one clip group) with its start, duration and time per decode step, on
    # the stream it runs on (events around the call; no profiler attached)
    seg_marks = []
    if a.segments:
        import threading

        def hwrap(obj, attr, name):
            orig = getattr(obj, attr)

            def f(*args, **kw):
                host_marks.append((name + " called", time.time()))
                r = orig(*args, **kw)
                host_marks.append((name + " returned", time.time()))
                return r
            setattr(obj, attr, f)
        engine._HOST_TRACE = host_marks
        hwrap(train, "plan_clip_groups", "plan_clip_groups")
        hwrap(engine.Engine, "_keys", "keys")

        lock = threading.Lock()
        orig_ds = engine.Engine._decode_staff

        def ds(self, S, prefix, keys, enc, h0, maxs, probs, gt_bar, steps, *rest, **kw):
            if kw.get("defer_launch"):          # Engine._decode_pair: timed there
                return orig_ds(self, S, prefix, keys, enc, h0, maxs, probs, gt_bar, steps, *rest, **kw)
            host_marks.append((f"host reaches fwd {prefix.split('.')[-1][:5]} rows {h0.shape[0]}", time.time()))
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_ds(self, S, prefix, keys, enc, h0, maxs, probs, gt_bar, steps, *rest)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            with lock:
                seg_marks.append((f"fwd {prefix.split('.')[-1][:5]} rows {h0.shape[0]:4d} steps {steps:3d}", e0, e1, steps))
            return r
        engine.Engine._decode_staff = ds
        orig_dp = engine.Engine._decode_pair

        def dp(self, calls, streams, pair):
            host_marks.append((f"host reaches fwd pair rows {calls[0][4].shape[0]}", time.time()))
            ev = []
            for st in streams:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(st)
                ev.append(e0)
            r = orig_dp(self, calls, streams, pair)
            for st, e0, args in zip(streams, ev, calls):
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(st)
                with lock:
                    seg_marks.append((f"fwd {args[1].split('.')[-1][:5]} rows {args[4].shape[0]:4d} steps {args[8]:3d} (one loop for both staves)", e0, e1, args[8]))
            return r
        engine.Engine._decode_pair = dp
        orig_nbp = engine_bwd._note_decoder_bwd_pair

        def nbp(calls, streams, pair):
            ev = []
            for st in streams:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(st)
                ev.append(e0)
            r = orig_nbp(calls, streams, pair)
            for st, e0, args in zip(streams, ev, calls):
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(st)
                sv = args[3]
                with lock:
                    seg_marks.append((f"bwd {sv['prefix'].split('.')[-1][:5]} rows {sv['groups'] * args[5].shape[0]:4d} steps {sv['steps']:3d} (one loop for both staves; incl. deferred products)", e0, e1, sv["steps"]))
            return r
        engine_bwd._note_decoder_bwd_pair = nbp
        orig_nb = engine_bwd._note_decoder_bwd

        def nb(eng, S, G, sv, keys, enc, *rest, **kw):
            if kw.get("defer_launch"):
                return orig_nb(eng, S, G, sv, keys, enc, *rest, **kw)
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_nb(eng, S, G, sv, keys, enc, *rest, **kw)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            with lock:
                seg_marks.append((f"bwd {sv['prefix'].split('.')[-1][:5]} rows {sv['groups'] * enc.shape[0]:4d} steps {sv['steps']:3d} (incl. its deferred products)", e0, e1, sv["steps"]))
            return r
        engine_bwd._note_decoder_bwd = nb
    totals = {}
    walls = []
    for k in range(a.steps + 1):
        marks.clear()
        gmarks.clear()
        seg_marks.clear()
        host_marks.clear()
        mark("start")
        t0 = time.time()
        step(b, 0.7, rng=random.Random(100 + k))
        mark("optimizer")
        torch.cuda.synchronize()
        walls.append((time.time() - t0) * 1e3)
        if k == 0:
            continue
        if a.segments:
            # host-side stalls of THIS step: consecutive milestones of one host thread (g0 / g1 / the caller) more than 15 ms apart
            by = {}
            for n, t in host_marks:
                key = n[:2] if n[:2] in ("g0", "g1") else "main"
                by.setdefault(key, []).append((n, t))
            for key, seq in sorted(by.items()):
                for (n0, ta), (n1, tb) in zip(seq[:-1], seq[1:]):
                    if tb - ta > 0.015:
                        print(f"step {k} ({walls[-1]:.0f} ms) host thread {key}: {1e3 * (tb - ta):6.1f} ms between '{n0}' (at {1e3 * (ta - t0):.1f}) and '{n1}'")
        for n, e in list(gmarks):
            gtot.setdefault(n, []).append(marks[0][1].elapsed_time(e))
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            name = n1 if not n1.startswith("before ") else {"before encoder bwd": "decoder bwd", "before convstack bwd": "encoder->conv glue"}.get(n1, "(gap) " + n1)
            totals[name] = totals.get(name, 0.0) + e0.elapsed_time(e1)
    tot = sum(totals.values())
    for n, t in totals.items():
        print(f"{t / a.steps:9.1f} ms  {100 * t / tot:5.1f} %  {n}")
    if a.segments:
        print("host clock of the LAST step (ms after the step was called): " + "; ".join(f"{n} {1e3 * (t - t0):.1f}" for n, t in host_marks))
        print("note-decoder calls of the LAST step (start ms after the step's start, duration ms, us per decode step):")
        for n, e0, e1, st in sorted(seg_marks, key=lambda m: marks[0][1].elapsed_time(m[1])):
            t0, d = marks[0][1].elapsed_time(e0), e0.elapsed_time(e1)
            print(f"  {t0:7.1f} {d:7.1f} {1e3 * d / max(st, 1):7.1f}  {n}")
    for n, t in sorted(gtot.items(), key=lambda kv: sorted(kv[1])[len(kv[1]) // 2]):
        print(f"{sorted(t)[len(t) // 2]:9.1f} ms after the start of the step (median; per step " + " ".join(f"{x:.0f}" for x in t) + f"): {n}")
    print("wall ms per step (host clock, first = warm-up):", " ".join(f"{w:.0f}" for w in walls), " groups:", step._last[2])
    print(f"{tot / a.steps:9.1f} ms  total  -> {a.batch / (tot / a.steps) * 1e3:.1f} clips/s")


if __name__ == "__main__":
    main()
