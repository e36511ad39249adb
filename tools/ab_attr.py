"""In-process comparison of several values of a module attribute (e.g. piano_a2s_amd.engine_bwd._LIN_WGRAD_AT), step by step on one box: every round runs one
step per value with the same coins.  usage: python tools/ab_attr.py piano_a2s_amd.engine_bwd._LIN_WGRAD_AT '"before"' '"beside"' 4 3 2 [--rounds 8] [--tail 0.01]"""
import importlib
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    argv = sys.argv[1:]
    rounds, tail = 8, 0.01
    for flag in ("--rounds", "--tail"):
        if flag in argv:
            i = argv.index(flag)
            v = argv[i + 1]
            del argv[i:i + 2]
            if flag == "--rounds":
                rounds = int(v)
            else:
                tail = float(v)
    path, values = argv[0], [eval(v) for v in argv[1:]]
    modname, attr = path.rsplit(".", 1)
    mod = importlib.import_module(modname)
    import models
    from piano_a2s_amd import spec, synthetic, train
    dev = torch.device("cuda:0")
    cfg = spec.default_cfg()
    torch.manual_seed(1)
    m = models.ScoreTranscription(**cfg).to(dev).train()
    step = train.TrainStep(m)
    batches = []
    for i in range(4):
        b = synthetic.make_batch(256, cfg, 1234 + i, full_tail=tail)
        batches.append([t.to(dev) if torch.is_tensor(t) else t for t in b])
    for v in values:
        setattr(mod, attr, v)
        step(batches[0], 0.7, rng=random.Random(50))
    torch.cuda.synchronize()
    ts = {repr(v): [] for v in values}
    for k in range(rounds):
        order = values if k % 2 == 0 else values[::-1]
        for v in order:
            setattr(mod, attr, v)
            torch.cuda.synchronize()
            t0 = time.time()
            step(batches[k % 4], 0.7, rng=random.Random(100 + k))
            torch.cuda.synchronize()
            ts[repr(v)].append((time.time() - t0) * 1e3)
    base = ts[repr(values[0])]
    for v in values:
        t = ts[repr(v)]
        d = [a - b for a, b in zip(t, base)]
        print(f"{path} = {v!r:10}: mean {sum(t) / len(t):7.1f} ms   vs first {sum(d) / len(d):+6.1f} ms   steps " + " ".join(f"{x:.0f}" for x in t))


if __name__ == "__main__":
    main()
