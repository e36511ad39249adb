"""Median time of the fused training step over fixed minibatches and coins (for A/B runs of process-wide switches that are read once: run it twice
with the environment set / unset on the same box): python tools/step_time.py [batch] [steps] [full_tail]"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import models
    from piano_a2s_amd import spec, synthetic, train
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    tail = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
    dev = torch.device("cuda:0")
    cfg = spec.default_cfg()
    torch.manual_seed(1)
    m = models.ScoreTranscription(**cfg).to(dev)
    m.train()
    step = train.TrainStep(m)
    batches = []
    for i in range(4):
        b = synthetic.make_batch(B, cfg, 1234 + i, full_tail=tail)
        batches.append([t.to(dev) if torch.is_tensor(t) else t for t in b])
    for k in range(3):
        step(batches[k % 4], 0.7, rng=random.Random(50 + k))
    torch.cuda.synchronize()
    ts = []
    for k in range(steps):
        torch.cuda.synchronize()
        t0 = time.time()
        step(batches[k % 4], 0.7, rng=random.Random(100 + k))
        torch.cuda.synchronize()
        ts.append((time.time() - t0) * 1e3)
    s = sorted(ts)
    print(f"median {s[len(s) // 2]:.1f} ms  mean {sum(ts) / len(ts):.1f}  steps: " + " ".join(f"{t:.0f}" for t in ts))


if __name__ == "__main__":
    main()
