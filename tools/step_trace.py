"""A few fused training steps on a bench-like minibatch (for rocprofv3 --kernel-trace): python tools/step_trace.py [batch] [steps] [full_tail]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import models
    from piano_a2s_amd import spec, synthetic, train
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    tail = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    dev = torch.device("cuda:0")
    cfg = spec.default_cfg()
    torch.manual_seed(1)
    m = models.ScoreTranscription(**cfg).to(dev)
    m.train()
    step = train.TrainStep(m)
    b = synthetic.make_batch(B, cfg, 1234, full_tail=tail)
    b = [t.to(dev) if torch.is_tensor(t) else t for t in b]
    for k in range(steps):
        step(b, 0.7, rng=random.Random(100 + k))
    torch.cuda.synchronize()
    print("done", step.report())


if __name__ == "__main__":
    main()
