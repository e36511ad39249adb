"""A few fused training steps at the bench's shapes, nothing else (for rocprofv3 passes over the kernels the step REALLY launches, e.g. the
BatchNorm-backward-fused weight-gradient instantiations): python tools/step_once.py [batch] [steps] [full_tail]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import models
    from piano_a2s_amd import spec, synthetic, train
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    tail = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    dev = torch.device("cuda:0")
    cfg = spec.default_cfg()
    torch.manual_seed(1)
    m = models.ScoreTranscription(**cfg).to(dev)
    m.train()
    step = train.TrainStep(m)
    # short bars: the decoder is not what these passes are after
    b = synthetic.make_batch(B, cfg, 1234, full_tail=tail, upper_range=(4, 12), lower_range=(3, 8))
    b = [t.to(dev) if torch.is_tensor(t) else t for t in b]
    for k in range(steps):
        step(b, 1.0, rng=random.Random(k))
    torch.cuda.synchronize()
    print("done", step.report())


if __name__ == "__main__":
    main()
