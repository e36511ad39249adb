"""A few fused training steps at the bench's shapes, nothing else (for rocprofv3 passes over the kernels the step REALLY launches, e.g. the
BatchNorm-backward-fused weight-gradient instantiations): python tools/step_once.py [batch] [steps] [full_tail] [bench]
(`bench`: the benchmark's bar lengths and teacher-forcing ratio instead of short bars -- for passes over the decoder's kernels; prints the step's
(clip, step) pair counts, from which the attention sweeps' algorithmic bytes follow)."""
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
    bench = len(sys.argv) > 4 and sys.argv[4] == "bench"
    if bench:
        b = synthetic.make_batch(B, cfg, 1234, full_tail=tail)
    else:
        b = synthetic.make_batch(B, cfg, 1234, full_tail=tail, upper_range=(4, 12), lower_range=(3, 8))
    b = [t.to(dev) if torch.is_tensor(t) else t for t in b]
    pairs = []
    for k in range(steps):
        step(b, 0.7 if bench else 1.0, rng=random.Random(k))
        pairs.append((step.attn_clip_steps, getattr(step, "attn_shared_clip_steps", 0)))
    torch.cuda.synchronize()
    print("done", step.report())
    T, H = 1201, cfg["hidden_size"]
    cs, sh = sum(p[0] for p in pairs), sum(p[1] for p in pairs)
    print(f"(clip, step) pairs over {steps} steps: {cs}, of which served by a pass shared between the staves: {sh}; algorithmic bytes of the forward sweeps "
          f"{(cs * 3 - sh * 2) * T * H * 4 / 1e9:.2f} GB (keys T x H + encoder outputs T x 2H per pair, fp32), of the backward sweeps the same")


if __name__ == "__main__":
    main()
