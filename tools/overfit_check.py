"""25 optimizer steps on one fixed synthetic batch: the loss must fall, and the fused step (finished rows skipped, teacher-forced bars
fused) must follow the plain per-bar step (identical for the first steps, then fp32 rounding differences grow slowly)."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import models
from piano_a2s_amd import spec, synthetic, train
dev = torch.device("cuda:0")
cfg = spec.default_cfg()
torch.manual_seed(0)
m = models.ScoreTranscription(**cfg).to(dev); m.train()
res = {}
for mode in ("plain", "fused"):
    torch.manual_seed(0)
    m = models.ScoreTranscription(**cfg).to(dev); m.train()
    step = train.TrainStep(m, dropout=False, skip_finished_rows=(mode == "fused"), fuse_bars=(mode == "fused"))
    batch = synthetic.make_batch(16, cfg, 5, spectrogram="ridges")
    batch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
    rng = random.Random(1)
    hist = []
    for k in range(25):
        l = step(batch, 0.7, rng=rng)
        hist.append(float(l[:, 0].sum()))
    res[mode] = hist
    print(mode, " ".join(f"{x:.3f}" for x in hist[::3]))
d = max(abs(a - b) / abs(a) for a, b in zip(res["plain"], res["fused"]))
print("max relative loss difference plain vs fused over 25 steps:", d)
