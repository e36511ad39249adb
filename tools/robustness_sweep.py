"""Odd shapes through the fused training step and the greedy decoder (full-width model): batch 1 / 2 / 7 / 33, frame counts that are
not multiples of the tile sizes, short and long bars.  Checks for finite losses, an applied update and exit without errors."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import models
    from piano_a2s_amd import spec, synthetic, train
    dev = torch.device("cuda:0")
    for B, T, maxlen, bars, tf in ((1, 37, (9, 5), 2, 0.7), (2, 101, (30, 17), 5, 1.0), (7, 203, (21, 33), 3, 0.5), (33, 150, (40, 24), 5, 0.8),
                                   (5, 64, (3, 2), 4, 0.0), (3, 1201, (50, 20), 5, 0.9)):
        cfg = spec.default_cfg(max_length=maxlen, max_bars=bars)
        torch.manual_seed(B)
        m = models.ScoreTranscription(**cfg).to(dev)
        m.train()
        step = train.TrainStep(m)
        batch = synthetic.make_batch(B, cfg, 3 + B, frames=T, upper_range=(1, maxlen[0]), lower_range=(1, maxlen[1]), full_tail=0.15)
        batch = [t.to(dev) if torch.is_tensor(t) else t for t in batch]
        for k in range(2):
            losses = step(batch, tf, rng=random.Random(k))
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(losses).all()) and float(step.opt.ctl[2]) == 1.0
        m.eval()
        with torch.no_grad():
            outs = m(spectrogram=batch[0], inference=True, device=dev)
        torch.cuda.synchronize()
        ok2 = all(bool(torch.isfinite(o).all()) for o in outs)
        print(f"B={B:3d} T={T:5d} max_length={maxlen} bars={bars} tf={tf}: train {'ok' if ok else 'FAILED'} loss {float(losses[:, 0].sum()):.4f}; greedy {'ok' if ok2 else 'FAILED'}")
        assert ok and ok2


if __name__ == "__main__":
    main()
