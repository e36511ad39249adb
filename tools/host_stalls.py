"""Who holds the host while a training step stalls?  faulthandler's watchdog thread dumps the Python stack of EVERY thread every few
milliseconds WITHOUT needing the interpreter lock, so a thread that sits inside a blocking HIP call with the lock held shows up as the one
whose innermost frame does not change while the others wait.  Prints, per step, the frames that were on top of some thread in at least
`--min` consecutive samples.
usage: python tools/host_stalls.py [--steps 6] [--period 0.004] [--min 5]"""
import argparse
import collections
import faulthandler
import os
import random
import re
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--period", type=float, default=0.004)
    ap.add_argument("--min", type=int, default=5)
    a = ap.parse_args()
    import models
    from piano_a2s_amd import spec, synthetic, train
    dev = torch.device("cuda:0")
    cfg = spec.default_cfg()
    torch.manual_seed(1)
    m = models.ScoreTranscription(**cfg).to(dev)
    m.train()
    step = train.TrainStep(m)
    b = synthetic.make_batch(a.batch, cfg, 1234, full_tail=0.01)
    b = [t.to(dev) if torch.is_tensor(t) else t for t in b]
    for k in range(2):
        step(b, 0.7, rng=random.Random(100 + k))
    torch.cuda.synchronize()
    for k in range(a.steps):
        f = tempfile.TemporaryFile(mode="w+")
        torch.cuda.synchronize()
        faulthandler.dump_traceback_later(a.period, repeat=True, file=f)
        t0 = time.time()
        step(b, 0.7, rng=random.Random(102 + k))
        t_issue = time.time() - t0
        torch.cuda.synchronize()
        wall = time.time() - t0
        faulthandler.cancel_dump_traceback_later()
        f.seek(0)
        text = f.read()
        samples = text.split("Timeout (")[1:]
        # per sample: thread id -> (innermost frame, second frame)
        runs = collections.defaultdict(lambda: [None, 0, 0])      # thread -> [frame, current run, sample index of run start]
        longest = collections.Counter()
        for si, smp in enumerate(samples):
            for blk in re.split(r"\n(?=Thread 0x|Current thread 0x)", smp):
                mt = re.match(r"(?:Thread|Current thread) (0x[0-9a-f]+)", blk.strip())
                if not mt:
                    continue
                frames = re.findall(r'File "([^"]+)", line (\d+) in (\S+)', blk)
                if not frames:
                    continue
                top = tuple((os.path.basename(p), int(l), fn) for p, l, fn in frames[:3])
                r = runs[mt.group(1)]
                if r[0] == top:
                    r[1] += 1
                else:
                    if r[0] is not None and r[1] >= a.min:
                        longest[(mt.group(1), r[0], r[2])] = r[1]
                    r[0], r[1], r[2] = top, 1, si
        for tid, r in runs.items():
            if r[0] is not None and r[1] >= a.min:
                longest[(tid, r[0], r[2])] = r[1]
        print(f"step {k}: wall {wall * 1e3:.0f} ms, host issue {t_issue * 1e3:.0f} ms, {len(samples)} samples")
        for (tid, top, start), n in sorted(longest.items(), key=lambda kv: kv[0][2]):
            print(f"   thread {tid[-6:]} from sample {start:4d} (~{start * a.period * 1e3:5.0f} ms) for {n:3d} samples (~{n * a.period * 1e3:4.0f} ms): "
                  + " <- ".join(f"{p}:{l} {fn}" for p, l, fn in top))


if __name__ == "__main__":
    main()
