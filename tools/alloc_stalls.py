"""Which tensor allocations of a training step take milliseconds, and do they coincide with a new segment (hipMalloc) in the caching allocator?
usage: python tools/alloc_stalls.py [--steps 10]"""
import argparse
import os
import random
import sys
import threading
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--ms", type=float, default=3.0)
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
    log = []
    seg = [0]
    lock = threading.Lock()

    def wrap(name):
        orig = getattr(torch, name)

        def f(*args, **kw):
            s0 = seg[0]
            t0 = time.time()
            r = orig(*args, **kw)
            dt = time.time() - t0
            if dt * 1e3 > a.ms and torch.is_tensor(r) and r.is_cuda:
                s1 = torch.cuda.memory_stats()["segment.all.allocated"]          # (only after a slow call: the query itself is slow)
                seg[0] = s1
                fr = traceback.extract_stack(limit=3)[0]
                with lock:
                    log.append((dt * 1e3, name, r.numel() * r.element_size() / 2**20, s1 - s0 if s0 >= 0 else None, f"{os.path.basename(fr.filename)}:{fr.lineno}",
                                threading.current_thread().name))
            return r
        setattr(torch, name, f)
    for n in ("empty", "zeros", "full", "empty_like", "zeros_like"):
        wrap(n)
    for k in range(a.steps + 2):
        log.clear()
        torch.cuda.synchronize()
        t0 = time.time()
        step(b, 0.7, rng=random.Random(100 + k))
        torch.cuda.synchronize()
        wall = (time.time() - t0) * 1e3
        ms = torch.cuda.memory_stats()
        print(f"step {k}: {wall:.0f} ms; reserved {ms['reserved_bytes.all.current'] / 2**30:.1f} GiB, allocated peak {ms['allocated_bytes.all.peak'] / 2**30:.1f} GiB, "
              f"segments ever {ms['segment.all.allocated']}, freed {ms['segment.all.freed']}, retries {ms['num_alloc_retries']}; slow allocations: {len(log)}")
        for dt, name, mb, dseg, where, th in sorted(log, reverse=True)[:8]:
            print(f"     {dt:7.1f} ms  torch.{name:10s} {mb:9.1f} MiB  new segments {dseg}  at {where}  [{th}]")


if __name__ == "__main__":
    main()
