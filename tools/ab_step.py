"""In-process A/B of a TrainStep attribute (default: skip_finished_rows): alternates the two settings step by step so that box-to-box
and thermal drift cancel.  usage: python tools/ab_step.py [--batch 256] [--pairs 6] [--attr skip_finished_rows]"""
import argparse
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--pairs", type=int, default=6)
    ap.add_argument("--reverse", action="store_true", help="run the True member of every pair first (first-occurrence effects -- a new segment structure's allocations -- then land on True)")
    ap.add_argument("--attr", default="skip_finished_rows", help="TrainStep attribute, lib:<key> for an a2s_debug_set switch, or env:<NAME> for an environment switch read per step")
    a = ap.parse_args()
    import models
    from piano_a2s_amd import spec, synthetic, train
    dev = torch.device("cuda:0")
    cfg = spec.default_cfg()
    torch.manual_seed(1)
    random.seed(1)
    m = models.ScoreTranscription(**cfg).to(dev)
    m.train()
    step = train.TrainStep(m)
    if a.attr.startswith("lib:"):
        from piano_a2s_amd import hip
        key = a.attr[4:].encode()

        class _Knob:
            def __setattr__(self, name, v):
                val = int(os.environ.get("A2S_AB_ON", "3" if key == b"conv_bf16x3" else "1")) if v else int(os.environ.get("A2S_AB_OFF", "0"))
                hip.check(hip.lib().a2s_debug_set(key, val), "a2s_debug_set")
        target = _Knob()
    elif a.attr.startswith("env:"):
        class _Env:
            def __setattr__(self, name, v):
                os.environ[name[4:]] = "1" if v else "0"
        target = _Env()
    else:
        target = step
    b = synthetic.make_batch(a.batch, cfg, 1234, full_tail=float(os.environ.get("A2S_AB_TAIL", "0.01")))
    b = [t.to(dev) if torch.is_tensor(t) else t for t in b]
    for v in (False, True):
        setattr(target, a.attr, v)
        step(b, 0.7)
    torch.cuda.synchronize()
    tot = {False: [], True: []}
    for k in range(a.pairs):
        for v in ((True, False) if a.reverse else (False, True)):
            setattr(target, a.attr, v)
            torch.cuda.synchronize()
            t0 = time.time()
            step(b, 0.7, rng=random.Random(100 + k))      # both members of a pair see the same coin flips (same fused segments)
            torch.cuda.synchronize()
            tot[v].append(time.time() - t0)
    ms = torch.cuda.memory_stats()
    import statistics
    d = [x - y for x, y in zip(tot[False], tot[True])]
    print(f"paired difference False - True: mean {statistics.mean(d) * 1e3:+.1f} ms, stdev {statistics.pstdev(d) * 1e3:.1f} ms over {len(d)} pairs")
    print("order False/True per pair (ms):", " ".join(f"{x * 1e3:.0f}/{y * 1e3:.0f}" for x, y in zip(tot[False], tot[True])))
    print(f"alloc retries {ms['num_alloc_retries']}, reserved {ms['reserved_bytes.all.peak'] / 2**30:.1f} GiB, allocated peak "
          f"{ms['allocated_bytes.all.peak'] / 2**30:.1f} GiB, hipMalloc calls {ms['segment.all.allocated']}")
    for v in (False, True):
        ts = sorted(tot[v])
        print(f"{a.attr}={v}: median {ts[len(ts) // 2] * 1e3:.1f} ms  min {ts[0] * 1e3:.1f}  max {ts[-1] * 1e3:.1f}  -> {a.batch / ts[len(ts) // 2]:.1f} clips/s")


if __name__ == "__main__":
    main()
