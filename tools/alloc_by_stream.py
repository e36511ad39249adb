"""Which stream's pool of the caching allocator holds how much after N fused training steps over the bench's minibatches (round 6: sizing of
train.reserve_pools).  usage: python tools/alloc_by_stream.py [batch] [steps]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import models
    from piano_a2s_amd import engine, spec, synthetic, train
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    dev = torch.device("cuda:0")
    cfg = spec.default_cfg()
    torch.manual_seed(1)
    random.seed(1234)
    m = models.ScoreTranscription(**cfg).to(dev).train()
    step = train.TrainStep(m)
    if len(sys.argv) > 3 and hasattr(train, "reserve_pools"):
        train.reserve_pools(dev, float(sys.argv[3]))
    batches = []
    for i in range(8):
        b = synthetic.make_batch(B, cfg, 1234 + i, full_tail=0.01)
        batches.append([t.to(dev) if torch.is_tensor(t) else t for t in b])
    segs = []
    for k in range(steps):
        step(batches[k % 8], 0.7)
        segs.append(torch.cuda.memory_stats().get("segment.all.allocated", 0))
    torch.cuda.synchronize()
    print("hipMalloc segments per step:", [b - a for a, b in zip([0] + segs[:-1], segs)])
    names = {0: "default"}
    for i, s in enumerate(engine.side_streams(dev)):
        names[s.cuda_stream] = f"side{i}"
    for g in (1, 2):
        names[engine.group_stream(dev, g).cuda_stream] = f"group{g}"
    by = {}
    for seg in torch.cuda.memory_snapshot():
        k = names.get(seg["stream"], hex(seg["stream"]))
        a = by.setdefault(k, [0, 0, 0])
        a[0] += seg["total_size"]; a[1] += seg["allocated_size"]; a[2] += 1
    for k, (tot, alloc, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
        print(f"stream {k:10s}: reserved {tot / 2**30:7.2f} GiB in {n:4d} segments, allocated now {alloc / 2**30:7.2f} GiB")
    print("peak allocated %.1f GiB, reserved %.1f GiB" % (torch.cuda.max_memory_allocated() / 2**30, torch.cuda.max_memory_reserved() / 2**30))


if __name__ == "__main__":
    main()
