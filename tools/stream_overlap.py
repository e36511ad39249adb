"""Do two HIP streams really run concurrently here?  Stream A: a train of big bandwidth-bound kernels; stream B: a chain of tiny dependent
kernels.  Times B's chain alone, A alone, and both together, for the engine's own stream objects."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import engine
dev = torch.device("cuda:0")
big = torch.empty(256 * 1201 * 768, device=dev)          # 945 MB, the attention working set
big2 = torch.empty_like(big)
small = torch.zeros(64, 512, device=dev)
sa, sb = engine.side_streams(dev)
sg = engine.group_stream(dev, 1)
cur = torch.cuda.current_stream()


def run(streamA, streamB, nA=60, nB=2000):
    torch.cuda.synchronize()
    t0 = time.time()
    if streamA is not None:
        with torch.cuda.stream(streamA):
            for _ in range(nA):
                big2.copy_(big)
    if streamB is not None:
        with torch.cuda.stream(streamB):
            x = small
            for _ in range(nB):
                x = x + 1.0
    torch.cuda.synchronize()
    return (time.time() - t0) * 1e3


for name, A, B in (("side0 + group", sa, sg), ("default + side0", cur, sa)):
    run(A, B, 5, 50)
    for nA, nB in ((60, 2000), (60, 8000), (240, 8000)):
        a = run(A, None, nA, nB)
        b = run(None, B, nA, nB)
        ab = run(A, B, nA, nB)
        print(f"{name:18s} nA {nA} nB {nB}: A alone {a:7.1f} ms   B alone {b:7.1f} ms   together {ab:7.1f} ms   (serial would be {a + b:.1f}, perfect overlap {max(a, b):.1f})")
