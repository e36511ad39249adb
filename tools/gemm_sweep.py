"""Sweep tile configuration / split-K of a2s_gemm_f32 on the decoder's per-step shapes.  usage: python tools/gemm_sweep.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402


def timed(fn, iters=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    L = hip.lib()
    shapes = [("q   = h Wh^T", 256, 512), ("gh  = h Whh^T", 1536, 512), ("gi  = x Wih^T", 1536, 528), ("out = o Wo^T", 173, 1024),
              ("q|gh fused", 1792, 512), ("dx  = dgi Wih", 528, 1536), ("dh  = dgh Whh", 512, 1536), ("dh  = [dgh|dq] W", 512, 1792)]
    for M in (256, 512, 768, 1280):
        ws = hip.gemm_workspace(M, dev)
        print(f"--- M = {M}")
        for name, N, K in shapes:
            A = torch.randn(M, K, device=dev)
            W = torch.randn(N, K, device=dev)
            Cm = torch.empty(M, N, device=dev)
            res = []
            for tile in (0, 1, 2, 3, 4):
                for sk in ((0, 1) if tile == 0 else (1, 2, 4)):
                    L.a2s_gemm_debug_tile(tile)
                    t = timed(lambda: hip.check(L.a2s_gemm_f32(hip.stream(), M, N, K, hip.f32(1.0), hip._p(A), C.c_long(K), C.c_long(1), hip._p(W), C.c_long(1),
                                                               C.c_long(K), hip.f32(0.0), hip._p(Cm), C.c_long(N), C.c_void_p(0), 0, 1, C.c_long(0), C.c_long(0),
                                                               C.c_long(0), sk, hip._p(ws), C.c_size_t(ws.numel() * 4)), "gemm"))
                    res.append((t, tile, sk))
            L.a2s_gemm_debug_tile(0)
            auto = [r for r in res if r[1] == 0 and r[2] == 0][0][0]
            best = min(res)
            gf = 2.0 * M * N * K / 1e3
            print(f"{name:18s} N={N:5d} K={K:5d}  auto {auto:6.1f} us ({gf / auto / 1e3:5.1f} TF/s)  best {best[0]:6.1f} us tile {best[1]} splitk {best[2]}   all: "
                  + " ".join(f"{t:.1f}[{ti},{sk}]" for t, ti, sk in sorted(res)[:5]))


if __name__ == "__main__":
    main()
