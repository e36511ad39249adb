"""Launch duration of the fused-rows attention step (forward and backward) at the training shapes, every clip active.
usage: python tools/attn_mq_bench.py [clips]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402

NULL = C.c_void_p(0)


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T, H = 1201, 256
    dev = torch.device("cuda:0")
    L = hip.lib()
    K = torch.exp(2 * torch.randn(clips, T, H, device=dev) * 0.5)      # key image
    enc = torch.randn(clips, T, 2 * H, device=dev)
    v = torch.randn(H, device=dev) * 0.3
    algo = clips * T * 3 * H * 4.0
    for groups in (1, 2, 3, 4, 5):
        R = groups * clips
        q = torch.randn(R, H, device=dev) * 0.5
        ctx, attw = torch.empty(R, 2 * H, device=dev), torch.empty(R, T, device=dev)
        dctx, dq, ds, dco = torch.randn(R, 2 * H, device=dev), torch.empty(R, H, device=dev), torch.empty(R, T, device=dev), torch.empty(R, 2 * H, device=dev)
        ws = hip.attn_workspace(clips, T, H, dev, groups=groups)
        f = lambda: hip.check(L.a2s_attn_step_fwd_rows(hip.stream(), hip._p(K), hip._p(enc), hip._p(q), C.c_long(H), hip._p(v), hip._p(ctx), C.c_long(2 * H),
                                                       NULL, C.c_long(0), hip._p(attw), R, T, H, hip._p(ws), clips, NULL, NULL, NULL, clips, 0), "fwd")
        b = lambda: hip.check(L.a2s_attn_step_bwd_rows(hip.stream(), hip._p(K), hip._p(enc), hip._p(q), C.c_long(H), hip._p(v), hip._p(attw), hip._p(ctx),
                                                       C.c_long(2 * H), hip._p(dctx), C.c_long(2 * H), NULL, C.c_long(0), hip._p(dco), C.c_long(2 * H),
                                                       hip._p(dq), C.c_long(H), hip._p(ds), R, T, H, hip._p(ws), clips, NULL, NULL, NULL, clips, 0), "bwd")
        tf, tb = timed(f), timed(b)
        print(f"groups {groups}: forward {tf:7.1f} us ({algo / tf / 1e3:6.0f} GB/s of K+enc, {tf / groups:6.1f} us per row-set)   "
              f"backward {tb:7.1f} us ({algo / tb / 1e3:6.0f} GB/s, {tb / groups:6.1f} us per row-set)")


if __name__ == "__main__":
    main()
