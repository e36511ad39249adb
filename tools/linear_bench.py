"""The three GEMMs of the 19200 -> 256 Linear at a training shape (rows = B x 1201), two-term fp16 split path: forward (operand
BatchNorm+ReLU), data gradient (+ BatchNorm-backward statistics epilogue), weight gradient (split-K).  usage: python tools/linear_bench.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402


def timed(fn, iters=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    rows, F, Cf = B * 1201, 480, 256
    K = 40 * F
    dev = torch.device("cuda:0")
    L = hip.lib()
    y4 = torch.randn(rows, K, device=dev)
    W = torch.randn(Cf, K, device=dev) * 0.007
    Wt = W.t().contiguous()
    dz = torch.randn(rows, Cf, device=dev) * 1e-5
    scale, shift = torch.rand(40, device=dev) + 0.5, torch.randn(40, device=dev) * 0.1
    mean, invstd = torch.zeros(40, device=dev), torch.ones(40, device=dev)
    wmax, dmax = hip.absmax(W), hip.absmax(dz)
    z = torch.empty(rows, Cf, device=dev)
    da = torch.empty(rows, K, device=dev)
    G = torch.zeros(Cf, K, device=dev)
    nblk = L.a2s_gemm_bnstats_blocks(rows, F)
    part = torch.empty((nblk, 40, 2), device=dev)
    fl = 2.0 * rows * K * Cf
    ms = timed(lambda: hip.linear(y4, W, out=z, x_affine=(scale, shift, F), two_term=(None, wmax)))
    print(f"forward        {ms:7.2f} ms  {fl / ms / 1e9:6.1f} TFLOP/s   (generic 256x256 two-term tile)")
    if L.a2s_linear_fwd_eligible(rows, Cf, K, F):
        bound = hip.act_bound(scale, shift, y4.view(rows, 40, F).abs().amax(dim=(0, 2)).contiguous())
        ms2 = timed(lambda: hip.linear_forward(y4, W, (scale, shift, F), bound, wmax, out=z))
        print(f"forward        {ms2:7.2f} ms  {fl / ms2 / 1e9:6.1f} TFLOP/s   {rows * K * 4.0 / ms2 / 1e6:6.0f} GB/s  (csrc/a2s_linear.hip)")
    if hasattr(L, "a2s_gemm_trace_read"):                # library built with -DGEMM_TRACE: phase timeline of the forward launch
        import numpy as np
        buf = np.zeros(8 * 24 * 8, dtype=np.uint64)
        assert L.a2s_gemm_trace_read(buf.ctypes.data_as(C.c_void_p)) == 0
        t = buf.reshape(8, 24, 8).astype(np.int64)
        d = np.stack([t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2], t[:, :, 4] - t[:, :, 3], t[:, :, 5] - t[:, :, 4]], axis=-1)
        for n, v in zip(["barrier1 wait", "stage (wait loads + split + ds_write)", "barrier2 wait", "issue next loads", "multiply (issue)"], d.reshape(-1, 5).mean(axis=0)):
            print(f"    {n:40s} {v:8.0f} clocks per k-tile")
        print(f"    k-tile period {(t[:, -1, 0] - t[:, 0, 0]).mean() / 23:.0f} clocks")
    ms = timed(lambda: hip.check(L.a2s_gemm_f32_bnstats_scaled(hip.stream(), rows, K, Cf, hip._p(dz), C.c_long(Cf), C.c_long(1), hip._p(Wt), C.c_long(1), C.c_long(Cf),
                                                               hip._p(da), C.c_long(K), hip._p(y4), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), F,
                                                               hip._p(part), hip._p(dmax), hip._p(wmax)), "bnstats"))
    print(f"data gradient  {ms:7.2f} ms  {fl / ms / 1e9:6.1f} TFLOP/s   (generic 256x256 two-term tile)")
    if L.a2s_linear_dgrad_eligible(rows, K, Cf, F):
        nb = L.a2s_linear_dgrad_ws_bytes(K, Cf)
        lws = torch.empty(nb // 4, dtype=torch.float32, device=dev)
        part2 = torch.empty((L.a2s_linear_dgrad_blocks(rows), 40, 2), device=dev)
        ms = timed(lambda: hip.check(L.a2s_linear_dgrad_bnstats(hip.stream(), rows, K, Cf, hip._p(dz), C.c_long(Cf), hip._p(Wt), hip._p(da), C.c_long(K), hip._p(y4),
                                                                hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), F, hip._p(part2), hip._p(dmax), hip._p(wmax),
                                                                hip._p(lws), C.c_size_t(nb), C.c_void_p(0)), "linear_dgrad"))
        print(f"data gradient  {ms:7.2f} ms  {fl / ms / 1e9:6.1f} TFLOP/s   {(2.0 * rows * K * 4 + rows * Cf * 4) / ms / 1e6:6.0f} GB/s  (csrc/a2s_linear.hip)")
    sk = L.a2s_gemm_pick_splitk(Cf, K, rows, 1)
    ms = timed(lambda: hip.gemm(dz, 1, Cf, y4, K, 1, G, K, Cf, K, rows, beta=1.0, splitk=sk, b_affine=(scale, shift, F), two_term=(dmax, None)))
    print(f"weight gradient{ms:7.2f} ms  {fl / ms / 1e9:6.1f} TFLOP/s  (generic 256x256 two-term tile, split-K {sk})")
    if L.a2s_linear_wgrad_eligible(rows, Cf, K, F):
        bound = hip.act_bound(scale, shift, y4.view(rows, 40, F).abs().amax(dim=(0, 2)).contiguous())
        ms = timed(lambda: hip.linear_wgrad(dz, y4, (scale, shift, F), dmax, bound, G))
        print(f"weight gradient{ms:7.2f} ms  {fl / ms / 1e9:6.1f} TFLOP/s   {rows * K * 4.0 / ms / 1e6:6.0f} GB/s  (csrc/a2s_linear.hip)")


if __name__ == "__main__":
    main()
