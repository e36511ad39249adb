"""Accuracy (against float64) and launch duration of the 128x128 GEMM tile with k-contiguous operands: fp32-input MFMA against the
3-term bf16 split (a2s_debug_set "gemm_bf16x3"), at the shapes of the 19200->256 Linear (with the operand BatchNorm+ReLU) and of the
encoder's input projections.  usage: python tools/gemm_split_check.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402


def timed(fn, iters=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 1201
    dev = torch.device("cuda:0")
    L = hip.lib()
    prev = L.a2s_debug_get(b"gemm_bf16x3")
    torch.manual_seed(0)
    for (M, K, N, affine) in ((1000, 19200, 256, True), (777, 1000, 384, False), (rows, 19200, 256, True), (rows, 512, 768, False)):
        x = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))
        w = torch.randn(N, K, device=dev) * 0.02
        b = torch.randn(N, device=dev)
        aff = (torch.rand(K // 480 if affine else 1, device=dev) + 0.5, torch.randn(K // 480 if affine else 1, device=dev) * 0.3, 480) if affine else None
        line = f"M {M:6d} K {K:5d} N {N:4d} affine {int(affine)}:"
        ref = mag = None
        if M <= 2000:
            xd = x.double()
            if affine:
                xd = torch.relu(xd * aff[0].double().repeat_interleave(480) + aff[1].double().repeat_interleave(480))
            ref = xd @ w.double().t() + b.double()
            mag = xd.abs() @ w.double().abs().t() + b.double().abs() + 1e-30
        for mode in (0, 1):
            L.a2s_debug_set(b"gemm_bf16x3", mode)
            y = hip.linear(x, w, b, x_affine=aff)
            torch.cuda.synchronize()
            if ref is not None:
                e = (y.double() - ref) / mag
                line += f"   {'split' if mode else 'fp32 '} max {e.abs().max().item():.2e} mean {e.mean().item():+.2e} std {e.std().item():.2e}"
            else:
                out = torch.empty(M, N, device=dev)
                ms = timed(lambda: hip.linear(x, w, b, out=out, x_affine=aff))
                line += f"   {'split' if mode else 'fp32 '} {ms:7.2f} ms {2.0 * M * N * K / ms / 1e9:6.1f} TFLOP/s"
        print(line)
    L.a2s_debug_set(b"gemm_bf16x3", prev)


if __name__ == "__main__":
    main()
