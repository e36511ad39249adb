"""Times the first ConvStack layer's kernels (forward, weight gradient with the fused BatchNorm backward) with the compile-time-shaped kernels
on and off (a2s_debug_set("conv_c1_fast")), at the bench shape.  usage: python tools/conv_c1_check.py [--batch 256]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--frames", type=int, default=1201)
    a = ap.parse_args()
    from piano_a2s_amd import hip
    L = hip.lib()
    dev = torch.device("cuda:0")
    B, T, F, Co = a.batch, a.frames, 480, 20
    NULL = C.c_void_p(0)
    x = torch.randn(B, T, 1, F, device=dev)
    w = torch.randn(Co, 1, 3, 3, device=dev) * 0.3
    y = torch.empty(B, T, Co, F, device=dev)
    gact = torch.randn(B, T, Co, F, device=dev)
    part = torch.zeros(L.a2s_conv3x3_stat_blocks(B, T, F, 1), Co, 2, device=dev)
    amax = torch.zeros(Co, device=dev)
    mean, invstd = torch.randn(Co, device=dev) * 0.1, torch.rand(Co, device=dev) + 0.5
    scale, shift = torch.rand(Co, device=dev) + 0.5, torch.randn(Co, device=dev) * 0.1
    c12 = torch.randn(2 * Co, device=dev) * 0.01
    nb = L.a2s_conv3x3_wgrad_workspace_bytes(1, Co)
    ws = torch.empty(nb // 4, device=dev)
    dW = torch.zeros(Co, 1, 3, 3, device=dev)

    def fwd():
        hip.check(L.a2s_conv3x3_ranged(hip.stream(), hip._p(x), hip._p(w), hip._p(y), NULL, NULL, NULL, hip._p(part), hip._p(amax), B, T, F, 1, Co, NULL), "conv")

    def wgrad():
        hip.check(L.a2s_conv3x3_wgrad_bn(hip.stream(), hip._p(gact), hip._p(y), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), hip._p(c12),
                                         NULL, hip._p(x), NULL, NULL, hip._p(dW), hip._p(ws), C.c_size_t(nb), B, T, F, 1, Co), "wgrad_bn")
    gb = {"forward": (B * T * F * 4 * (1 + Co)) / 1e9, "wgrad+bn": (B * T * F * 4 * (1 + 2 * Co)) / 1e9}
    for name, fn in (("forward", fwd), ("wgrad+bn", wgrad)):
        for fast in (0, 1):
            hip.check(L.a2s_debug_set(b"conv_c1_fast", fast), "debug_set")
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"{name:9s} conv_c1_fast={fast}: {ms:6.3f} ms   {gb[name] / ms:6.2f} TB/s of the algorithmic {gb[name]:.1f} GB", flush=True)
    hip.check(L.a2s_debug_set(b"conv_c1_fast", 1), "debug_set")


if __name__ == "__main__":
    main()
