"""Launch durations of the ConvStack MFMA kernels at the training shapes.  usage: python tools/conv_bench.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402


def timed(fn, iters=4):
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
    T, F = 1201, 480
    dev = torch.device("cuda:0")
    L = hip.lib()
    for ci, co, flip, what in ((1, 20, 0, "conv1 fwd"), (20, 20, 0, "conv2 fwd"), (20, 40, 0, "conv3 fwd"), (40, 40, 0, "conv4 fwd"), (40, 40, 1, "conv4 dgrad"), (40, 20, 1, "conv3 dgrad")):
        x = torch.randn(B, T, ci, F, device=dev)
        y = torch.empty(B, T, co, F, device=dev)
        w = torch.randn((ci, co, 3, 3) if flip else (co, ci, 3, 3), device=dev) * 0.05
        scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        nblk = L.a2s_conv3x3_stat_blocks(B, T, F, ci)
        partial = torch.empty(nblk, co, 2, device=dev)
        cws = hip.conv_workspace(ci, dev)
        if ci == 1:
            scale = shift = None
        ms = timed(lambda: hip.check(L.a2s_conv3x3(hip.stream(), hip._p(x), hip._p(w), hip._p(y), hip._p(None if flip else scale), hip._p(None if flip else shift),
                                                   hip._p(None if flip else partial), B, T, F, ci, co, flip, hip._p(cws)), "conv"))
        fl = 2.0 * 9 * ci * co * B * T * F
        print(f"{what:12s} {ci:2d}->{co:2d}  {ms:8.2f} ms  {fl / ms / 1e9:6.1f} TFLOP/s")
        del x, y
    for ci, co, what in ((1, 20, "conv1 wgrad"), (20, 20, "conv2 wgrad"), (20, 40, "conv3 wgrad"), (40, 40, "conv4 wgrad")):
        x = torch.randn(B, T, ci, F, device=dev)
        dy = torch.randn(B, T, co, F, device=dev)
        dW = torch.zeros(co, ci, 3, 3, device=dev)
        scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        nbytes = L.a2s_conv3x3_wgrad_workspace_bytes(ci, co)
        ws = torch.empty(nbytes // 4, device=dev)
        if ci == 1:
            scale = shift = None
        ms = timed(lambda: hip.check(L.a2s_conv3x3_wgrad(hip.stream(), hip._p(dy), hip._p(x), hip._p(scale), hip._p(shift), hip._p(dW), hip._p(ws),
                                                         C.c_size_t(nbytes), B, T, F, ci, co), "wgrad"))
        fl = 2.0 * 9 * ci * co * B * T * F
        print(f"{what:12s} {ci:2d}->{co:2d}  {ms:8.2f} ms  {fl / ms / 1e9:6.1f} TFLOP/s")
        del x, dy


if __name__ == "__main__":
    main()
