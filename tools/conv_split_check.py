"""Accuracy (against float64) and launch duration of the two 3x3-convolution kernels: fp32-input MFMA and the 3-term bf16 split.
usage: python tools/conv_split_check.py [B]"""
import os
import sys

import torch
import torch.nn.functional as Fn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402


def run(L, x, w, scale, shift, flip, co, stats=True):
    B, T, ci, F = x.shape
    dev = x.device
    y = torch.full((B, T, co, F), float("nan"), device=dev)
    nblk = L.a2s_conv3x3_stat_blocks(B, T, F, ci)
    partial = torch.zeros(nblk, co, 2, device=dev)
    cws = hip.conv_workspace(ci, dev)
    hip.check(L.a2s_conv3x3(hip.stream(), hip._p(x), hip._p(w), hip._p(y), hip._p(scale), hip._p(shift), hip._p(partial if stats else None),
                            B, T, F, ci, co, flip, hip._p(cws)), "conv")
    torch.cuda.synchronize()
    return y, partial


def reference(x, w, scale, shift, flip):
    xd = x.double()
    if scale is not None:
        xd = torch.relu(xd * scale.double()[None, None, :, None] + shift.double()[None, None, :, None])
    wd = w.double()
    if flip:
        wd = wd.transpose(0, 1).flip(2, 3)
    return Fn.conv2d(xd.permute(0, 2, 1, 3), wd, padding=1).permute(0, 2, 1, 3)


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
    dev = torch.device("cuda:0")
    L = hip.lib()
    torch.manual_seed(0)
    for (b, T, F) in ((2, 37, 100), (1, 9, 481), (2, 64, 480)):
        for ci, co, flip in ((20, 20, 0), (20, 40, 0), (40, 40, 0), (40, 40, 1), (40, 20, 1), (20, 20, 1)):
            x = torch.randn(b, T, ci, F, device=dev) * torch.exp(torch.randn(b, T, ci, F, device=dev))
            w = torch.randn((ci, co, 3, 3) if flip else (co, ci, 3, 3), device=dev) * 0.05
            scale = None if flip else torch.rand(ci, device=dev) + 0.5
            shift = None if flip else torch.randn(ci, device=dev) * 0.1
            ref = reference(x, w, scale, shift, flip)
            mag = reference(x.abs() if flip else x, w.abs(), scale, shift, flip).abs() + 1e-30     # sum |a||b|: the error scale
            out, bias = {}, {}
            for mode in (0, 3):
                L.a2s_debug_set(b"conv_bf16x3", mode)
                y, part = run(L, x, w, scale, shift, flip, co, stats=not flip)
                err = ((y.double() - ref).abs() / mag).max().item()
                bias[mode] = ((y.double() - ref) / mag).mean().item(), ((y.double() - ref) / mag).std().item()
                s = part.double().sum(0)
                serr = (s[:, 0] - ref.sum((0, 1, 3))).abs().max().item() if not flip else 0.0
                out[mode] = (err, serr, torch.isnan(y).any().item())
            print(f"B{b} T{T} F{F} {ci:2d}->{co:2d} flip{flip}: fp32 mfma err/|a||b| {out[0][0]:.2e} (sum {out[0][1]:.1e})   split {out[3][0]:.2e} (sum {out[3][1]:.1e}) nan={out[3][2]}"
                  f"   signed mean/std of err/|a||b|: fp32 {bias[0][0]:+.2e}/{bias[0][1]:.2e}  split {bias[3][0]:+.2e}/{bias[3][1]:.2e}")
    T, F = 1201, 480
    for ci, co, flip, what in ((20, 20, 0, "conv2 fwd"), (20, 40, 0, "conv3 fwd"), (40, 40, 0, "conv4 fwd"), (40, 40, 1, "conv4 dgrad"), (40, 20, 1, "conv3 dgrad")):
        x = torch.randn(B, T, ci, F, device=dev)
        y = torch.empty(B, T, co, F, device=dev)
        w = torch.randn((ci, co, 3, 3) if flip else (co, ci, 3, 3), device=dev) * 0.05
        scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        nblk = L.a2s_conv3x3_stat_blocks(B, T, F, ci)
        partial = torch.empty(nblk, co, 2, device=dev)
        cws = hip.conv_workspace(ci, dev)
        line = f"{what:12s} {ci:2d}->{co:2d}"
        for mode in (0, 3):
            L.a2s_debug_set(b"conv_bf16x3", mode)
            ms = timed(lambda: hip.check(L.a2s_conv3x3(hip.stream(), hip._p(x), hip._p(w), hip._p(y), hip._p(None if flip else scale), hip._p(None if flip else shift),
                                                       hip._p(None if flip else partial), B, T, F, ci, co, flip, hip._p(cws)), "conv"))
            fl = 2.0 * 9 * ci * co * B * T * F
            line += f"   {'split' if mode else 'fp32 '} {ms:8.2f} ms {fl / ms / 1e9:6.1f} TFLOP/s"
        print(line)
        del x, y
    L.a2s_debug_set(b"conv_bf16x3", 3)


if __name__ == "__main__":
    main()
