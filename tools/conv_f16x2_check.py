"""The two-term fp16 convolution path (conv3x3_split<.., 2>) against float64, next to the three-term bf16 path and the fp32-input MFMA
kernel: accuracy relative to sum |a||b| (forward with BatchNorm+ReLU on load, data gradient with tiny gradient magnitudes and the
power-of-two operand scale), error of the batch-statistics sums, and launch durations at the training shapes.
usage: python tools/conv_f16x2_check.py [B]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as Fn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402

MODES = (("fp32", 0, 0), ("bf16x3", 3, 0), ("f16x2", 3, 3))


def set_mode(L, bf, f16):
    L.a2s_debug_set(b"conv_bf16x3", bf)
    L.a2s_debug_set(b"conv_f16x2", f16)


def conv(L, x, w, scale, shift, flip, co, amax=None, yl=None, bn=None):
    B, T, ci, F = x.shape
    dev = x.device
    y = torch.full((B, T, co, F), float("nan"), device=dev)
    nblk = L.a2s_conv3x3_stat_blocks(B, T, F, ci)
    partial = torch.zeros(nblk, co, 2, device=dev)
    cws = hip.conv_workspace(ci, dev)
    if flip:
        hip.check(L.a2s_conv3x3_dgrad_bnstats_scaled(hip.stream(), hip._p(x), hip._p(w), hip._p(y), hip._p(yl), hip._p(bn[0]), hip._p(bn[1]), hip._p(bn[2]),
                                                     hip._p(bn[3]), hip._p(partial), B, T, F, ci, co, hip._p(cws), hip._p(amax)), "dgrad")
    else:
        hip.check(L.a2s_conv3x3(hip.stream(), hip._p(x), hip._p(w), hip._p(y), hip._p(scale), hip._p(shift), hip._p(partial), B, T, F, ci, co, 0, hip._p(cws)), "conv")
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
    for (b, T, F) in ((2, 37, 100), (2, 64, 480)):
        for ci, co, flip, gmag in ((20, 20, 0, 1.0), (20, 40, 0, 1.0), (40, 40, 0, 1.0), (40, 40, 1, 1e-6), (40, 20, 1, 3e-4), (20, 20, 1, 1e-9)):
            x = gmag * torch.randn(b, T, ci, F, device=dev) * torch.exp(torch.randn(b, T, ci, F, device=dev))      # heavy-tailed magnitudes
            w = torch.randn((ci, co, 3, 3) if flip else (co, ci, 3, 3), device=dev) * 0.05
            scale = None if flip else torch.rand(ci, device=dev) + 0.5
            shift = None if flip else torch.randn(ci, device=dev) * 0.1
            yl = torch.randn(b, T, co, F, device=dev) if flip else None
            bn = [torch.randn(co, device=dev) * 0.1, torch.rand(co, device=dev) + 0.5, torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1] if flip else None
            amax = x.abs().max().reshape(1).float() if flip else None
            ref = reference(x, w, scale, shift, flip)
            mag = reference(x.abs() if flip else x, w.abs(), scale, shift, flip).abs() + 1e-300
            line = f"B{b} T{T} F{F} {ci:2d}->{co:2d} {'dgrad' if flip else 'fwd  '} |x|~{gmag:.0e}:"
            for name, bf, f16 in MODES:
                set_mode(L, bf, f16)
                y, part = conv(L, x, w, scale, shift, flip, co, amax, yl, bn)
                torch.cuda.synchronize()
                rel = (y.double() - ref) / mag
                s = part.double().sum(0)
                serr = ((s[:, 0] - ref.sum((0, 1, 3))).abs() / mag.sum((0, 1, 3))).max().item() if not flip else float("nan")
                line += f"  {name} max {rel.abs().max().item():.2e} mean {rel.mean().item():+.1e} std {rel.std().item():.1e} sum {serr:.1e}{' NaN!' if torch.isnan(y).any() else ''}"
            print(line, flush=True)
    T, F = 1201, 480
    for ci, co, flip, what in ((20, 20, 0, "conv2 fwd"), (20, 40, 0, "conv3 fwd"), (40, 40, 0, "conv4 fwd"), (40, 40, 1, "conv4 dgrad"), (40, 20, 1, "conv3 dgrad"),
                               (20, 20, 1, "conv2 dgrad")):
        x = torch.randn(B, T, ci, F, device=dev)
        w = torch.randn((ci, co, 3, 3) if flip else (co, ci, 3, 3), device=dev) * 0.05
        scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        yl = torch.randn(B, T, co, F, device=dev) if flip else None
        bn = [torch.randn(co, device=dev) * 0.1, torch.rand(co, device=dev) + 0.5, torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1] if flip else None
        amax = x.abs().max().reshape(1).float() if flip else None
        line = f"{what:12s} {ci:2d}->{co:2d}"
        flops = 2.0 * 9 * ci * co * B * T * F
        for name, bf, f16 in MODES:
            set_mode(L, bf, f16)
            ms = timed(lambda: conv(L, x, w, None if flip else scale, None if flip else shift, flip, co, amax, yl, bn))
            line += f"   {name} {ms:7.2f} ms {flops / ms / 1e9:6.1f} TFLOP/s"
        print(line, flush=True)
    set_mode(L, 3, 3)


if __name__ == "__main__":
    main()


def wgrad_check(B):
    """Weight gradient: fp32-input MFMA kernel / three-term bf16 / two-term fp16 against float64, then durations at the training shapes."""
    dev = torch.device("cuda:0")
    L = hip.lib()
    torch.manual_seed(1)

    def run(dy, x, scale, shift, ci, co, amax):
        b, T, _, F = x.shape
        dW = torch.zeros(co, ci, 3, 3, device=dev)
        nb = L.a2s_conv3x3_wgrad_workspace_bytes(ci, co)
        ws = torch.empty(nb // 4, device=dev)
        hip.check(L.a2s_conv3x3_wgrad_scaled(hip.stream(), hip._p(dy), hip._p(x), hip._p(scale), hip._p(shift), hip._p(dW), hip._p(ws), C.c_size_t(nb),
                                             b, T, F, ci, co, hip._p(amax)), "wgrad")
        return dW

    modes = (("fp32", 0, 0), ("bf16x3", 2, 0), ("f16x2", 2, 2))
    for (b, T, F, ci, co, gmag) in ((2, 37, 100, 20, 20, 1e-6), (2, 64, 480, 20, 40, 3e-4), (2, 64, 480, 40, 40, 1e-8)):
        x = torch.randn(b, T, ci, F, device=dev)
        dy = gmag * torch.randn(b, T, co, F, device=dev) * torch.exp(torch.randn(b, T, co, F, device=dev))
        scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        act = torch.relu(x.double() * scale.double()[None, None, :, None] + shift.double()[None, None, :, None]).permute(0, 2, 1, 3).cpu()
        g = dy.double().permute(0, 2, 1, 3).cpu()
        ref = torch.nn.grad.conv2d_weight(act, (co, ci, 3, 3), g, padding=1)
        mag = torch.nn.grad.conv2d_weight(act.abs(), (co, ci, 3, 3), g.abs(), padding=1) + 1e-300
        amax = dy.abs().max().reshape(1).float()
        line = f"wgrad B{b} T{T} F{F} {ci:2d}->{co:2d} |dy|~{gmag:.0e}:"
        for name, sp, f16 in modes:
            L.a2s_debug_set(b"wgrad_bf16x3", sp)
            L.a2s_debug_set(b"wgrad_f16x2", f16)
            dW = run(dy, x, scale, shift, ci, co, amax)
            torch.cuda.synchronize()
            rel = (dW.double().cpu() - ref) / mag
            line += f"  {name} max {rel.abs().max().item():.2e} mean {rel.mean().item():+.1e}{' NaN!' if torch.isnan(dW).any() else ''}"
        print(line, flush=True)
    T, F = 1201, 480
    for ci, co, what in ((20, 20, "conv2 wgrad"), (20, 40, "conv3 wgrad"), (40, 40, "conv4 wgrad")):
        x = torch.randn(B, T, ci, F, device=dev)
        dy = 1e-5 * torch.randn(B, T, co, F, device=dev)
        scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        amax = dy.abs().max().reshape(1).float()
        flops = 2.0 * 9 * ci * co * B * T * F
        line = f"{what:12s} {ci:2d}->{co:2d}"
        for name, sp, f16 in modes:
            L.a2s_debug_set(b"wgrad_bf16x3", sp)
            L.a2s_debug_set(b"wgrad_f16x2", f16)
            ms = timed(lambda: run(dy, x, scale, shift, ci, co, amax))
            line += f"   {name} {ms:7.2f} ms {flops / ms / 1e9:6.1f} TFLOP/s"
        print(line, flush=True)
    L.a2s_debug_set(b"wgrad_bf16x3", 1)
    L.a2s_debug_set(b"wgrad_f16x2", 1)


if __name__ == "__main__" and os.environ.get("A2S_CHECK_WGRAD", "1") == "1":
    wgrad_check(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
