"""The row-streaming weight gradient (csrc/a2s_conv_wrows.hip) against float64 on small and ragged shapes, then its launch duration at
the training shapes next to round 2's kernels (`wgrad_rows` = 0).  usage: python tools/wgrad_rows_check.py [B] [--no-check] [--no-time] | --trace (library built with -DWR_TRACE)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
L = hip.lib()
ROWS = 1


def check():
    worst = 0.0
    for (Cin, Cout) in ((40, 40), (20, 40), (20, 20), (40, 20)):
        for (B, T, F) in ((2, 9, 24), (1, 41, 480), (3, 37, 100), (1, 1, 132), (2, 70, 256), (5, 3, 128), (300, 2, 260)):
            g = torch.Generator().manual_seed(Cin * 100 + Cout + T)
            x = torch.randn(B, T, Cin, F, generator=g) * torch.exp(torch.randn(B, T, Cin, F, generator=g))
            dy = 1e-4 * torch.randn(B, T, Cout, F, generator=g) * torch.exp(torch.randn(B, T, Cout, F, generator=g))
            for affine in (True, False):
                scale, shift = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
                if affine:
                    a64 = torch.relu(x.double().permute(0, 2, 1, 3) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
                else:
                    a64 = x.double().permute(0, 2, 1, 3)
                g64 = dy.double().permute(0, 2, 1, 3)
                ref = torch.nn.grad.conv2d_weight(a64, (Cout, Cin, 3, 3), g64, padding=1)
                mag = torch.nn.grad.conv2d_weight(a64.abs(), (Cout, Cin, 3, 3), g64.abs(), padding=1) + 1e-300
                errs = []
                for rows in (ROWS, 0):
                    hip.check(L.a2s_debug_set(b"wgrad_rows", rows), "set")
                    if affine:
                        dW = hip.conv3x3_wgrad_for_test(dy.to(dev), x.to(dev), scale.to(dev), shift.to(dev))
                    else:
                        dW = torch.zeros(Cout, Cin, 3, 3, device=dev)
                        ws = torch.empty(L.a2s_conv3x3_wgrad_workspace_bytes(Cin, Cout) // 4, device=dev)
                        xd, dyd = x.to(dev), dy.to(dev)
                        hip.conv3x3_wgrad(dyd, xd, None, None, dW, ws, hip.absmax(dyd), None)
                        torch.cuda.synchronize()
                    dW = dW.cpu().double()
                    errs.append(float(((dW - ref).abs() / mag).nan_to_num(1e9).max()))
                print(f"wgrad {Cin}->{Cout} B{B} T{T} F{F} affine{int(affine)}: rows {errs[0]:.2e}   round-2 kernel {errs[1]:.2e}", flush=True)
                worst = max(worst, errs[0])
    hip.check(L.a2s_debug_set(b"wgrad_rows", ROWS), "set")
    print("WORST", f"{worst:.3e}", "OK" if worst < 2e-6 else "FAIL", flush=True)


def timed(fn, iters=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench(B):
    T, F = 1201, 480
    for ci, co, what in ((40, 40, "conv4 wgrad"), (20, 40, "conv3 wgrad"), (20, 20, "conv2 wgrad")):
        x = torch.randn(B, T, ci, F, device=dev)
        dy = torch.randn(B, T, co, F, device=dev) * 1e-4
        scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        dW = torch.zeros(co, ci, 3, 3, device=dev)
        ws = torch.empty(L.a2s_conv3x3_wgrad_workspace_bytes(ci, co) // 4, device=dev)
        bound = hip.act_bound(scale, shift, x.abs().amax(dim=(0, 1, 3)).contiguous())
        dmax = hip.absmax(dy)
        res = {}
        for rows in (0, ROWS):
            hip.check(L.a2s_debug_set(b"wgrad_rows", rows), "set")
            res[rows] = timed(lambda: hip.conv3x3_wgrad(dy, x, scale, shift, dW, ws, dmax, bound))
        hip.check(L.a2s_debug_set(b"wgrad_rows", ROWS), "set")
        fl = 2.0 * 9 * ci * co * B * T * F
        gb = 4.0 * B * T * F * (ci + co)
        print(f"{what:12s} {ci:2d}->{co:2d} B={B}: round 2 {res[0]:7.2f} ms   rows {res[ROWS]:7.2f} ms = {fl / res[ROWS] / 1e9:6.1f} TFLOP/s (x3 products on the pipe), "
              f"{gb / res[ROWS] / 1e6:5.0f} GB/s algorithmic", flush=True)
        del x, dy


def trace(B):
    """(library built with -DWR_TRACE) share of its time each role's lead wave waits in the per-row barrier"""
    T, F = 1201, 480
    for ci, co in ((40, 40), (20, 40), (20, 20)):
        x = torch.randn(B, T, ci, F, device=dev)
        dy = torch.randn(B, T, co, F, device=dev) * 1e-4
        scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        dW = torch.zeros(co, ci, 3, 3, device=dev)
        ws = torch.zeros(L.a2s_conv3x3_wgrad_workspace_bytes(ci, co) // 4, device=dev)
        hip.check(L.a2s_debug_set(b"wgrad_rows", 1), "set")
        hip.conv3x3_wgrad(dy, x, scale, shift, dW, ws, hip.absmax(dy), hip.act_bound(scale, shift, x.abs().amax(dim=(0, 1, 3)).contiguous()))
        torch.cuda.synchronize()
        n = min(256, B * 4) * co * ci * 9
        tr = ws[n:n + 256 * 8].view(torch.int64).view(256, 2, 2).double().cpu()
        print(f"{ci}->{co}: multiply wave waits {100 * float((tr[:, 0, 0] / tr[:, 0, 1]).mean()):.1f} % of {float(tr[:, 0, 1].mean()) / 1201:.0f} ticks per row; "
              f"staging wave waits {100 * float((tr[:, 1, 0] / tr[:, 1, 1]).mean()):.1f} %", flush=True)


if __name__ == "__main__":
    if "--trace" in sys.argv:
        trace(64)
        sys.exit(0)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--no-check" not in sys.argv:
        check()
    if "--no-time" not in sys.argv:
        bench(int(args[0]) if args else 64)
