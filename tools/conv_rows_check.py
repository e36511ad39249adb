"""The row-streaming convolution (csrc/a2s_conv_rows.hip) against float64 on small shapes (forward with operand BatchNorm + batch statistics,
data gradient with the BatchNorm-backward statistics epilogue, plain data gradient), then its launch durations at the training shapes next to
the tiled kernels of round 2 (`conv_rows` = 0).  usage: python tools/conv_rows_check.py [B] [--no-check] [--no-time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
L = hip.lib()
ROWS = int(os.environ.get("ROWS", "7"))          # conv_rows switch: 1 = first generation everywhere, 3 = rows16 where it exists, 7 / 15 = + its two-accumulator-set form


def fwd(x, w, scale, shift, rows):
    hip.check(L.a2s_debug_set(b"conv_rows", rows), "set")
    B, T, Cin, F = x.shape
    Cout = w.shape[0]
    y = torch.full((B, T, Cout, F), float("nan"), device=dev)
    part = torch.zeros(L.a2s_conv3x3_stat_blocks(B, T, F, Cin), Cout, 2, device=dev)
    amax = torch.full((Cout,), -1.0, device=dev)
    cws = hip.conv_workspace(Cin, dev)
    hip.conv3x3_forward(x, w, y, scale, shift, part, cws, None, amax)
    torch.cuda.synchronize()
    return y, part.double().sum(0), amax


def dgrad(dy, w, yl, bn, rows, stats=True):
    hip.check(L.a2s_debug_set(b"conv_rows", rows), "set")
    B, T, Cout, F = dy.shape            # layer Cout = channels of dy; layer Cin = channels of dx
    Cin = w.shape[1]
    dx = torch.full((B, T, Cin, F), float("nan"), device=dev)
    amax = hip.absmax(dy)
    part = torch.zeros(L.a2s_conv3x3_stat_blocks(B, T, F, Cout), Cin, 2, device=dev)
    cws = hip.conv_workspace(Cout, dev)
    if stats:
        hip.check(L.a2s_conv3x3_dgrad_bnstats_scaled(hip.stream(), hip._p(dy), hip._p(w), hip._p(dx), hip._p(yl), hip._p(bn[0]), hip._p(bn[1]), hip._p(bn[2]),
                                                     hip._p(bn[3]), hip._p(part), B, T, F, Cout, Cin, hip._p(cws), hip._p(amax)), "dgrad")
    else:
        hip.check(L.a2s_conv3x3(hip.stream(), hip._p(dy), hip._p(w), hip._p(dx), None, None, None, B, T, F, Cout, Cin, 1, hip._p(cws)), "dgrad plain")
    torch.cuda.synchronize()
    return dx, part.double().sum(0)


def check():
    worst = 0.0
    for (Cin, Cout) in ((20, 20), (20, 40), (40, 40)):
        for (B, T, F) in ((2, 9, 24), (1, 41, 480), (3, 37, 100), (1, 6, 132), (2, 70, 256)):
            g = torch.Generator().manual_seed(Cin * 100 + Cout + T)
            x = torch.randn(B, T, Cin, F, generator=g) * torch.exp(torch.randn(B, T, Cin, F, generator=g))
            w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
            scale, shift = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
            a64 = torch.relu(x.double().permute(0, 2, 1, 3) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
            ref = torch.nn.functional.conv2d(a64, w.double(), padding=1).permute(0, 2, 1, 3)
            mag = torch.nn.functional.conv2d(a64.abs(), w.double().abs(), padding=1).permute(0, 2, 1, 3) + 1e-300
            y, sums, amax = fwd(x.to(dev), w.to(dev), scale.to(dev), shift.to(dev), ROWS)
            y = y.cpu().double()
            bad = int((~torch.isfinite(y)).sum())
            err = float(((y - ref).abs() / mag).nan_to_num(1e9).max())
            s_err = float((sums.cpu()[:, 0] - ref.sum(dim=(0, 1, 3))).abs().max() / mag.sum(dim=(0, 1, 3)).max())
            s2_err = float((sums.cpu()[:, 1] - (ref ** 2).sum(dim=(0, 1, 3))).abs().max() / (ref ** 2).sum(dim=(0, 1, 3)).max())
            a_err = float((amax.cpu().double() - ref.abs().amax(dim=(0, 1, 3))).abs().max() / ref.abs().max())
            print(f"fwd   {Cin}->{Cout} B{B} T{T} F{F}: err {err:.2e} (vs sum|a||b|)  nonfinite {bad}  sum {s_err:.1e}  sumsq {s2_err:.1e}  absmax {a_err:.1e}", flush=True)
            worst = max(worst, err, s_err * 10, s2_err * 0.01, a_err * 0.1)
    for (Cl_out, Cl_in) in ((40, 40), (40, 20), (20, 20)):          # layer (Cin = Cl_in -> Cout = Cl_out): dy has Cl_out channels
        for (B, T, F) in ((2, 9, 24), (1, 41, 480), (3, 37, 100), (2, 70, 256)):
            g = torch.Generator().manual_seed(Cl_out * 100 + Cl_in + T)
            dy = 1e-4 * torch.randn(B, T, Cl_out, F, generator=g) * torch.exp(torch.randn(B, T, Cl_out, F, generator=g))
            w = torch.randn(Cl_out, Cl_in, 3, 3, generator=g) * 0.2
            w64 = w.double().transpose(0, 1).flip(2, 3)
            ref = torch.nn.functional.conv2d(dy.double().permute(0, 2, 1, 3), w64, padding=1).permute(0, 2, 1, 3)
            mag = torch.nn.functional.conv2d(dy.double().abs().permute(0, 2, 1, 3), w64.abs(), padding=1).permute(0, 2, 1, 3) + 1e-300
            yl = torch.randn(B, T, Cl_in, F, generator=g)
            mean, invstd = torch.randn(Cl_in, generator=g) * 0.1, torch.rand(Cl_in, generator=g) + 0.5
            bsc, bsh = torch.rand(Cl_in, generator=g) + 0.5, torch.randn(Cl_in, generator=g) * 0.3
            bn = [t.to(dev) for t in (mean, invstd, bsc, bsh)]
            for stats in (True, False):
                dx, sums = dgrad(dy.to(dev), w.to(dev), yl.to(dev), bn, ROWS, stats)
                dx = dx.cpu().double()
                bad = int((~torch.isfinite(dx)).sum())
                err = float(((dx - ref).abs() / mag).nan_to_num(1e9).max())
                line = f"dgrad {Cl_in}<-{Cl_out} B{B} T{T} F{F} stats{int(stats)}: err {err:.2e}  nonfinite {bad}"
                if stats:
                    on = (yl.double() * bsc.double().view(1, 1, -1, 1) + bsh.double().view(1, 1, -1, 1)) > 0
                    gm = torch.where(on, ref, torch.zeros_like(ref))
                    xhat = (yl.double() - mean.double().view(1, 1, -1, 1)) * invstd.double().view(1, 1, -1, 1)
                    sc = float(torch.where(on, mag, torch.zeros_like(mag)).sum(dim=(0, 1, 3)).max())
                    e1 = float((sums.cpu()[:, 0] - gm.sum(dim=(0, 1, 3))).abs().max()) / sc
                    e2 = float((sums.cpu()[:, 1] - (gm * xhat).sum(dim=(0, 1, 3))).abs().max()) / (sc * float(xhat.abs().max()))
                    line += f"  stats {e1:.1e} {e2:.1e}"
                    worst = max(worst, e1 * 0.1, e2 * 0.1)
                print(line, flush=True)
                worst = max(worst, err)
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
    for ci, co, flip, what in ((20, 20, 0, "conv2 fwd"), (20, 40, 0, "conv3 fwd"), (40, 40, 0, "conv4 fwd"), (40, 40, 1, "conv4 dgrad"), (40, 20, 1, "conv3 dgrad"), (20, 20, 1, "conv2 dgrad")):
        x = torch.randn(B, T, ci, F, device=dev) * (1e-4 if flip else 1.0)
        y = torch.empty(B, T, co, F, device=dev)
        w = torch.randn((ci, co, 3, 3) if flip else (co, ci, 3, 3), device=dev) * 0.05
        scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        cws = hip.conv_workspace(ci, dev)
        in_amax = x.abs().amax(dim=(0, 1, 3)).contiguous()
        out_amax = torch.zeros(co, device=dev)
        xmax = hip.absmax(x)
        yl = torch.randn(B, T, co, F, device=dev) if flip else None
        bn = [torch.randn(co, device=dev) * 0.1, torch.rand(co, device=dev) + 0.5, torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1]
        res = {}
        for rows in (0, 1, 3, 7, 15):         # 7 / 15: rows16 with two accumulator sets for the forward / forward + data-gradient launches (round 6)
            hip.check(L.a2s_debug_set(b"conv_rows", rows), "set")
            nblk = L.a2s_conv3x3_stat_blocks(B, T, F, ci)
            partial = torch.empty(nblk, co, 2, device=dev)
            if flip:
                fn = lambda: hip.check(L.a2s_conv3x3_dgrad_bnstats_scaled(hip.stream(), hip._p(x), hip._p(w), hip._p(y), hip._p(yl), hip._p(bn[0]), hip._p(bn[1]),
                                                                          hip._p(bn[2]), hip._p(bn[3]), hip._p(partial), B, T, F, ci, co, hip._p(cws), hip._p(xmax)), "dgrad")
            elif rows:
                fn = lambda: hip.conv3x3_forward(x, w, y, scale, shift, partial, cws, in_amax, out_amax)
            else:
                fn = lambda: hip.check(L.a2s_conv3x3(hip.stream(), hip._p(x), hip._p(w), hip._p(y), hip._p(scale), hip._p(shift), hip._p(partial), B, T, F, ci, co, 0,
                                                     hip._p(cws)), "conv")
            res[rows] = timed(fn)
        hip.check(L.a2s_debug_set(b"conv_rows", ROWS), "set")
        fl = 2.0 * 9 * ci * co * B * T * F
        gb = 4.0 * B * T * F * (ci + co + (co if flip else 0))
        best = min(res[3], res[7], res[15])
        print(f"{what:12s} {ci:2d}->{co:2d} B={B}: tiled {res[0]:7.2f} ms   rows {res[1]:7.2f} ms   rows16 {res[3]:7.2f} ms   two sets fwd {res[7]:7.2f} / fwd+dgrad {res[15]:7.2f} ms"
              f" = {fl / best / 1e9:6.1f} TFLOP/s, {gb / best / 1e6:5.0f} GB/s algorithmic", flush=True)
        del x, y, yl


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--no-check" not in sys.argv:
        check()
    if "--no-time" not in sys.argv:
        bench(int(args[0]) if args else 64)
