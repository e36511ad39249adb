"""Isolated timing of the ConvStack BatchNorm-backward pass (statistics + apply) at a training shape: achieved HBM rate with and
without the max|dx| output.  usage: python tools/bn_bwd_bench.py [--batch 64] [--frames 501] [--channels 40]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=501)
    ap.add_argument("--channels", type=int, default=40)
    ap.add_argument("--bins", type=int, default=480)
    a = ap.parse_args()
    from piano_a2s_amd import hip
    L = hip.lib()
    dev = torch.device("cuda:0")
    rows, Cc, F = a.batch * a.frames, a.channels, a.bins
    g = torch.randn(rows, Cc, F, device=dev)
    x = torch.randn(rows, Cc, F, device=dev)
    mean, invstd = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    scale, shift = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    part = torch.empty(L.a2s_bn_bwd_partial_floats(C.c_long(rows), Cc, F), dtype=torch.float32, device=dev)
    c12 = torch.empty(2 * Cc, device=dev)
    amax = torch.zeros(1, device=dev)
    dx = torch.empty_like(g)
    nbytes = g.numel() * 4

    def run(with_amax):
        hip.check(L.a2s_bn_bwd_amax(hip.stream(), hip._p(g), hip._p(x), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), None, hip.f32(1.0),
                                    hip._p(dg), hip._p(db), hip._p(dx), hip._p(part), hip._p(c12), C.c_long(rows), Cc, F,
                                    hip._p(amax) if with_amax else None), "a2s_bn_bwd_amax")

    for with_amax in (False, True):
        for _ in range(3):
            run(with_amax)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(with_amax)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"rows {rows} C {Cc} F {F} amax={with_amax}: {ms:.2f} ms per pass (reduce + apply = 5 tensor passes, {5 * nbytes / 1e9:.1f} GB) "
              f"-> {5 * nbytes / ms / 1e6:.0f} GB/s")
    print("amax", float(amax), "reference", float(dx.abs().max()))


if __name__ == "__main__":
    main()
