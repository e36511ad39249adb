"""Phase timeline of the row-streaming convolution from in-kernel shader-clock stamps (variant build -DRW_TRACE):
  python -c "from piano_a2s_amd import build; print(build.build_variant('rwtrace', ['-DRW_TRACE'], ['a2s_conv_rows.hip']))"
  A2S_LIB=piano_a2s_amd/csrc/_obj/liba2s_hip_rwtrace.so python tools/conv_rows_trace.py [B] [Cin] [Cout] [flip]
Stamps of lane 0 of one even-row wave and one odd-row wave of 8 mid-grid workgroups, iterations 20..51 of their strip."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ci = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    co = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    flip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    T, F = 1201, 480
    dev = torch.device("cuda:0")
    L = hip.lib()
    x = torch.randn(B, T, ci, F, device=dev) * (1e-4 if flip else 1.0)
    y = torch.empty(B, T, co, F, device=dev)
    w = torch.randn((ci, co, 3, 3) if flip else (co, ci, 3, 3), device=dev) * 0.05
    scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    partial = torch.empty(L.a2s_conv3x3_stat_blocks(B, T, F, ci), co, 2, device=dev)
    cws = hip.conv_workspace(ci, dev)
    in_amax = x.abs().amax(dim=(0, 1, 3)).contiguous()
    out_amax = torch.zeros(co, device=dev)
    xmax = hip.absmax(x)
    yl = torch.randn(B, T, co, F, device=dev)
    bn = [torch.randn(co, device=dev) * 0.1, torch.rand(co, device=dev) + 0.5, torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1]
    for _ in range(2):
        if flip:
            hip.check(L.a2s_conv3x3_dgrad_bnstats_scaled(hip.stream(), hip._p(x), hip._p(w), hip._p(y), hip._p(yl), hip._p(bn[0]), hip._p(bn[1]), hip._p(bn[2]), hip._p(bn[3]),
                                                         hip._p(partial), B, T, F, ci, co, hip._p(cws), hip._p(xmax)), "dgrad")
        else:
            hip.conv3x3_forward(x, w, y, scale, shift, partial, cws, in_amax, out_amax)
    torch.cuda.synchronize()
    n_wg, n_it = 8, 32
    buf = np.zeros(n_wg * 2 * n_it * 8, dtype=np.uint64)
    assert L.a2s_rows_trace_read(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.reshape(n_wg, 2, n_it, 8).astype(np.int64)
    if os.environ.get("ROWS16", "1") == "1" and co == 40:
        tt = t[:, :, :, :5]
        for mh in (0, 1):
            d = np.diff(tt[:, mh], axis=-1).reshape(-1, 4).mean(0)
            period = np.diff(tt[:, mh, :, 0], axis=-1).mean()
            print(f"rows16, column half {mh} of channel group 0: row period {period:.0f} clocks (1 output row x 120 columns)")
            for n, v in zip(["multiply (+ conversion of row t+2)", "barrier wait", "epilogue", "issue next row's loads"], d):
                print(f"    {n:40s} {v:8.0f}")
        return
    names = ["M: multiply (even-row waves also convert row t+3)", "barrier wait", "odd-row waves: convert row t+4", "E: epilogue", "issue next row loads, advance", "barrier wait", "(loop)"]
    for rp, who in ((0, "even-row wave"), (1, "odd-row wave")):
        tt = t[:, rp, :, :7]
        d = np.concatenate([np.diff(tt, axis=-1), (t[:, rp, 1:, 0:1] - tt[:, :-1, 6:7]).mean(1, keepdims=True).repeat(tt.shape[1], 1)[..., None].reshape(tt.shape[0], tt.shape[1], 1)], axis=-1).reshape(-1, 7).mean(0)
        period = np.diff(t[:, rp, :, 0], axis=-1).mean()
        print(f"{who}: iteration period {period:.0f} clocks (2 output rows x 120 columns)")
        for n, v in zip(names, d):
            print(f"    {n:55s} {v:8.0f}")
    print("one workgroup, even-row wave, stamps relative to its first:", (t[0, 0, :3] - t[0, 0, 0, 0]).tolist())


if __name__ == "__main__":
    main()
