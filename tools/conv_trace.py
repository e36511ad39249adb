"""Phase timeline of the split-operand convolution kernel from in-kernel shader-clock stamps (library built with -DC4_TRACE):
per stage of wave 0 of 8 mid-grid workgroups -- barrier waits, staging (commit), input issue, multiply, epilogue.
usage: A2S_LIB=.../liba2s_hip_trace.so python tools/conv_trace.py [B] [Cin] [Cout]"""
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
    T, F = 1201, 480
    dev = torch.device("cuda:0")
    L = hip.lib()
    x = torch.randn(B, T, ci, F, device=dev)
    y = torch.empty(B, T, co, F, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    nblk = L.a2s_conv3x3_stat_blocks(B, T, F, ci)
    partial = torch.empty(nblk, co, 2, device=dev)
    cws = hip.conv_workspace(ci, dev)
    wgrad = os.environ.get("A2S_TRACE_WGRAD") == "1"
    if wgrad:                                            # weight-gradient kernel instead (stamps of rounds 100..123 of 8 workgroups)
        dy = torch.randn(B, T, co, F, device=dev) * 1e-4
        dW = torch.zeros(co, ci, 3, 3, device=dev)
        nbytes = L.a2s_conv3x3_wgrad_workspace_bytes(ci, co)
        ws = torch.empty(nbytes // 4, device=dev)
        amax = hip.absmax(dy)
        for _ in range(2):
            hip.check(L.a2s_conv3x3_wgrad_scaled(hip.stream(), hip._p(dy), hip._p(x), hip._p(scale), hip._p(shift), hip._p(dW), hip._p(ws),
                                                 C.c_size_t(nbytes), B, T, F, ci, co, hip._p(amax)), "wgrad")
    for _ in range(0 if wgrad else 2):
        hip.check(L.a2s_conv3x3(hip.stream(), hip._p(x), hip._p(w), hip._p(y), hip._p(scale), hip._p(shift), hip._p(partial), B, T, F, ci, co, 0, hip._p(cws)), "conv")
    torch.cuda.synchronize()
    buf = np.zeros(8 * 24 * 8, dtype=np.uint64)
    rc = L.a2s_conv_trace_read(buf.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    t = buf.reshape(8, 24, 8).astype(np.int64)
    names = ["barrier1 wait", "commit (+weight DMA issue)", "barrier2 wait", "issue next loads", "multiply (issue)", "epilogue"]
    if wgrad:
        names[1], names[5] = "commit", "(unused)"
    tot = np.zeros(6)
    for wg in range(8):
        for q in range(24):
            s = t[wg, q]
            d = [s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], (s[6] - s[5]) if s[6] > s[5] else 0]
            tot += np.array(d, dtype=float)
    per = tot / (8 * 24)
    for n, v in zip(names, per):
        print(f"{n:28s} {v:9.0f} clocks per stage")
    whole = (t[:, -1, 5] - t[:, 0, 0]).mean() / 24
    print(f"stage period {whole:.0f} clocks (sum of phases {per.sum():.0f})")
    print("one workgroup, stamps relative to its first:", (t[0, :6, :7] - t[0, 0, 0]).tolist())


if __name__ == "__main__":
    main()
