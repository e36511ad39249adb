"""Per-step cost of the encoder recurrences (a2s_gru_seq_fwd / a2s_gru_seq_bwd), one direction alone and two concurrently.
usage: python tools/gru_step_bench.py [B] [T]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    H = 256
    dev = torch.device("cuda:0")
    L = hip.lib()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = []
    for d in range(2):
        bufs.append(dict(gi=torch.randn(B, T, 3 * H, device=dev) * 0.5, w=torch.randn(3 * H, H, device=dev) * 0.06, b=torch.zeros(3 * H, device=dev),
                         out=torch.empty(B, T, 2 * H, device=dev), hbuf=torch.empty(2, B, H, device=dev), gh=torch.empty(B, 3 * H, device=dev),
                         gates=torch.empty(T, B, 4 * H, device=dev), hn=torch.empty(B, H, device=dev), ws=hip.gemm_workspace(B, dev),
                         dout=torch.randn(B, T, 2 * H, device=dev) * 0.01, dgi=torch.empty(B, T, 3 * H, device=dev), dghs=torch.empty(B, T, 3 * H, device=dev),
                         dgh_first=torch.empty(B, 3 * H, device=dev), dhbuf=torch.empty(2, B, H, device=dev), dgh_tmp=torch.empty(B, 3 * H, device=dev)))

    def fwd(d):
        x = bufs[d]
        hip.check(L.a2s_gru_seq_fwd(hip.stream(), hip._p(x["gi"]), C.c_long(T * 3 * H), C.c_long(3 * H), hip._p(x["w"]), hip._p(x["b"]),
                                    C.c_void_p(x["out"].data_ptr() + 4 * d * H), C.c_long(T * 2 * H), C.c_long(2 * H), hip._p(x["hbuf"]), hip._p(x["gh"]),
                                    hip._p(x["gates"]), hip._p(x["hn"]), B, T, H, d, hip._p(x["ws"]), C.c_size_t(x["ws"].numel() * 4)), "fwd")

    def bwd(d):
        x = bufs[d]
        hip.check(L.a2s_gru_seq_bwd(hip.stream(), C.c_void_p(x["dout"].data_ptr() + 4 * d * H), C.c_long(T * 2 * H), C.c_long(2 * H),
                                    C.c_void_p(x["out"].data_ptr() + 4 * d * H), C.c_long(T * 2 * H), C.c_long(2 * H), hip._p(x["gates"]), hip._p(x["w"]),
                                    C.c_void_p(0), hip._p(x["dgi"]), hip._p(x["dghs"]), hip._p(x["dgh_first"]), hip._p(x["dhbuf"]), hip._p(x["dgh_tmp"]),
                                    B, T, H, d, hip._p(x["ws"]), C.c_size_t(x["ws"].numel() * 4)), "bwd")

    for fused in (0, 1, 2):                      # 0: three launches per step, 1: one fused launch per step, 2: one persistent launch for all steps
        hip.check(L.a2s_debug_set(b"gru_fused", 1 if fused else 0), "set")
        hip.check(L.a2s_debug_set(b"gru_persist", 1 if fused == 2 else 0), "set")
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            for conc in (1, 2):
                for _ in range(2):
                    torch.cuda.synchronize()
                    t0 = time.time()
                    for d in range(conc):
                        with torch.cuda.stream(streams[d]):
                            fn(d)
                    torch.cuda.synchronize()
                    dt = time.time() - t0
                mode = ""
                if fused == 2:
                    w = bufs[0]["ws"].view(torch.int32)
                    off = 0 if name == "fwd" else 3 * H * H
                    mode = f"  [workgroups exchanging through their XCD's L2: {int(w[off + 1])}, abort word {int(w[off])}]"
                print(f"fused={fused} {name} directions={conc}: {dt / T * 1e6:6.1f} us per step (B={B}, T={T}){mode}")


if __name__ == "__main__":
    main()
