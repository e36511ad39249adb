"""Time per decode step of a few-clip note-decoder call: persistent launch vs launch-per-step (forward, training mode, teacher forced).
usage: python tools/dec_persist_bench.py [clips] [bars] [steps]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import engine, hip, spec  # noqa: E402


def main():
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    bars = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 398
    dev = torch.device("cuda:0")
    cfg = spec.default_cfg()
    st = spec.procedural_state(cfg, 5, eos_bias=0.0, lively="token")
    S = {k: v.to(dev) for k, v in st.items()}
    T, H = 1201, 256
    R = clips * bars
    eng = engine.Engine(cfg)
    enc = torch.tanh(torch.randn(clips, T, 2 * H, device=dev))
    keys = eng._keys(S, "decoder.upper_decoder.attn", enc.view(clips * T, 2 * H), H).view(clips, T, H)
    h0 = torch.tanh(torch.randn(R, 2 * H, device=dev))
    U = cfg["max_length"][0]
    gt = torch.randint(0, 140, (R, U), device=dev)
    probs = torch.zeros(R, U, 173, device=dev)
    until = torch.full((R,), steps, dtype=torch.int32)
    n = steps
    import ctypes
    active = dict(until=until.to(dev), order=torch.arange(clips, dtype=torch.int32, device=dev), rank=torch.arange(clips, dtype=torch.int32, device=dev),
                  n_active=(ctypes.c_int * n)(*([clips] * n)), n_clips=clips, m_active=None,
                  row_list=torch.arange(R, dtype=torch.int32, device=dev), n_rows_active=(ctypes.c_int * n)(*([R] * n)))
    flags = [(1 << bars) - 1] * n
    flags_dev = torch.tensor(flags, dtype=torch.int32, device=dev)
    attn_ws = hip.attn_workspace(clips, T, H, dev, groups=bars)
    gemm_ws = hip.gemm_workspace(R, dev)
    for persist in (False, True, False, True):
        os.environ["A2S_DEC_PERSIST"] = "1" if persist else "0"
        hip.check(hip.lib().a2s_debug_set(b"dec_persist", 1 if persist else 0), "set")
        ts = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            ids, lengths, sv = eng._decode_staff(S, "decoder.upper_decoder", keys, enc, h0, U, probs, gt, n, flags, True, 0.0, R, T, attn_ws, gemm_ws, active, None,
                                                 flags_dev, persist)
            torch.cuda.synchronize()
            ts.append(time.time() - t0)
        ws = sv.get("persist_ws")
        extra = ""
        if ws is not None:
            w = ws[:8].view(torch.int32)
            extra = f"  [abort word {int(w[0])}, workgroups on their clip's XCD {int(w[1])}]"
        print(f"persist={persist}: {min(ts) / n * 1e6:7.1f} us per step ({clips} clips x {bars} bars, {n} steps){extra}  finite={bool(torch.isfinite(sv['h']).all())}")


if __name__ == "__main__":
    main()
