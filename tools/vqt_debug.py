"""Stage-by-stage check of the GPU VQT against numpy (decimation chain, per-octave framed GEMMs)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip, vqt as P

dev = torch.device("cuda:0")
g = np.random.default_rng(3)
N = 16000 * 2
t = np.arange(N) / 16000.
y = 0.6 * np.sin(2 * np.pi * 440.0 * t) + 0.3 * np.sin(2 * np.pi * 1318.5 * t + 0.4) + 0.2 * np.sin(2 * np.pi * 55.0 * t) + 0.02 * g.standard_normal(N)
vq = P.VQT(dev)
h, half = P.decimation_filter()
cur = y.copy()
cur_d = torch.from_numpy(y.astype(np.float32)).to(dev).unsqueeze(0)
for i, o in enumerate(vq.octaves):
    n_fft, hop, nb = o["n_fft"], o["hop"], o["hi"] - o["lo"]
    frames = 1 + len(cur) // hop
    yp = np.concatenate([np.zeros(n_fft // 2), cur, np.zeros(n_fft)])
    idx = np.arange(n_fft)[None, :] + hop * np.arange(frames)[:, None]
    want = yp[idx] @ o["bank"].astype(float)                      # (frames, 2nb)
    plen = (frames - 1) * hop + n_fft
    plen = max(plen, n_fft // 2 + cur_d.shape[1]); plen += (-plen) % 4
    padded = torch.zeros((1, plen), dtype=torch.float32, device=dev)
    padded[:, n_fft // 2:n_fft // 2 + cur_d.shape[1]] = cur_d
    Cc = torch.zeros((1, frames, 2 * nb), dtype=torch.float32, device=dev)
    for part in (0, 1):
        hip.gemm(padded, hop, 1, o["bank_dev"], 2 * nb, 1, Cc, 2 * nb, frames, nb, n_fft, batch=1, bsA=plen, bsB=0, bsC=frames * 2 * nb, b_off=part * nb, c_off=part * nb)
    torch.cuda.synchronize()
    got = Cc[0].cpu().numpy()
    print(f"octave {i}: hop {hop} n_fft {n_fft} frames {frames}: rel err {np.abs(got - want).max() / np.abs(want).max():.3e}")
    if o.get("decimate_after") and i + 1 < len(vq.octaves):
        n_out = (len(cur) + 1) // 2
        nxt = np.convolve(cur, h)[half:half + 2 * n_out:2]
        d = vq._decimate(cur_d)
        torch.cuda.synchronize()
        print(f"   decimate {len(cur)} -> {n_out}: rel err {np.abs(d[0].cpu().numpy() - nxt).max() / np.abs(nxt).max():.3e}")
        cur, cur_d = nxt, torch.from_numpy(nxt.astype(np.float32)).to(dev).unsqueeze(0)

# ---- octave 0 in detail
o = vq.octaves[0]
n_fft, hop, nb = o["n_fft"], o["hop"], o["hi"] - o["lo"]
cur_d = torch.from_numpy(y.astype(np.float32)).to(dev).unsqueeze(0)
frames = 1 + len(y) // hop
plen = (frames - 1) * hop + n_fft
plen += (-plen) % 4
padded = torch.zeros((1, plen), dtype=torch.float32, device=dev)
padded[:, n_fft // 2:n_fft // 2 + len(y)] = cur_d
A = padded[0].unfold(0, n_fft, hop)[:frames]
print("bank dtype", o["bank"].dtype, o["bank_dev"].dtype, o["bank_dev"].shape, o["bank_dev"].is_contiguous(), "absmax", float(o["bank_dev"].abs().max()))
want_t = (A.double() @ o["bank_dev"].double()).cpu()
for part in (0, 1):
    Cc = torch.zeros((1, frames, 2 * nb), dtype=torch.float32, device=dev)
    hip.gemm(padded, hop, 1, o["bank_dev"], 2 * nb, 1, Cc, 2 * nb, frames, nb, n_fft, batch=1, bsA=plen, bsB=0, bsC=frames * 2 * nb, b_off=part * nb, c_off=part * nb)
    torch.cuda.synchronize()
    got = Cc[0].double().cpu()
    sl = slice(part * nb, (part + 1) * nb)
    print("part", part, "err vs torch", float((got[:, sl] - want_t[:, sl]).abs().max() / want_t.abs().max()), "got absmax", float(got.abs().max()), "want absmax", float(want_t.abs().max()))
    print("   got row0", got[0, sl][:4].tolist(), "want", want_t[0, sl][:4].tolist())
