"""First divergence between the persistent note decoder and the launch-per-step path (per saved tensor and step)."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_dec_persist import _cfg, _forward
from piano_a2s_amd import spec, synthetic

dev = torch.device("cuda:0")
B, frames, tf = int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 97, float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
cfg = _cfg()
st = spec.procedural_state(cfg, 40 + B, eos_bias=2.0, lively="token")
S = {k: v.to(dev) for k, v in st.items()}
batch = synthetic.make_batch(B, cfg, 7 + B, frames=frames, upper_range=(5, 30), lower_range=(3, 18), full_tail=0.15, spectrogram="ridges")
o0, c0, _ = _forward(cfg, {k: v.clone() for k, v in S.items()}, batch, dev, False, tf, 3)
o1, c1, e1 = _forward(cfg, {k: v.clone() for k, v in S.items()}, batch, dev, True, tf, 3)
for ci, (a, b) in enumerate(zip(c0, c1)):
    n = a["steps"]
    print(f"call {ci}: steps {n} persistent used {b['used']}")
    if b.get("used"):
        pass
    for name in ("q", "attw", "x", "o", "h", "gates"):
        ta, tb = a[name], b[name]
        for s in range(min(n, ta.shape[0])):
            d = (ta[s] - tb[s]).abs()
            bad = ~torch.isfinite(tb[s])
            if bad.any() or float(d.max()) > 1e-4 * max(1.0, float(ta[s].abs().max())):
                idx = torch.nonzero(bad if bad.any() else d == d.max())[0].tolist()
                print(f"   {name}: first divergence at step {s}: max diff {float(d[~bad].max()) if (~bad).any() else float('nan'):.3e}, non-finite {int(bad.sum())}, at {idx}: ref {float(ta[s][tuple(idx)]):.5f} got {float(tb[s][tuple(idx)]):.5f}; "
                      f"rows with diff: {torch.nonzero(d.amax(dim=-1) > 1e-4).flatten().tolist()[:10]}")
                break
        else:
            print(f"   {name}: equal over {min(n, ta.shape[0])} steps")
    xa, xb = a["x"][:n + 1], b["x"][:n + 1]
    d = (xa - xb).abs().amax(dim=-1)          # (steps+1, rows)
    bad = torch.nonzero(d > 1e-4)
    if bad.numel():
        print("   x mismatches (step,row):", bad[:12].tolist(), " token-part diff", float((xa[..., :16] - xb[..., :16]).abs().max()), " ctx-part diff", float((xa[..., 16:] - xb[..., 16:]).abs().max()))
        s0, r0 = bad[0].tolist()
        print("   ids around:", a["ids"][r0, max(0, s0 - 2):s0 + 2].tolist(), b["ids"][r0, max(0, s0 - 2):s0 + 2].tolist(), "steps", n)
a, b = c0[0], c1[0]
d = (a["x"][0] - b["x"][0]).abs()
for r in range(min(3, d.shape[0])):
    cols = torch.nonzero(d[r] > 1e-4).flatten().tolist()
    print("row", r, "bad cols", len(cols), cols[:40])
    if cols:
        print("   ref", [round(float(a["x"][0][r][c]), 4) for c in cols[:8]], "got", [float(b["x"][0][r][c]) for c in cols[:8]])
