"""Where the folded attention combine (dec_gru_step_cmb) first differs from the combine kernel: saved tensors of every decoder call, step by step."""
import random, sys, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from piano_a2s_amd import hip, spec, synthetic
from tests.test_gpu_dec_persist import _cfg, _forward
L = hip.lib()
dev = torch.device("cuda:0")
cfg = _cfg()
B, tf = 5, 0.7
st = spec.procedural_state(cfg, 40 + B, eos_bias=2.0, lively="token")
S = {k: v.to(dev) for k, v in st.items()}
batch = synthetic.make_batch(B, cfg, 7 + B, frames=301, upper_range=(5, 30), lower_range=(3, 18), full_tail=0.15, spectrogram="ridges")
res = []
for defer in (0, 1):
    hip.check(L.a2s_debug_set(b"attn_defer_combine", defer), "x")
    n0 = L.a2s_launch_count()
    import piano_a2s_amd.engine as E
    E.Engine.skip_finished_rows = True
    E.Engine.fuse_bars = True
    o, c, _ = _forward(cfg, {k: v.clone() for k, v in S.items()}, batch, dev, False, tf, 3)
    print("defer", defer, "launches", L.a2s_launch_count() - n0)
    res.append((o, c))
(o0, c0), (o1, c1) = res
for ci, (a, b) in enumerate(zip(c0, c1)):
    n = a["steps"]
    for s in range(n):
        bad = [(name, float((a[name][s] - b[name][s]).abs().max())) for name in ("x", "attw", "gates", "h", "o", "q") if not torch.equal(a[name][s], b[name][s])]
        if bad:
            print("call", ci, "steps", n, "rows", a["x"].shape[1], "first differing step", s, bad)
            d = (a["x"][s] - b["x"][s]).abs()
            print("   x rows differing:", d.amax(1).nonzero().flatten().tolist(), "cols range", d.amax(0).nonzero().flatten()[[0, -1]].tolist() if d.any() else None)
            break
    else:
        print("call", ci, "steps", n, "identical")
