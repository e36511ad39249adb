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

# the whole step: parameters after the update, folds off / forward fold / all folds
import models
from piano_a2s_amd import train
cfg2 = spec.default_cfg(freq_bins=48, max_length=(24, 14))
b2 = synthetic.make_batch(6, cfg2, 23, frames=97, upper_range=(4, 22), lower_range=(3, 12), full_tail=0.1)
db2 = [t.to(dev) if torch.is_tensor(t) else t for t in b2]
torch.manual_seed(9)
init = models.ScoreTranscription(**cfg2).state_dict()
flats = []
for fwd_fold, bwd_fold in ((0, 0), (0, 0), (1, 0), (1, 1), (1, 1)):
    hip.check(L.a2s_debug_set(b"attn_defer_combine", fwd_fold), "x")
    hip.check(L.a2s_debug_set(b"dec_bwd_fold", bwd_fold), "x")
    m = models.ScoreTranscription(**cfg2); m.load_state_dict(init); m = m.to(dev).train()
    st_ = train.TrainStep(m, dropout=False, clip_groups=False)
    st_.keep_grads = True
    n0 = L.a2s_launch_count()
    st_(db2, 0.6, rng=random.Random(5)); torch.cuda.synchronize()
    flats.append((fwd_fold, bwd_fold, L.a2s_launch_count() - n0, {k: v.clone() for k, v in st_.last_grads.items()}))
base = flats[0][3]
for f, b, n, g in flats[1:]:
    worst = max((float((g[k] - base[k]).abs().max() / (base[k].abs().max() + 1e-30)), k) for k in base)
    print(f"folds fwd={f} bwd={b}: launches {n} (baseline {flats[0][2]}); worst relative gradient difference to the baseline {worst[0]:.3e} ({worst[1]})")
