"""Launch durations of the staff-embedding recurrence (csrc/a2s_seq.hip staff_emb_fwd, csrc/a2s_bwd.hip staff_emb_bwd) at the shapes
the training step calls them with: the bulk clip group (248 rows, lengths U{20..120}) and the long-clip group (8 rows, one of 398).
usage: python tools/staff_emb_time.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piano_a2s_amd import hip, spec  # noqa: E402

dev = torch.device("cuda:0")
L = hip.lib()
NULL = C.c_void_p(0)
cfg = spec.default_cfg()
st = spec.procedural_state(cfg, 11)
P, _ = spec.split_state(st)
names = [f"decoder.staff_emb.{w}_{sfx}" for sfx in ("l0", "l0_reverse") for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
Sd = {k: v.to(dev) for k, v in P.items() if k in names or k == "decoder.note_emb.weight"}
E, S = cfg["note_emb_size"], cfg["staff_emb_size"]
V = Sd["decoder.note_emb.weight"].shape[0]
warr = (C.c_void_p * 8)(*[Sd[n].data_ptr() for n in names])


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for what, R, maxlen, lo, hi in (("bulk group", 248, 120, 20, 120), ("long-clip group", 8, 398, 20, 120)):
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, V, (R, maxlen), generator=g).to(dev)
    lengths = torch.randint(lo, hi + 1, (R,), generator=g)
    if maxlen > hi:
        lengths[0] = maxlen
    lengths = lengths.to(dev)
    out = torch.zeros(R, 2 * S, device=dev)
    hsave = torch.zeros(L.a2s_staff_emb_save_floats(R, maxlen, S) if hasattr(L, "a2s_staff_emb_save_floats") else R * 2 * maxlen * S, device=dev)
    dtok = torch.randn(R, 2 * S, device=dev)
    grads = [torch.zeros_like(Sd[n]) for n in names]
    gptrs = torch.tensor([t.data_ptr() for t in grads], dtype=torch.int64, device=dev)
    emb_grad = torch.zeros_like(Sd["decoder.note_emb.weight"])
    fwd = lambda: hip.check(L.a2s_staff_emb_fwd(hip.stream(), hip._p(Sd["decoder.note_emb.weight"]), warr, hip._p(ids), NULL, C.c_long(maxlen), hip._p(lengths),
                                                C.c_long(1), hip._p(out), C.c_long(2 * S), 0, hip._p(hsave), R, maxlen, E, S), "fwd")
    bwd = lambda: hip.check(L.a2s_staff_emb_bwd(hip.stream(), hip._p(Sd["decoder.note_emb.weight"]), warr, hip._p(gptrs), hip._p(emb_grad), hip._p(ids), NULL,
                                                C.c_long(maxlen), hip._p(lengths), C.c_long(1), hip._p(dtok), C.c_long(2 * S), 0, hip._p(hsave), R, maxlen, E, S), "bwd")
    tf, tb = timed(fwd), timed(bwd)
    print(f"{what:16s} R={R:3d} maxlen={maxlen:3d} (E={E}, S={S}, V={V}): forward {tf:7.1f} us = {tf / maxlen:5.2f} us per step of the longest row, "
          f"backward {tb:7.1f} us = {tb / maxlen:5.2f} us per step", flush=True)
