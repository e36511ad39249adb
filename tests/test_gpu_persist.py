"""Persistent encoder recurrences (csrc/a2s_persist.hip: one launch for all T steps of a GRU direction, forward and BPTT) against
(i) the oracle GRU on the CPU and (ii) the launch-per-step kernels they replace, at the model's hidden size (the persistent path exists for
H = 256 only) and at batch sizes that are / are not multiples of the 16-row tile."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
NULL = C.c_void_p(0)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _run_direction(L, hip, dev, gi, w_hh, b_hh, dout, dhn, B, T, H, d, persist):
    hip.check(L.a2s_debug_set(b"gru_persist", 1 if persist else 0), "debug_set")
    ws = torch.empty(16 * B * 2048, dtype=torch.float32, device=dev)                  # engine.Engine.encoder passes hip.gemm_workspace(B)
    out = torch.zeros(B, T, 2 * H, device=dev)
    hbuf, gh, hn = torch.empty(2, B, H, device=dev), torch.empty(B, 3 * H, device=dev), torch.empty(B, H, device=dev)
    gates = torch.empty(T, B, 4 * H, device=dev)
    hip.check(L.a2s_gru_seq_fwd(hip.stream(), hip._p(gi), C.c_long(T * 3 * H), C.c_long(3 * H), hip._p(w_hh), hip._p(b_hh),
                                C.c_void_p(out.data_ptr() + 4 * d * H), C.c_long(T * 2 * H), C.c_long(2 * H), hip._p(hbuf), hip._p(gh), hip._p(gates),
                                hip._p(hn), B, T, H, d, hip._p(ws), C.c_size_t(ws.numel() * 4)), "fwd")
    dgi, dghs = torch.empty(B, T, 3 * H, device=dev), torch.empty(B, T, 3 * H, device=dev)
    dgh_first, dhbuf, dgh_tmp = torch.empty(B, 3 * H, device=dev), torch.empty(2, B, H, device=dev), torch.empty(B, 3 * H, device=dev)
    hip.check(L.a2s_gru_seq_bwd(hip.stream(), C.c_void_p(dout.data_ptr() + 4 * d * H), C.c_long(T * 2 * H), C.c_long(2 * H),
                                C.c_void_p(out.data_ptr() + 4 * d * H), C.c_long(T * 2 * H), C.c_long(2 * H), hip._p(gates), hip._p(w_hh), hip._p(dhn),
                                hip._p(dgi), hip._p(dghs), hip._p(dgh_first), hip._p(dhbuf), hip._p(dgh_tmp), B, T, H, d, hip._p(ws),
                                C.c_size_t(ws.numel() * 4)), "bwd")
    torch.cuda.synchronize()
    return dict(out=out[..., d * H:(d + 1) * H].clone(), hn=hn, gates=gates, dgi=dgi, dghs=dghs, dgh_first=dgh_first)


@pytest.mark.parametrize("B,T", [(37, 29), (16, 5), (256, 64), (3, 2)])
def test_persistent_recurrence_equals_stepwise(dev, B, T):
    """Same inputs through the persistent launch and through the launch-per-step kernels: forward state, saved gates, final state, and every
    per-step gradient the deferred weight-gradient products read.  The forward pass sums in the same order (bit-identical); the BPTT product
    runs two accumulator chains instead of one (fp32 round-off)."""
    from piano_a2s_amd import hip
    L = hip.lib()
    H = 256
    g = torch.Generator().manual_seed(100 + B)
    prev = L.a2s_debug_get(b"gru_persist")
    try:
        for d in (0, 1):
            gi = (torch.randn(B, T, 3 * H, generator=g) * 0.8).to(dev)
            w_hh = (torch.randn(3 * H, H, generator=g) * 0.08).to(dev)
            b_hh = (torch.randn(3 * H, generator=g) * 0.1).to(dev)
            dout = torch.randn(B, T, 2 * H, generator=g).to(dev)
            dhn = torch.randn(B, H, generator=g).to(dev)
            a = _run_direction(L, hip, dev, gi, w_hh, b_hh, dout, dhn, B, T, H, d, persist=False)
            b = _run_direction(L, hip, dev, gi, w_hh, b_hh, dout, dhn, B, T, H, d, persist=True)
            for k in ("out", "hn", "gates"):
                assert torch.isfinite(b[k]).all(), f"{k}: non-finite (a spin timed out?)"
                assert torch.equal(a[k], b[k]), f"direction {d} {k}: max diff {float((a[k] - b[k]).abs().max()):.3e}"
            for k in ("dgi", "dghs", "dgh_first"):
                assert torch.isfinite(b[k]).all(), f"{k}: non-finite (a spin timed out?)"
                e = _rel(b[k], a[k])
                assert e < 2e-6, f"direction {d} {k}: {e:.3e}"
    finally:
        hip.check(L.a2s_debug_set(b"gru_persist", prev), "debug_set")


def test_persistent_recurrence_vs_oracle(dev):
    """H = 256 bi-directional layer through the persistent path against the oracle GRU (autograd on the CPU): outputs, final states, dX-side
    gradients (dgi) and the recurrent weight gradient assembled from dgh_shift."""
    from oracle import model_ref
    from piano_a2s_amd import hip
    L = hip.lib()
    B, T, I, H = 5, 19, 48, 256
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, T, I, generator=g, requires_grad=True)
    P = {}
    for sfx in ("l0", "l0_reverse"):
        P[f"g.weight_ih_{sfx}"] = (torch.randn(3 * H, I, generator=g) * 0.2).requires_grad_(True)
        P[f"g.weight_hh_{sfx}"] = (torch.randn(3 * H, H, generator=g) * 0.08).requires_grad_(True)
        P[f"g.bias_ih_{sfx}"] = (torch.randn(3 * H, generator=g) * 0.1).requires_grad_(True)
        P[f"g.bias_hh_{sfx}"] = (torch.randn(3 * H, generator=g) * 0.1).requires_grad_(True)
    of, hf = model_ref.gru_direction(x, P, "g", "l0")
    orr, hr = model_ref.gru_direction(x, P, "g", "l0_reverse", reverse=True)
    dout = torch.randn(B, T, 2 * H, generator=g)
    dhn = [torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)]
    ((torch.cat([of, orr], 2) * dout).sum() + (hf * dhn[0]).sum() + (hr * dhn[1]).sum()).backward()
    prev = L.a2s_debug_get(b"gru_persist")
    hip.check(L.a2s_debug_set(b"gru_persist", 1), "debug_set")
    try:
        xd = x.detach().to(dev).reshape(B * T, I)
        doutd = dout.to(dev)
        for d, (sfx, o_ref, h_ref) in enumerate((("l0", of, hf), ("l0_reverse", orr, hr))):
            Wih, Whh = P[f"g.weight_ih_{sfx}"].detach().to(dev), P[f"g.weight_hh_{sfx}"].detach().to(dev)
            gi = hip.linear(xd, Wih, P[f"g.bias_ih_{sfx}"].detach().to(dev)).view(B, T, 3 * H)
            r = _run_direction(L, hip, dev, gi, Whh, P[f"g.bias_hh_{sfx}"].detach().to(dev), doutd, dhn[d].to(dev), B, T, H, d, persist=True)
            assert _rel(r["out"], o_ref) < 1e-5 and _rel(r["hn"], h_ref) < 1e-5
            dWih = r["dgi"].view(B * T, 3 * H).t() @ xd
            assert _rel(dWih, P[f"g.weight_ih_{sfx}"].grad) < 2e-5
            outd = torch.zeros(B, T, H, device=dev)
            outd.copy_(r["out"])
            dWhh = r["dghs"].view(B * T, 3 * H).t() @ outd.view(B * T, H)
            assert _rel(dWhh, P[f"g.weight_hh_{sfx}"].grad) < 2e-5
            dbhh = r["dghs"].sum((0, 1)) + r["dgh_first"].sum(0)
            assert _rel(dbhh, P[f"g.bias_hh_{sfx}"].grad) < 2e-5
    finally:
        hip.check(L.a2s_debug_set(b"gru_persist", prev), "debug_set")
