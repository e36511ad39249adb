"""Per-kernel parity of the backward kernels on the MI355X against torch.autograd (CPU, fp32) of the oracle's
primitives.  NB raw pointers do not keep tensors alive: every device operand is bound to a local name."""
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
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _report(name, err):
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/op_errors.txt", "a") as f:
        f.write(f"bwd {name}: {err:.3e}\n")


def test_gru_gates_bwd(dev):
    from oracle import model_ref
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(1)
    R, I, H = 5, 12, 32
    x = torch.randn(R, I, generator=g)
    h = torch.randn(R, H, generator=g, requires_grad=True)
    w_ih, w_hh = torch.randn(3 * H, I, generator=g) * 0.3, torch.randn(3 * H, H, generator=g) * 0.3
    b_ih, b_hh = torch.randn(3 * H, generator=g) * 0.1, torch.randn(3 * H, generator=g) * 0.1
    gi = (x @ w_ih.t() + b_ih).requires_grad_(True)
    gh = (h.detach() @ w_hh.t() + b_hh).requires_grad_(True)
    Hh = H
    r = torch.sigmoid(gi[:, :Hh] + gh[:, :Hh]); z = torch.sigmoid(gi[:, Hh:2 * Hh] + gh[:, Hh:2 * Hh])
    n = torch.tanh(gi[:, 2 * Hh:] + r * gh[:, 2 * Hh:])
    hn = (1 - z) * n + z * h
    dh = torch.randn(R, H, generator=g)
    hn.backward(dh)
    # forward on the device to produce the saved gates, then backward
    gid, ghd, hd = gi.detach().to(dev), gh.detach().to(dev), h.detach().to(dev)
    hout, save = torch.empty(R, H, device=dev), torch.empty(R, 4 * H, device=dev)
    hip.check(L.a2s_gru_gates_fwd(hip.stream(), hip._p(gid), C.c_long(3 * H), hip._p(ghd), C.c_long(3 * H), hip._p(hd), C.c_long(H),
                                  hip._p(hout), C.c_long(H), NULL, C.c_long(0), hip._p(save), R, H), "gates fwd")
    dhd = dh.to(dev)
    dgi, dgh, dhp = torch.empty(R, 3 * H, device=dev), torch.empty(R, 3 * H, device=dev), torch.empty(R, H, device=dev)
    hip.check(L.a2s_gru_gates_bwd(hip.stream(), hip._p(dhd), C.c_long(H), NULL, C.c_long(0), hip._p(save), hip._p(hd), C.c_long(H),
                                  hip._p(dgi), C.c_long(3 * H), hip._p(dgh), C.c_long(3 * H), NULL, C.c_long(0), hip._p(dhp), C.c_long(H), R, H), "gates bwd")
    torch.cuda.synchronize()
    e = max(_rel(hout, hn), _rel(dgi, gi.grad), _rel(dgh, gh.grad), _rel(dhp, h.grad))
    _report("gru_gates", e)
    assert e < 1e-5, e


@pytest.mark.parametrize("H,T,B,S,split", [(32, 41, 3, 4, False), (256, 1201, 2, 3, False), (256, 1201, 2, 3, True), (256, 1201, 70, 2, True), (256, 37, 300, 2, True)])
def test_attention_backward_step_and_deferred_keys(dev, H, T, B, S, split):
    """S steps with different queries through one attention layer: dq per step, deferred dK / dv / dEnc."""
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(H + S)
    enc = torch.randn(B, T, 2 * H, generator=g, requires_grad=True)
    K = (torch.randn(B, T, H, generator=g) * 0.7).requires_grad_(True)
    v = (torch.randn(H, generator=g) * 0.5).requires_grad_(True)
    qs = [(torch.randn(B, H, generator=g) * 0.7).requires_grad_(True) for _ in range(S)]
    dctxs = [torch.randn(B, 2 * H, generator=g) for _ in range(S)]
    loss = 0
    ctxs, aws = [], []
    for q, dc in zip(qs, dctxs):
        a = torch.softmax((torch.tanh(K + q.unsqueeze(1)) * v).sum(-1), dim=1)
        ctx = torch.bmm(a.unsqueeze(1), enc).squeeze(1)
        ctxs.append(ctx.detach()); aws.append(a.detach())
        loss = loss + (ctx * dc).sum()
    loss.backward()
    Kd, encd, vd = torch.exp(2 * K.detach()).to(dev), enc.detach().to(dev), v.detach().to(dev)      # the kernels take the key image exp(2K)
    q_all = torch.stack([q.detach() for q in qs]).to(dev)                    # (S,B,H)
    attw = torch.stack(aws).to(dev)
    ctx_all = torch.stack(ctxs).to(dev)
    dctx_all = torch.stack(dctxs).to(dev)
    dq_all, ds_all = torch.empty(S, B, H, device=dev), torch.empty(S, B, T, device=dev)
    ws = hip.attn_workspace(B, T, H, dev) if split else None
    for s in range(S):
        hip.check(L.a2s_attn_step_bwd(hip.stream(), hip._p(Kd), hip._p(encd), C.c_void_p(q_all[s].data_ptr()), C.c_long(H), hip._p(vd),
                                      C.c_void_p(attw[s].data_ptr()), C.c_void_p(ctx_all[s].data_ptr()), C.c_long(2 * H),
                                      C.c_void_p(dctx_all[s].data_ptr()), C.c_long(2 * H), NULL, C.c_long(0), NULL, C.c_long(0),
                                      C.c_void_p(dq_all[s].data_ptr()), C.c_long(H), C.c_void_p(ds_all[s].data_ptr()), B, T, H, hip._p(ws)), "attn bwd")
    dK = torch.zeros(B, T, H, device=dev)
    nblk = L.a2s_attn_dk_blocks(B, T)
    dvp = torch.zeros(nblk, H, device=dev)
    hip.check(L.a2s_attn_dk_accum(hip.stream(), hip._p(Kd), hip._p(q_all), hip._p(ds_all), hip._p(vd), hip._p(dK), hip._p(dvp), B, T, S, H, C.c_void_p(0), 1), "dk")
    dv = torch.zeros(H, device=dev)
    hip.check(L.a2s_col_sum(hip.stream(), hip._p(dvp), C.c_long(H), hip._p(dv), C.c_long(nblk), H, hip.f32(1.0), hip.f32(0.0), NULL, C.c_size_t(0)), "col_sum")
    # deferred dEnc[b] = sum_s a_s[b]^T dctx_s[b]  as one batched GEMM (T x S)(S x 2H)
    dEnc = torch.zeros(B, T, 2 * H, device=dev)
    hip.gemm(attw, 1, B * T, dctx_all, B * 2 * H, 1, dEnc, 2 * H, T, 2 * H, S, batch=B, bsA=T, bsB=2 * H, bsC=T * 2 * H)
    torch.cuda.synchronize()
    errs = {"dq": max(_rel(dq_all[s], qs[s].grad) for s in range(S)), "dK": _rel(dK, K.grad), "dv": _rel(dv, v.grad), "dEnc": _rel(dEnc, enc.grad)}
    for k, e in errs.items():
        _report(f"attention H{H} B{B} T{T} split={split} {k}", e)
    assert max(errs.values()) < 5e-5, errs


def test_col_sum_two_stage(dev):
    """Long matrices go through the two-stage (partial slabs + final) reduction; alpha/beta semantics must be unchanged."""
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(12)
    rows, Cc, ld = 50011, 300, 320
    x = torch.randn(rows, ld, generator=g)
    out0 = torch.randn(Cc, generator=g)
    ref = 0.5 * x[:, :Cc].double().sum(0) + 2.0 * out0.double()
    xd, outd = x.to(dev), out0.to(dev)
    ws = torch.empty(1024 * Cc, device=dev)
    hip.check(L.a2s_col_sum(hip.stream(), hip._p(xd), C.c_long(ld), hip._p(outd), C.c_long(rows), Cc, hip.f32(0.5), hip.f32(2.0), hip._p(ws),
                            C.c_size_t(ws.numel())), "col_sum 2-stage")
    torch.cuda.synchronize()
    e = _rel(outd, ref.float())
    _report("col_sum two-stage", e)
    assert e < 2e-5, e


def test_log_softmax_bwd_colsum_scatter(dev):
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(4)
    B, bars, U, V, steps, bar = 3, 5, 12, 173, 7, 2
    x = torch.randn(B, bars, U, V, generator=g, requires_grad=True)
    y = torch.log_softmax(x, dim=-1)
    gy = torch.randn(B, bars, U, V, generator=g)
    y.backward(gy)
    yd, gd = y.detach().to(dev), gy.to(dev)
    dx = torch.empty(steps, B, V, device=dev)
    off = 4 * bar * U * V
    hip.check(L.a2s_log_softmax_bwd_rows(hip.stream(), C.c_void_p(gd.data_ptr() + off), C.c_void_p(yd.data_ptr() + off), C.c_long(bars * U * V),
                                         steps, hip._p(dx), B * steps, V, B, 1), "lsm bwd")
    ref = x.grad[:, bar, :steps].permute(1, 0, 2)
    cs = torch.zeros(V, device=dev)
    hip.check(L.a2s_col_sum(hip.stream(), hip._p(dx), C.c_long(V), hip._p(cs), C.c_long(steps * B), V, hip.f32(1.0), hip.f32(0.0), NULL, C.c_size_t(0)), "col_sum")
    # embedding scatter with duplicate ids and a keep mask
    E, R = 16, 40
    ids = torch.randint(0, 9, (R,), generator=g)
    gt = torch.randn(R, 24, generator=g)
    keep = (torch.rand(R, E, generator=g) > 0.3).to(torch.uint8)
    tab_ref = torch.zeros(173, E).index_add_(0, ids, gt[:, 4:4 + E] * keep * 2.0)
    tab = torch.zeros(173, E, device=dev)
    idsd, gtd, keepd = ids.to(dev), gt.to(dev), keep.to(dev)
    hip.check(L.a2s_embed_scatter_add(hip.stream(), hip._p(tab), hip._p(idsd), NULL, C.c_long(1), 0, hip._p(gtd), C.c_long(24), 4, R, E,
                                      hip._p(keepd), hip.f32(2.0)), "scatter")
    torch.cuda.synchronize()
    e = max(_rel(dx, ref), _rel(cs, ref.sum((0, 1))), _rel(tab, tab_ref))
    _report("lsm_bwd/col_sum/scatter", e)
    assert e < 1e-5, e


@pytest.mark.parametrize("layout", ["planes", "cols"])
def test_bn_relu_dropout_backward(dev, layout):
    from oracle import model_ref
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(8)
    Cc = 20
    if layout == "planes":
        rows, F = 14, 24
        x = (torch.randn(rows, Cc, F, generator=g) * 1.5 + 0.5).requires_grad_(True)       # (rows, C, F)
        xin, cdim, mask = x, 1, None
    else:
        rows, F = 300, 1
        x = (torch.randn(rows, Cc, generator=g) * 1.5 + 0.5).requires_grad_(True)
        xin, cdim = x, 1
        mask = (torch.rand(rows, Cc, generator=g) > 0.2).to(torch.uint8)
    P = {"bn.weight": (torch.rand(Cc, generator=g) + 0.5).requires_grad_(True), "bn.bias": (torch.randn(Cc, generator=g) * 0.3).requires_grad_(True)}
    Bf = {"bn.running_mean": torch.zeros(Cc), "bn.running_var": torch.ones(Cc), "bn.num_batches_tracked": torch.tensor(0)}
    y = torch.relu(model_ref.batch_norm(xin, P, Bf, "bn", True, channel_dim=cdim))
    if mask is not None:
        y = y * mask / 0.8
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd, gyd = x.detach().to(dev), gy.to(dev)
    dims = [d for d in range(x.dim()) if d != 1]
    mean = x.detach().mean(dims); var = x.detach().var(dims, unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    scale = P["bn.weight"].detach() * invstd; shift = P["bn.bias"].detach() - mean * scale
    md, isd, scd, shd = mean.to(dev), invstd.to(dev), scale.to(dev), shift.to(dev)
    dgam, dbet, dx = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev), torch.empty_like(xd)
    part = torch.empty(L.a2s_bn_bwd_partial_floats(C.c_long(rows), Cc, F), device=dev)
    c12 = torch.empty(2 * Cc, device=dev)
    maskd = mask.to(dev) if mask is not None else None
    hip.check(L.a2s_bn_bwd(hip.stream(), hip._p(gyd), hip._p(xd), hip._p(md), hip._p(isd), hip._p(scd), hip._p(shd), hip._p(maskd),
                           hip.f32(1 / 0.8), hip._p(dgam), hip._p(dbet), hip._p(dx), hip._p(part), hip._p(c12), C.c_long(rows), Cc, F), "bn_bwd")
    torch.cuda.synchronize()
    errs = {"dx": _rel(dx, x.grad), "dgamma": _rel(dgam, P["bn.weight"].grad), "dbeta": _rel(dbet, P["bn.bias"].grad)}
    for k, e in errs.items():
        _report(f"bn_bwd {layout} {k}", e)
    assert max(errs.values()) < 2e-5, errs


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("Cin,Cout", [(1, 20), (20, 20), (20, 40), (40, 40)])
def test_conv_weight_gradient(dev, Cin, Cout, split):
    """split = 1: the split-operand kernel conv3x3_wgrad_bf16x3 for every eligible launch (switch "wgrad_bf16x3" = 2; the default 1 uses it
    for 40 -> 40 channels only; Cin = 1 keeps its streaming kernel)."""
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(Cin + Cout)
    B, T, F = (2, 11, 56) if not split else (3, 13, 100)
    x = torch.randn(B, T, Cin, F, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2).requires_grad_(True)
    scale, shift = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    use_affine = Cin != 1
    xin = x.permute(0, 2, 1, 3)
    if use_affine:
        xin = torch.relu(xin * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    dy = torch.randn(B, Cout, T, F, generator=g)
    torch.nn.functional.conv2d(xin, w, padding=1).backward(dy)
    xd, dyd = x.to(dev), dy.permute(0, 2, 1, 3).contiguous().to(dev)
    scd, shd = scale.to(dev), shift.to(dev)
    dW = torch.zeros(Cout, Cin, 3, 3, device=dev)
    nb = L.a2s_conv3x3_wgrad_workspace_bytes(Cin, Cout)
    ws = torch.empty(nb // 4, device=dev)
    previous = L.a2s_debug_get(b"wgrad_bf16x3")
    hip.check(L.a2s_debug_set(b"wgrad_bf16x3", 2 * split), "debug_set")
    try:
        hip.check(L.a2s_conv3x3_wgrad(hip.stream(), hip._p(dyd), hip._p(xd), hip._p(scd) if use_affine else NULL, hip._p(shd) if use_affine else NULL,
                                      hip._p(dW), hip._p(ws), C.c_size_t(nb), B, T, F, Cin, Cout), "wgrad")
        torch.cuda.synchronize()
    finally:
        hip.check(L.a2s_debug_set(b"wgrad_bf16x3", previous), "debug_set")
    e = _rel(dW, w.grad)
    _report(f"conv wgrad {Cin}->{Cout} split{split}", e)
    assert e < 2e-5, e


def test_gru_sequence_bptt(dev):
    """One direction pair: dX, dW_ih, dW_hh, db_ih, db_hh and the h_n gradient path, vs autograd of the oracle GRU."""
    from oracle import model_ref
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(21)
    B, T, I, H = 3, 17, 24, 32
    x = torch.randn(B, T, I, generator=g, requires_grad=True)
    P = {}
    for sfx in ("l0", "l0_reverse"):
        P[f"g.weight_ih_{sfx}"] = (torch.randn(3 * H, I, generator=g) * 0.3).requires_grad_(True)
        P[f"g.weight_hh_{sfx}"] = (torch.randn(3 * H, H, generator=g) * 0.3).requires_grad_(True)
        P[f"g.bias_ih_{sfx}"] = (torch.randn(3 * H, generator=g) * 0.1).requires_grad_(True)
        P[f"g.bias_hh_{sfx}"] = (torch.randn(3 * H, generator=g) * 0.1).requires_grad_(True)
    of, hf = model_ref.gru_direction(x, P, "g", "l0")
    orr, hr = model_ref.gru_direction(x, P, "g", "l0_reverse", reverse=True)
    dout = torch.randn(B, T, 2 * H, generator=g)
    dhn = [torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)]
    ((torch.cat([of, orr], 2) * dout).sum() + (hf * dhn[0]).sum() + (hr * dhn[1]).sum()).backward()
    Pd = {k: v.detach().to(dev) for k, v in P.items()}
    xd = x.detach().to(dev).reshape(B * T, I)
    out = torch.empty(B, T, 2 * H, device=dev)
    doutd = dout.to(dev)
    dX = torch.zeros(B * T, I, device=dev)
    worst = 0.0
    for d, sfx in enumerate(("l0", "l0_reverse")):
        gi = hip.linear(xd, Pd[f"g.weight_ih_{sfx}"], Pd[f"g.bias_ih_{sfx}"])
        hbuf, gh, hn = torch.empty(2, B, H, device=dev), torch.empty(B, 3 * H, device=dev), torch.empty(B, H, device=dev)
        gates = torch.empty(T, B, 4 * H, device=dev)
        hip.check(L.a2s_gru_seq_fwd(hip.stream(), hip._p(gi), C.c_long(T * 3 * H), C.c_long(3 * H), hip._p(Pd[f"g.weight_hh_{sfx}"]),
                                    hip._p(Pd[f"g.bias_hh_{sfx}"]), C.c_void_p(out.data_ptr() + 4 * d * H), C.c_long(T * 2 * H), C.c_long(2 * H),
                                    hip._p(hbuf), hip._p(gh), hip._p(gates), hip._p(hn), B, T, H, d, NULL, C.c_size_t(0)), "fwd")
        dgi = torch.empty(B, T, 3 * H, device=dev); dghs = torch.empty(B, T, 3 * H, device=dev)
        dgh_first, dhbuf, dgh_tmp = torch.empty(B, 3 * H, device=dev), torch.empty(2, B, H, device=dev), torch.empty(B, 3 * H, device=dev)
        dhnd = dhn[d].to(dev)
        hip.check(L.a2s_gru_seq_bwd(hip.stream(), C.c_void_p(doutd.data_ptr() + 4 * d * H), C.c_long(T * 2 * H), C.c_long(2 * H),
                                    C.c_void_p(out.data_ptr() + 4 * d * H), C.c_long(T * 2 * H), C.c_long(2 * H), hip._p(gates),
                                    hip._p(Pd[f"g.weight_hh_{sfx}"]), hip._p(dhnd), hip._p(dgi), hip._p(dghs), hip._p(dgh_first), hip._p(dhbuf),
                                    hip._p(dgh_tmp), B, T, H, d, NULL, C.c_size_t(0)), "bwd")
        dgi2 = dgi.view(B * T, 3 * H)
        dWih = torch.zeros(3 * H, I, device=dev)
        hip.gemm(dgi2, 1, 3 * H, xd, I, 1, dWih, I, 3 * H, I, B * T)                      # dW_ih = dgi^T x
        dWhh = torch.zeros(3 * H, H, device=dev)
        hip.gemm(dghs.view(B * T, 3 * H), 1, 3 * H, out, 2 * H, 1, dWhh, H, 3 * H, H, B * T, b_off=d * H)   # dW_hh = dgh_shift^T out[:, dir half]
        dbih, dbhh = torch.zeros(3 * H, device=dev), torch.zeros(3 * H, device=dev)
        hip.check(L.a2s_col_sum(hip.stream(), hip._p(dgi2), C.c_long(3 * H), hip._p(dbih), C.c_long(B * T), 3 * H, hip.f32(1.0), hip.f32(0.0), NULL, C.c_size_t(0)), "cs")
        hip.check(L.a2s_col_sum(hip.stream(), hip._p(dghs), C.c_long(3 * H), hip._p(dbhh), C.c_long(B * T), 3 * H, hip.f32(1.0), hip.f32(0.0), NULL, C.c_size_t(0)), "cs")
        hip.check(L.a2s_col_sum(hip.stream(), hip._p(dgh_first), C.c_long(3 * H), hip._p(dbhh), C.c_long(B), 3 * H, hip.f32(1.0), hip.f32(1.0), NULL, C.c_size_t(0)), "cs")
        hip.gemm(dgi2, 3 * H, 1, Pd[f"g.weight_ih_{sfx}"], I, 1, dX, I, B * T, I, 3 * H, beta=1.0)       # dX += dgi W_ih
        torch.cuda.synchronize()
        errs = {"dW_ih": _rel(dWih, P[f"g.weight_ih_{sfx}"].grad), "dW_hh": _rel(dWhh, P[f"g.weight_hh_{sfx}"].grad),
                "db_ih": _rel(dbih, P[f"g.bias_ih_{sfx}"].grad), "db_hh": _rel(dbhh, P[f"g.bias_hh_{sfx}"].grad)}
        for k, e in errs.items():
            _report(f"gru_bptt {sfx} {k}", e)
        worst = max(worst, *errs.values())
    e = _rel(dX, x.grad.reshape(B * T, I))
    _report("gru_bptt dX", e)
    assert max(worst, e) < 2e-5, (worst, e)


def test_staff_embedding_backward(dev):
    from oracle import model_ref
    from piano_a2s_amd import hip, spec
    L = hip.lib()
    cfg = spec.default_cfg(freq_bins=24, conv_feature_size=32, hidden_size=32, max_length=(12, 8))
    st = spec.procedural_state(cfg, 11)
    P, _ = spec.split_state(st)
    names = [f"decoder.staff_emb.{w}_{sfx}" for sfx in ("l0", "l0_reverse") for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    P = {k: (v.clone().requires_grad_(True) if (k in names or k == "decoder.note_emb.weight") else v) for k, v in P.items()}
    g = torch.Generator().manual_seed(6)
    R, maxlen, E, S = 4, 12, 16, 32
    ids = torch.randint(0, 20, (R, maxlen), generator=g)                 # duplicates on purpose
    lengths = torch.tensor([1, 12, 5, 7])
    tok = model_ref._staff_token(ids, lengths, P).squeeze(1)              # (R, 64)
    dtok = torch.randn(R, 2 * S, generator=g)
    (tok * dtok).sum().backward()
    Sd = {k: v.detach().to(dev) for k, v in P.items()}
    idsd, lend, dtokd = ids.to(dev), lengths.to(dev), dtok.to(dev)
    out = torch.zeros(R, 2 * S, device=dev)
    hsave = torch.zeros(R, 2, maxlen, S, device=dev)
    warr = (C.c_void_p * 8)(*[Sd[n].data_ptr() for n in names])
    hip.check(L.a2s_staff_emb_fwd(hip.stream(), hip._p(Sd["decoder.note_emb.weight"]), warr, hip._p(idsd), NULL, C.c_long(maxlen), hip._p(lend),
                                  C.c_long(1), hip._p(out), C.c_long(2 * S), 0, hip._p(hsave), R, maxlen, E, S), "fwd")
    grads = [torch.zeros_like(Sd[n]) for n in names]
    gptrs = torch.tensor([t.data_ptr() for t in grads], dtype=torch.int64, device=dev)      # DEVICE array of pointers
    emb_grad = torch.zeros_like(Sd["decoder.note_emb.weight"])
    hip.check(L.a2s_staff_emb_bwd(hip.stream(), hip._p(Sd["decoder.note_emb.weight"]), warr, hip._p(gptrs), hip._p(emb_grad), hip._p(idsd), NULL,
                                  C.c_long(maxlen), hip._p(lend), C.c_long(1), hip._p(dtokd), C.c_long(2 * S), 0, hip._p(hsave), R, maxlen, E, S), "bwd")
    torch.cuda.synchronize()
    errs = {n.split(".")[-1]: _rel(gr, P[n].grad) for n, gr in zip(names, grads)}
    errs["note_emb"] = _rel(emb_grad, P["decoder.note_emb.weight"].grad)
    errs["fwd"] = _rel(out, tok)
    for k, e in errs.items():
        _report(f"staff_emb_bwd {k}", e)
    assert max(errs.values()) < 2e-5, errs


@pytest.mark.parametrize("H,T,clips,groups,split", [(256, 1201, 3, 2, True), (256, 333, 70, 3, True), (256, 77, 5, 5, True), (256, 61, 5, 5, True), (256, 61, 5, 4, True),
                                                    (256, 61, 5, 3, True), (256, 61, 5, 2, True), (32, 41, 3, 3, False)])
@pytest.mark.parametrize("fused_combine", [0, 1])
def test_fused_rows_attention_forward_backward(dev, H, T, clips, groups, split, fused_combine):
    """The attention step over `groups` bars of the same clips (row = group * clips + clip) with finished rows skipped: forward
    context / weights, dq, and the deferred dK / dEnc over S steps against torch autograd on the unfinished (row, step) pairs.
    split=True: the multi-row split-T kernels (hidden 256); split=False: the one-workgroup-per-row kernels (no skipping).
    fused_combine=1: the forward partials merged by each clip's last-arriving workgroup (ticket counters in the workspace head,
    switch "attn_fused_combine", off by default) instead of the combine launch."""
    from piano_a2s_amd import hip
    L = hip.lib()
    if fused_combine and not split:
        pytest.skip("the fused combine belongs to the split-T kernels")
    previous = L.a2s_debug_get(b"attn_fused_combine")
    hip.check(L.a2s_debug_set(b"attn_fused_combine", fused_combine), "debug_set")
    try:
        _fused_rows_attention_case(dev, H, T, clips, groups, split, fused_combine)
    finally:
        hip.check(L.a2s_debug_set(b"attn_fused_combine", previous), "debug_set")


def _fused_rows_attention_case(dev, H, T, clips, groups, split, fused_combine):
    from piano_a2s_amd import hip
    L = hip.lib()
    S, R = 4, groups * clips
    g = torch.Generator().manual_seed(H + T + groups)
    enc = torch.randn(clips, T, 2 * H, generator=g, requires_grad=True)
    K = (torch.randn(clips, T, H, generator=g) * 0.7).requires_grad_(True)
    v = (torch.randn(H, generator=g) * 0.5).requires_grad_(True)
    qs = [(torch.randn(R, H, generator=g) * 0.7).requires_grad_(True) for _ in range(S)]
    dctxs = [torch.randn(R, 2 * H, generator=g) for _ in range(S)]
    until = torch.randint(0, S + 1, (R,), generator=g).to(torch.int32)
    until[0] = S                                               # at least one row runs to the end
    if not split:
        until[:] = S                                           # the one-WG-per-row kernels compute every row
    clip_until = until.view(groups, clips).amax(dim=0)
    order = torch.argsort(clip_until, descending=True, stable=True).to(torch.int32)
    rank = torch.empty_like(order)
    rank[order.long()] = torch.arange(clips, dtype=torch.int32)
    rep = lambda x: x.repeat(groups, *([1] * (x.dim() - 1)))   # rows of group j read clip r % clips
    loss = 0
    ctxs, aws = [], []
    for s, (q, dc) in enumerate(zip(qs, dctxs)):
        a = torch.softmax((torch.tanh(rep(K) + q.unsqueeze(1)) * v).sum(-1), dim=1)
        ctx = torch.bmm(a.unsqueeze(1), rep(enc)).squeeze(1)
        live = (until > s).float().unsqueeze(1)
        ctxs.append((ctx * live).detach()); aws.append((a * live).detach())
        loss = loss + (ctx * dc * live).sum()
    loss.backward()
    Kd, encd, vd = torch.exp(2 * K.detach()).to(dev), enc.detach().to(dev), v.detach().to(dev)      # the kernels take the key image exp(2K)
    q_all = torch.stack([q.detach() for q in qs]).to(dev)
    dctx_all = torch.stack(dctxs).to(dev)
    d_until, d_order, d_rank = until.to(dev), order.to(dev), rank.to(dev)
    ws = hip.attn_workspace(clips, T, H, dev, groups=groups) if split else None
    ctx_all, attw = torch.full((S, R, 2 * H), 7.0, device=dev), torch.full((S, R, T), 7.0, device=dev)
    dq_all, ds_all = torch.full((S, R, H), 7.0, device=dev), torch.full((S, R, T), 7.0, device=dev)
    dctx_out = torch.full((S, R, 2 * H), 7.0, device=dev)
    for s in range(S):
        n_active = int((clip_until > s).sum())
        hip.check(L.a2s_attn_step_fwd_rows(hip.stream(), hip._p(Kd), hip._p(encd), C.c_void_p(q_all[s].data_ptr()), C.c_long(H), hip._p(vd),
                                           C.c_void_p(ctx_all[s].data_ptr()), C.c_long(2 * H), NULL, C.c_long(0), C.c_void_p(attw[s].data_ptr()),
                                           R, T, H, hip._p(ws), clips, hip._p(d_order), hip._p(d_rank), hip._p(d_until), n_active, s), "fwd rows")
    for s in range(S):
        n_active = int((clip_until > s).sum())
        hip.check(L.a2s_attn_step_bwd_rows(hip.stream(), hip._p(Kd), hip._p(encd), C.c_void_p(q_all[s].data_ptr()), C.c_long(H), hip._p(vd),
                                           C.c_void_p(attw[s].data_ptr()), C.c_void_p(ctx_all[s].data_ptr()), C.c_long(2 * H),
                                           C.c_void_p(dctx_all[s].data_ptr()), C.c_long(2 * H), NULL, C.c_long(0),
                                           C.c_void_p(dctx_out[s].data_ptr()), C.c_long(2 * H), C.c_void_p(dq_all[s].data_ptr()), C.c_long(H),
                                           C.c_void_p(ds_all[s].data_ptr()), R, T, H, hip._p(ws), clips, hip._p(d_order), hip._p(d_rank),
                                           hip._p(d_until), n_active, s), "bwd rows")
    dK = torch.zeros(clips, T, H, device=dev)
    nblk = L.a2s_attn_dk_blocks(clips, T)
    dvp = torch.zeros(nblk, H, device=dev)
    hip.check(L.a2s_attn_dk_accum(hip.stream(), hip._p(Kd), hip._p(q_all), hip._p(ds_all), hip._p(vd), hip._p(dK), hip._p(dvp), clips, T, S, H,
                                  hip._p(d_until), groups), "dk")
    dv = torch.zeros(H, device=dev)
    hip.check(L.a2s_col_sum(hip.stream(), hip._p(dvp), C.c_long(H), hip._p(dv), C.c_long(nblk), H, hip.f32(1.0), hip.f32(0.0), NULL, C.c_size_t(0)), "col_sum")
    dEnc = torch.zeros(clips, T, 2 * H, device=dev)      # (step, group) is one flat reduction index of the batched GEMM
    hip.gemm(attw, 1, clips * T, dctx_out, clips * 2 * H, 1, dEnc, 2 * H, T, 2 * H, S * groups, batch=clips, bsA=T, bsB=2 * H, bsC=T * 2 * H)
    torch.cuda.synchronize()
    live_dq = torch.stack([(until > s).float() for s in range(S)]).unsqueeze(-1)
    errs = {"ctx": _rel(ctx_all, torch.stack(ctxs)), "attw": _rel(attw, torch.stack(aws)),
            "dq": _rel(dq_all, torch.stack([q.grad for q in qs]) * live_dq), "dK": _rel(dK, K.grad), "dv": _rel(dv, v.grad), "dEnc": _rel(dEnc, enc.grad)}
    for k, e in errs.items():
        _report(f"fused rows H{H} T{T} clips{clips} groups{groups} split={split} fused_combine={fused_combine} {k}", e)
    assert max(errs.values()) < 5e-5, errs


@pytest.mark.parametrize("Cin,Cout,F", [(20, 40, 48), (40, 40, 40), (1, 20, 480), (1, 20, 24)])
def test_weight_gradient_with_fused_batchnorm_backward(dev, Cin, Cout, F):
    """a2s_conv3x3_wgrad_bn (BatchNorm input gradient formed while staging the dy operand, dy written out) against the two-pass form
    a2s_bn_bwd (statistics + apply) followed by a2s_conv3x3_wgrad: same dW, same dy."""
    from piano_a2s_amd import hip
    L = hip.lib()
    B, T = 2, 9
    g0 = torch.Generator().manual_seed(Cin * 100 + Cout + F)
    gact = torch.randn(B, T, Cout, F, generator=g0).to(dev)
    y = torch.randn(B, T, Cout, F, generator=g0).to(dev)
    x = torch.randn(B, T, Cin, F, generator=g0).to(dev)
    mean, invstd = (torch.randn(Cout, generator=g0) * 0.1).to(dev), (torch.rand(Cout, generator=g0) + 0.5).to(dev)
    gamma, beta = (torch.rand(Cout, generator=g0) + 0.5).to(dev), (torch.randn(Cout, generator=g0) * 0.1).to(dev)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    in_scale = (torch.rand(Cin, generator=g0) + 0.5).to(dev) if Cin > 1 else None
    in_shift = (torch.randn(Cin, generator=g0) * 0.1).to(dev) if Cin > 1 else None
    rows = B * T
    part = torch.empty(L.a2s_bn_bwd_partial_floats(C.c_long(rows), Cout, F), device=dev)
    nb = L.a2s_conv3x3_wgrad_workspace_bytes(Cin, Cout)
    ws = torch.empty(nb // 4, device=dev)

    def stats(dx):
        dgam, dbet, c12 = torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev), torch.empty(2 * Cout, device=dev)
        hip.check(L.a2s_bn_bwd(hip.stream(), hip._p(gact), hip._p(y), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), NULL, hip.f32(1.0),
                               hip._p(dgam), hip._p(dbet), hip._p(dx), hip._p(part), hip._p(c12), C.c_long(rows), Cout, F), "bn_bwd")
        return dgam, dbet, c12
    # two passes
    dy_ref = torch.empty_like(gact)
    dgam_ref, dbet_ref, _ = stats(dy_ref)
    dW_ref = torch.zeros(Cout, Cin, 3, 3, device=dev)
    hip.check(L.a2s_conv3x3_wgrad(hip.stream(), hip._p(dy_ref), hip._p(x), hip._p(in_scale), hip._p(in_shift), hip._p(dW_ref), hip._p(ws), C.c_size_t(nb),
                                  B, T, F, Cin, Cout), "wgrad")
    # fused
    dgam, dbet, c12 = stats(None)
    dy = torch.full_like(gact, 7.0)
    dW = torch.zeros(Cout, Cin, 3, 3, device=dev)
    hip.check(L.a2s_conv3x3_wgrad_bn(hip.stream(), hip._p(gact), hip._p(y), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), hip._p(c12),
                                     hip._p(dy), hip._p(x), hip._p(in_scale), hip._p(in_shift), hip._p(dW), hip._p(ws), C.c_size_t(nb), B, T, F, Cin, Cout),
              "wgrad_bn")
    torch.cuda.synchronize()
    errs = {"dy": _rel(dy, dy_ref), "dW": _rel(dW, dW_ref), "dgamma": _rel(dgam, dgam_ref), "dbeta": _rel(dbet, dbet_ref)}
    for k, e in errs.items():
        _report(f"wgrad+bn fused Cin{Cin} Cout{Cout} F{F} {k}", e)
    assert max(errs.values()) < 2e-5, errs


@pytest.mark.parametrize("ci,co,gmag", [(20, 40, 3e-5), (40, 40, 1e-8), (20, 20, 2.0)])
def test_conv_weight_gradient_two_term_fp16(dev, ci, co, gmag):
    """a2s_conv3x3_wgrad_scaled (conv3x3_wgrad_split<.., 2>: dy and the activated input as two fp16 terms each, three MFMA products, dy
    scaled by the power of two derived from the max |dy| scalar) against float64, at gradient magnitudes from 1e-8 to O(1); error relative
    to sum |dy||x| at the fp32-input kernel's level."""
    import ctypes as C
    from piano_a2s_amd import hip
    L = hip.lib()
    torch.manual_seed(3)
    b, T, F = 2, 23, 132
    x = torch.randn(b, T, ci, F)
    dy = gmag * torch.randn(b, T, co, F) * torch.exp(torch.randn(b, T, co, F))
    scale, shift = torch.rand(ci) + 0.5, torch.randn(ci) * 0.1
    act = torch.relu(x.double() * scale.double()[None, None, :, None] + shift.double()[None, None, :, None]).permute(0, 2, 1, 3)
    g = dy.double().permute(0, 2, 1, 3)
    ref = torch.nn.grad.conv2d_weight(act, (co, ci, 3, 3), g, padding=1)
    mag = torch.nn.grad.conv2d_weight(act.abs(), (co, ci, 3, 3), g.abs(), padding=1)
    xd, dyd, scd, shd = x.to(dev), dy.to(dev), scale.to(dev), shift.to(dev)
    amax = dyd.abs().max().reshape(1)
    nb = L.a2s_conv3x3_wgrad_workspace_bytes(ci, co)
    ws = torch.empty(nb // 4, device=dev)
    try:
        L.a2s_debug_set(b"wgrad_f16x2", 2)                     # every eligible launch on the two-term kernel
        dW = torch.zeros(co, ci, 3, 3, device=dev)
        hip.check(L.a2s_conv3x3_wgrad_scaled(hip.stream(), hip._p(dyd), hip._p(xd), hip._p(scd), hip._p(shd), hip._p(dW), hip._p(ws), C.c_size_t(nb),
                                             b, T, F, ci, co, hip._p(amax)), "wgrad")
        torch.cuda.synchronize()
    finally:
        L.a2s_debug_set(b"wgrad_f16x2", 1)
    err = ((dW.cpu().double() - ref).abs() / mag).max().item()
    assert err < 2e-7, err


def test_staff_embedding_kernels_for_the_model_sizes_match_the_generic_ones(dev):
    """Round 3: for note_emb_size 16 / staff_emb_size 32 the staff-embedding recurrence runs on register-resident kernels (a2s_seq.hip
    staff_emb_fwd_e16s32, a2s_bwd.hip staff_emb_bwd_e16s32).  Same calls through both implementations (switch `staff_emb_fast`) at the
    model's real lengths: a full-length row (398), a one-token row, a zero-length row, duplicates among the ids."""
    from piano_a2s_amd import hip, spec
    L = hip.lib()
    cfg = spec.default_cfg()
    st = spec.procedural_state(cfg, 5)
    P, _ = spec.split_state(st)
    names = [f"decoder.staff_emb.{w}_{sfx}" for sfx in ("l0", "l0_reverse") for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    Sd = {k: v.to(dev) for k, v in P.items() if k in names or k == "decoder.note_emb.weight"}
    E, S = cfg["note_emb_size"], cfg["staff_emb_size"]
    assert (E, S) == (16, 32)
    g = torch.Generator().manual_seed(9)
    R, maxlen = 37, 398
    ids = torch.randint(0, 40, (R, maxlen), generator=g).to(dev)
    lengths = torch.randint(1, 200, (R,), generator=g)
    lengths[0], lengths[1], lengths[2] = maxlen, 1, 0
    lengths = lengths.to(dev)
    dtok = torch.randn(R, 2 * S + 6, generator=g).to(dev)
    warr = (C.c_void_p * 8)(*[Sd[n].data_ptr() for n in names])
    res = {}
    prev = L.a2s_debug_get(b"staff_emb_fast")
    try:
        for fast in (0, 1):
            hip.check(L.a2s_debug_set(b"staff_emb_fast", fast), "set")
            out = torch.zeros(R, 2 * S + 6, device=dev)
            hsave = torch.zeros(R, 2, maxlen, S, device=dev)
            hip.check(L.a2s_staff_emb_fwd(hip.stream(), hip._p(Sd["decoder.note_emb.weight"]), warr, hip._p(ids), NULL, C.c_long(maxlen), hip._p(lengths),
                                          C.c_long(1), hip._p(out), C.c_long(out.stride(0)), 3, hip._p(hsave), R, maxlen, E, S), "fwd")
            grads = [torch.zeros_like(Sd[n]) for n in names]
            gptrs = torch.tensor([t.data_ptr() for t in grads], dtype=torch.int64, device=dev)
            emb_grad = torch.zeros_like(Sd["decoder.note_emb.weight"])
            hip.check(L.a2s_staff_emb_bwd(hip.stream(), hip._p(Sd["decoder.note_emb.weight"]), warr, hip._p(gptrs), hip._p(emb_grad), hip._p(ids), NULL,
                                          C.c_long(maxlen), hip._p(lengths), C.c_long(1), hip._p(dtok), C.c_long(dtok.stride(0)), 3, hip._p(hsave), R, maxlen, E, S), "bwd")
            torch.cuda.synchronize()
            res[fast] = [out, hsave, emb_grad] + grads
    finally:
        L.a2s_debug_set(b"staff_emb_fast", prev)
    assert torch.isfinite(res[1][0]).all()
    worst = max(_rel(a, b) for a, b in zip(res[1], res[0]))
    _report("staff_emb e16s32 vs generic", worst)
    assert worst < 2e-5, worst


@pytest.mark.parametrize("Cin,Cout", [(40, 40), (20, 40), (20, 20)])
def test_row_streaming_weight_gradient_matches_the_tiled_kernels(dev, Cin, Cout):
    """csrc/a2s_conv_wrows.hip against round 2's weight-gradient kernels (switch `wgrad_rows`) and float64: F not a multiple of the 128-column
    strip, more work items than workgroups (the accumulators change sign between items), T = 1 and T = 2 clips' worth of halo rows.
    The row kernel keeps ONE fp32 accumulator chain per (co, ci, tap) and workgroup over all of a strip's positions, the tiled kernels many
    short ones: its error is that of a long fp32 sum (2-7e-7 of sum |dz||a|; independent of the operand scales: tools/wgrad_rows_check.py
    with -DWR_KD=10 / 12 / 14 gives the same digits), hence the bar of 1e-6 and 8x the tiled kernels' error."""
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(Cin + Cout)
    worst = 0.0
    for (B, T, F) in ((3, 17, 132), (280, 2, 260), (2, 1, 24)):
        x = torch.randn(B, T, Cin, F, generator=g) * torch.exp(torch.randn(B, T, Cin, F, generator=g))
        dy = 1e-4 * torch.randn(B, T, Cout, F, generator=g) * torch.exp(torch.randn(B, T, Cout, F, generator=g))
        scale, shift = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
        a64 = torch.relu(x.double().permute(0, 2, 1, 3) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
        ref = torch.nn.grad.conv2d_weight(a64, (Cout, Cin, 3, 3), dy.double().permute(0, 2, 1, 3), padding=1)
        mag = torch.nn.grad.conv2d_weight(a64.abs(), (Cout, Cin, 3, 3), dy.double().permute(0, 2, 1, 3).abs(), padding=1) + 1e-300
        prev = L.a2s_debug_get(b"wgrad_rows")
        errs = {}
        try:
            for rows in (1, 0):
                hip.check(L.a2s_debug_set(b"wgrad_rows", rows), "set")
                dW = hip.conv3x3_wgrad_for_test(dy.to(dev), x.to(dev), scale.to(dev), shift.to(dev)).cpu().double()
                assert torch.isfinite(dW).all()
                errs[rows] = float(((dW - ref).abs() / mag).max())
        finally:
            L.a2s_debug_set(b"wgrad_rows", prev)
        _report(f"wgrad rows {Cin}->{Cout} B{B} T{T} F{F}: rows / tiled", errs[1])
        assert errs[1] < 1e-6 and errs[1] <= 8 * errs[0] + 2e-7, (B, T, F, errs)
        worst = max(worst, errs[1])
    assert worst < 1e-6
