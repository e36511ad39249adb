"""Per-kernel parity on the MI355X: every C-ABI entry point of liba2s_hip.so against the same operation in
plain fp32 PyTorch on the CPU (oracle primitives where they exist).  Tolerances are fp32 round-off scaled by
the reduction length; each is stated at the assertion."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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
        f.write(f"{name}: {err:.3e}\n")


@pytest.mark.parametrize("M,N,K", [(7, 173, 1024), (64, 1536, 528), (300, 256, 960), (1201 * 2, 256, 19200 // 8),
                                   (16, 768, 256), (33, 97, 653), (129, 130, 31)])
def test_gemm_nt(dev, M, N, K):
    from piano_a2s_amd import hip
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    ref = A @ W.t() + b
    Ad, Wd, bd = A.to(dev), W.to(dev), b.to(dev)
    out = hip.linear(Ad, Wd, bd)
    torch.cuda.synchronize()
    err = _rel(out, ref)
    _report(f"gemm_nt {M}x{N}x{K}", err)
    assert err < 2e-6 * max(1.0, K ** 0.5 / 8), err        # fp32 accumulation of K products


def test_gemm_strided_forms(dev):
    """The transposed / sub-matrix / accumulate / activation / split-K forms the model uses."""
    from piano_a2s_amd import hip
    g = torch.Generator().manual_seed(3)
    M, N, K = 96, 80, 4100
    A = torch.randn(K, M, generator=g)          # A(m,k) = A[k, m]   (m contiguous: wgrad form)
    Bm = torch.randn(K, N, generator=g)         # B(k,n) = B[k, n]
    ref = A.t() @ Bm
    out = torch.zeros(M, N, device=dev)
    sk = 8
    Ad, Bd = A.to(dev), Bm.to(dev)
    hip.gemm(Ad, 1, M, Bd, N, 1, out, N, M, N, K, splitk=sk)
    torch.cuda.synchronize()
    err = _rel(out, ref)
    _report("gemm_tn_splitk", err)
    assert err < 2e-5, err
    # sub-matrix of the weight (attention W_h half), beta accumulate + tanh
    H = 32
    W = torch.randn(H, 4 * H, generator=g)
    x1, x2 = torch.randn(5, 2 * H, generator=g), torch.randn(5, 2 * H, generator=g)
    ref = torch.tanh(x1 @ W[:, :2 * H].t() + x2 @ W[:, 2 * H:].t())
    out = torch.empty(5, H, device=dev)
    Wd = W.to(dev)
    x1d, x2d = x1.to(dev), x2.to(dev)
    hip.gemm(x1d, 2 * H, 1, Wd, 1, 4 * H, out, H, 5, H, 2 * H)
    hip.gemm(x2d, 2 * H, 1, Wd, 1, 4 * H, out, H, 5, H, 2 * H, beta=1.0, act=2, b_off=2 * H)
    torch.cuda.synchronize()
    err = _rel(out, ref)
    _report("gemm_submatrix_beta_tanh", err)
    assert err < 5e-6, err


@pytest.mark.parametrize("Cin,Cout", [(1, 20), (20, 20), (20, 40), (40, 40)])
@pytest.mark.parametrize("B,T,F", [(2, 9, 24), (1, 41, 480), (1, 6, 100)])
def test_conv3x3_with_input_affine_and_stats(dev, Cin, Cout, B, T, F):
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(Cin * 100 + Cout + T)
    x = torch.randn(B, T, Cin, F, generator=g)                      # (B,T,C,F) layout
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
    scale = torch.rand(Cin, generator=g) + 0.5
    shift = torch.randn(Cin, generator=g) * 0.3
    use_affine = Cin != 1
    xin = x.permute(0, 2, 1, 3)                                      # NCHW for the torch reference
    if use_affine:
        xin = torch.relu(xin * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    ref = torch.nn.functional.conv2d(xin, w, padding=1).permute(0, 2, 1, 3).contiguous()
    y = torch.empty(B, T, Cout, F, device=dev)
    nblk = L.a2s_conv3x3_stat_blocks(B, T, F, Cin)
    part = torch.zeros(nblk, Cout, 2, device=dev)
    # NB: raw pointers do not keep tensors alive -- every device operand is bound to a name for the whole call
    xd, wd, scd, shd = x.to(dev), w.to(dev), scale.to(dev), shift.to(dev)
    cws = hip.conv_workspace(Cin, dev)
    hip.check(L.a2s_conv3x3(hip.stream(), hip._p(xd), hip._p(wd), hip._p(y),
                            hip._p(scd) if use_affine else C.c_void_p(0),
                            hip._p(shd) if use_affine else C.c_void_p(0), hip._p(part), B, T, F, Cin, Cout, 0, hip._p(cws)), "conv")
    torch.cuda.synchronize()
    err = _rel(y, ref)
    _report(f"conv3x3 {Cin}->{Cout} B{B} T{T} F{F}", err)
    assert err < 5e-6, err                                           # K <= 360 products
    sums = part.cpu().double().sum(0)
    ref_s = ref.double().sum(dim=(0, 1, 3))
    ref_s2 = (ref.double() ** 2).sum(dim=(0, 1, 3))
    assert float((sums[:, 0] - ref_s).abs().max()) < 1e-3 * float(ref_s.abs().max().clamp_min(1.0))
    assert float((sums[:, 1] - ref_s2).abs().max()) < 1e-4 * float(ref_s2.abs().max())


def test_conv3x3_flip_is_data_gradient(dev):
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(5)
    B, T, F, Cin, Cout = 2, 7, 24, 20, 40
    x = torch.randn(B, Cin, T, F, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
    dy = torch.randn(B, Cout, T, F, generator=g)
    torch.nn.functional.conv2d(x, w, padding=1).backward(dy)
    ref = x.grad.permute(0, 2, 1, 3).contiguous()                    # (B,T,Cin,F)
    dyl = dy.permute(0, 2, 1, 3).contiguous().to(dev)                # (B,T,Cout,F)
    dx = torch.empty(B, T, Cin, F, device=dev)
    wd = w.to(dev)
    cws = hip.conv_workspace(Cout, dev)
    hip.check(L.a2s_conv3x3(hip.stream(), hip._p(dyl), hip._p(wd), hip._p(dx), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0),
                            B, T, F, Cout, Cin, 1, hip._p(cws)), "conv flip")
    torch.cuda.synchronize()
    err = _rel(dx, ref)
    _report("conv3x3 dgrad(flip)", err)
    assert err < 5e-6, err


@pytest.mark.parametrize("Cin,Cout,flip", [(20, 20, 0), (20, 40, 0), (40, 40, 0), (40, 40, 1), (40, 20, 1), (20, 20, 1)])
@pytest.mark.parametrize("B,T,F", [(2, 9, 24), (1, 41, 480), (1, 6, 100)])
def test_conv3x3_split_operand_kernel(dev, Cin, Cout, flip, B, T, F):
    """conv3x3_bf16x3 (fp32 operands as three bf16 terms on the bf16 matrix pipes) against float64: forward launches with the
    producer's BatchNorm+ReLU and the batch statistics, data-gradient launches with the BatchNorm-backward statistics epilogue.
    The error is measured against sum |a||b| (what an fp32 dot product is allowed to lose) and must stay at the fp32 level."""
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(Cin * 100 + Cout + T + flip)
    x = torch.randn(B, T, Cin, F, generator=g) * torch.exp(torch.randn(B, T, Cin, F, generator=g))      # wide dynamic range
    w = torch.randn((Cin, Cout, 3, 3) if flip else (Cout, Cin, 3, 3), generator=g) * 0.2
    scale, shift = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    xd64 = x.double().permute(0, 2, 1, 3)
    if not flip:
        xd64 = torch.relu(xd64 * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    w64 = w.double().transpose(0, 1).flip(2, 3) if flip else w.double()
    ref = torch.nn.functional.conv2d(xd64, w64, padding=1).permute(0, 2, 1, 3).contiguous()
    mag = torch.nn.functional.conv2d(xd64.abs(), w64.abs(), padding=1).permute(0, 2, 1, 3) + 1e-30
    y = torch.full((B, T, Cout, F), float("nan"), device=dev)
    nblk = L.a2s_conv3x3_stat_blocks(B, T, F, Cin)
    part = torch.zeros(nblk, Cout, 2, device=dev)
    xd, wd, scd, shd = x.to(dev), w.to(dev), scale.to(dev), shift.to(dev)
    cws = hip.conv_workspace(Cin, dev)
    yl = torch.randn(B, T, Cout, F, generator=g)
    mean, invstd = torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5
    bsc, bsh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
    yld, md, isd, bscd, bshd = yl.to(dev), mean.to(dev), invstd.to(dev), bsc.to(dev), bsh.to(dev)
    previous = L.a2s_debug_get(b"conv_bf16x3")
    hip.check(L.a2s_debug_set(b"conv_bf16x3", 3), "debug_set")
    try:
        if flip:
            hip.check(L.a2s_conv3x3_dgrad_bnstats(hip.stream(), hip._p(xd), hip._p(wd), hip._p(y), hip._p(yld), hip._p(md), hip._p(isd), hip._p(bscd),
                                                  hip._p(bshd), hip._p(part), B, T, F, Cin, Cout, hip._p(cws)), "dgrad_bnstats")
        else:
            hip.check(L.a2s_conv3x3(hip.stream(), hip._p(xd), hip._p(wd), hip._p(y), hip._p(scd), hip._p(shd), hip._p(part), B, T, F, Cin, Cout, 0,
                                    hip._p(cws)), "conv")
        torch.cuda.synchronize()
    finally:
        hip.check(L.a2s_debug_set(b"conv_bf16x3", previous), "debug_set")
    assert not torch.isnan(y).any()
    err = float(((y.cpu().double() - ref).abs() / mag).max())
    _report(f"conv3x3 split {Cin}->{Cout} flip{flip} B{B} T{T} F{F} (vs sum|a||b|)", err)
    assert err < 2e-6, err                                            # measured 4e-7 ... 8e-7, the fp32-input MFMA kernel 4e-7 ... 9e-7
    assert _rel(y, ref.float()) < 5e-6
    sums = part.cpu().double().sum(0)
    if flip:
        on = (yl.double() * bsc.double().view(1, 1, -1, 1) + bsh.double().view(1, 1, -1, 1)) > 0
        gm = torch.where(on, ref, torch.zeros_like(ref))
        xhat = (yl.double() - mean.double().view(1, 1, -1, 1)) * invstd.double().view(1, 1, -1, 1)
        ref_s, ref_s2 = gm.sum(dim=(0, 1, 3)), (gm * xhat).sum(dim=(0, 1, 3))
        scale_s = float(torch.where(on, mag, torch.zeros_like(mag)).sum(dim=(0, 1, 3)).max())
        assert float((sums[:, 0] - ref_s).abs().max()) < 1e-5 * scale_s
        assert float((sums[:, 1] - ref_s2).abs().max()) < 1e-5 * scale_s * float(xhat.abs().max())
    else:
        ref_s, ref_s2 = ref.sum(dim=(0, 1, 3)), (ref ** 2).sum(dim=(0, 1, 3))
        assert float((sums[:, 0] - ref_s).abs().max()) < 1e-3 * float(ref_s.abs().max().clamp_min(1.0))
        assert float((sums[:, 1] - ref_s2).abs().max()) < 1e-4 * float(ref_s2.abs().max())


@pytest.mark.parametrize("K,affine", [(1000, False), (960, True), (19200, True)])
def test_gemm_split_operand_path(dev, K, affine):
    """The 128x128 GEMM tile with two k-contiguous operands on the bf16 matrix pipes (fp32 operands as three exact bf16 terms; opt-in
    switch "gemm_bf16x3") against float64, with and without the operand BatchNorm+ReLU; M is large enough for the 128x128 configuration."""
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(K)
    M, N = 12800 + 37, 256
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    period = 480 if K == 19200 else 40
    aff = ((torch.rand(K // period, generator=g) + 0.5).to(dev), (torch.randn(K // period, generator=g) * 0.3).to(dev), period) if affine else None
    xd = x.double()
    if affine:
        xd = torch.relu(xd * aff[0].double().repeat_interleave(period) + aff[1].double().repeat_interleave(period))
    ref = xd @ w.double().t() + b.double()
    mag = xd.abs() @ w.double().abs().t() + b.double().abs() + 1e-30
    previous = L.a2s_debug_get(b"gemm_bf16x3")
    errs = {}
    try:
        for mode in (0, 1):
            hip.check(L.a2s_debug_set(b"gemm_bf16x3", mode), "debug_set")
            y = hip.linear(x, w, b, x_affine=aff)
            torch.cuda.synchronize()
            errs[mode] = float(((y.double() - ref).abs() / mag).max())
    finally:
        hip.check(L.a2s_debug_set(b"gemm_bf16x3", previous), "debug_set")
    _report(f"gemm split K{K} affine{int(affine)} (vs sum|a||b|): fp32-input", errs[0])
    _report(f"gemm split K{K} affine{int(affine)} (vs sum|a||b|): split", errs[1])
    assert errs[1] < 2e-6, errs
    assert errs[1] < 3 * errs[0] + 1e-7, errs          # at the fp32-input kernel's level


@pytest.mark.parametrize("form", ["data_gradient", "weight_gradient"])
def test_gemm_split_operand_path_row_contiguous_operands(dev, form):
    """Split-operand GEMM tiles with operands whose ROW dimension is the contiguous one (transposed staging): the data-gradient form
    (B = W[k][n]) and the weight-gradient form (both operands row-contiguous, K = the long row count, split-K, BatchNorm+ReLU on B)."""
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(len(form))
    if form == "data_gradient":
        M, K, N = 12800 + 37, 1000, 256
        a = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
        w = (torch.randn(K, N, generator=g) * 0.05).to(dev)
        ref = a.double() @ w.double()
        mag = a.double().abs() @ w.double().abs() + 1e-30
        run = lambda out: hip.gemm(a, K, 1, w, N, 1, out, N, M, N, K)
        shape = (M, N)
    else:
        R, Mo, No, period = 30000 + 3, 256, 512, 64
        dz = torch.randn(R, Mo, generator=g).to(dev)
        x = (torch.randn(R, No, generator=g) * torch.exp(torch.randn(R, 1, generator=g))).to(dev)
        aff = ((torch.rand(No // period, generator=g) + 0.5).to(dev), (torch.randn(No // period, generator=g) * 0.3).to(dev), period)
        xa = torch.relu(x.double() * aff[0].double().repeat_interleave(period) + aff[1].double().repeat_interleave(period))
        ref = dz.double().t() @ xa
        mag = dz.double().abs().t() @ xa.abs() + 1e-30
        run = lambda out: hip.gemm(dz, 1, Mo, x, No, 1, out, No, Mo, No, R, splitk=24, b_affine=aff)
        shape = (Mo, No)
    previous = L.a2s_debug_get(b"gemm_bf16x3")
    errs = {}
    try:
        for mode in (0, 1):
            hip.check(L.a2s_debug_set(b"gemm_bf16x3", mode), "debug_set")
            out = torch.full(shape, float("nan"), device=dev)
            run(out)
            torch.cuda.synchronize()
            errs[mode] = float(((out.double() - ref).abs() / mag).max())
    finally:
        hip.check(L.a2s_debug_set(b"gemm_bf16x3", previous), "debug_set")
    _report(f"gemm split {form} (vs sum|a||b|): fp32-input", errs[0])
    _report(f"gemm split {form} (vs sum|a||b|): split", errs[1])
    assert errs[1] < 2e-6, errs
    assert errs[1] < 3 * errs[0] + 1e-7, errs


@pytest.mark.parametrize("form", ["forward", "data_gradient", "weight_gradient"])
def test_gemm_two_term_fp16_path(dev, form):
    """The 128x128 GEMM tiles on the two-term fp16 split (three matrix-core products; power-of-two operand scales from max|operand|
    device scalars) against float64 and against the three-term bf16 split, in the three forms of the 19200 -> 256 Linear: forward
    (O(1) activations unscaled + BatchNorm/ReLU while staging, small weights scaled), data gradient (tiny gradients x small weights,
    both scaled), weight gradient (transposed staging, split-K, tiny gradients scaled x affine activations unscaled)."""
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(7 + len(form))
    if form == "forward":
        M, K, N, period = 12800 + 37, 1920, 256, 48
        x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.007).to(dev)
        aff = ((torch.rand(K // period, generator=g) + 0.5).to(dev), (torch.randn(K // period, generator=g) * 0.3).to(dev), period)
        xa = torch.relu(x.double() * aff[0].double().repeat_interleave(period) + aff[1].double().repeat_interleave(period))
        ref, mag = xa @ w.double().t(), xa.abs() @ w.double().abs().t() + 1e-300
        wmax = hip.absmax(w)
        assert float(wmax) == float(w.abs().max())
        run = lambda out, tt: hip.linear(x, w, out=out, x_affine=aff, two_term=(None, wmax) if tt else None)
        shape = (M, N)
    elif form == "data_gradient":
        M, K, N = 12800 + 37, 256, 1920
        dz = (torch.randn(M, K, generator=g) * 3e-7 * torch.exp(2 * torch.randn(M, 1, generator=g))).to(dev)
        wt = (torch.randn(N, K, generator=g) * 0.007).to(dev)                 # k-contiguous copy of the weight, as engine_bwd passes it
        ref, mag = dz.double() @ wt.double().t(), dz.double().abs() @ wt.double().abs().t() + 1e-300
        amax, wmax = hip.absmax(dz), hip.absmax(wt)
        # rows 2^-20 and more below the tensor's maximum meet the ABSOLUTE floor of the two-term split: the second term of an element
        # below 2^-3 (after scaling the maximum to 2^12) is an fp16 subnormal, quantum 2^-24, i.e. an error of at most 2^-37 max|dz| per
        # element whatever its size -- negligible in every sum the gradient enters, but not relative to such a row alone
        mag = mag + (2.0 ** -36 / 2e-6) * (float(amax) * wt.double().abs().sum(dim=1)[None, :] + float(wmax) * dz.double().abs().sum(dim=1)[:, None])
        run = lambda out, tt: hip.gemm(dz, K, 1, wt, 1, K, out, N, M, N, K, two_term=(amax, wmax) if tt else None)
        shape = (M, N)
    else:
        R, Mo, No, period = 30000 + 3, 256, 512, 64
        dz = (torch.randn(R, Mo, generator=g) * 3e-7 * torch.exp(2 * torch.randn(R, 1, generator=g))).to(dev)
        x = (torch.randn(R, No, generator=g) * torch.exp(torch.randn(R, 1, generator=g))).to(dev)
        aff = ((torch.rand(No // period, generator=g) + 0.5).to(dev), (torch.randn(No // period, generator=g) * 0.3).to(dev), period)
        xa = torch.relu(x.double() * aff[0].double().repeat_interleave(period) + aff[1].double().repeat_interleave(period))
        ref, mag = dz.double().t() @ xa, dz.double().abs().t() @ xa.abs() + 1e-300
        amax = hip.absmax(dz)
        run = lambda out, tt: hip.gemm(dz, 1, Mo, x, No, 1, out, No, Mo, No, R, splitk=24, b_affine=aff, two_term=(amax, None) if tt else None)
        shape = (Mo, No)
    assert L.a2s_debug_get(b"gemm_f16x2") == 1 and L.a2s_debug_get(b"gemm_bf16x3") == 1
    errs = {}
    for tt in (False, True):
        out = torch.full(shape, float("nan"), device=dev)
        run(out, tt)
        torch.cuda.synchronize()
        errs[tt] = float(((out.double() - ref).abs() / mag).max())
    _report(f"gemm two-term {form} (vs sum|a||b|): three bf16 terms", errs[False])
    _report(f"gemm two-term {form} (vs sum|a||b|): two fp16 terms", errs[True])
    assert errs[True] < 2e-6, errs
    assert errs[True] < 3 * errs[False] + 1e-7, errs


@pytest.mark.parametrize("training", [True, False])
def test_bn_finalize_matches_oracle_batch_norm(dev, training):
    from oracle import model_ref
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(9)
    Cc, n = 20, 4000
    x = torch.randn(n, Cc, generator=g) * 2 + 3                      # non-zero mean: exercises the E[x^2]-m^2 form
    P = {"bn.weight": torch.rand(Cc, generator=g) + 0.5, "bn.bias": torch.randn(Cc, generator=g)}
    Bf = {"bn.running_mean": torch.randn(Cc, generator=g), "bn.running_var": torch.rand(Cc, generator=g) + 0.5,
          "bn.num_batches_tracked": torch.tensor(3)}
    Bd = {k: v.clone().to(dev) for k, v in Bf.items()}
    ref = model_ref.batch_norm(x, P, Bf, "bn", training, channel_dim=1)
    rpb = 64
    nblk = (n + rpb - 1) // rpb
    part = torch.zeros(nblk, Cc, 2, device=dev)
    xd = x.to(dev)
    hip.check(L.a2s_col_stats(hip.stream(), hip._p(xd), hip._p(part), C.c_long(n), Cc, rpb), "col_stats")
    mean, invstd, scale, shift = (torch.empty(Cc, device=dev) for _ in range(4))
    gd, bd = P["bn.weight"].to(dev), P["bn.bias"].to(dev)
    hip.check(L.a2s_bn_finalize(hip.stream(), hip._p(part), nblk, Cc, C.c_double(n), hip._p(gd),
                                hip._p(bd), hip._p(Bd["bn.running_mean"]), hip._p(Bd["bn.running_var"]),
                                hip._p(Bd["bn.num_batches_tracked"]), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift),
                                hip.f32(1e-5), hip.f32(0.1), int(training)), "bn_finalize")
    torch.cuda.synchronize()
    y = xd * scale + shift
    err = _rel(y, ref)
    _report(f"bn training={training}", err)
    assert err < 5e-6, err
    for k in Bf:
        assert _rel(Bd[k].float(), Bf[k].float()) < 1e-5, k


@pytest.mark.parametrize("H,T,B,split", [(32, 41, 3, False), (256, 1201, 2, False), (256, 1201, 2, True), (256, 1201, 70, True), (256, 37, 300, True)])
def test_attention_step(dev, H, T, B, split):
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(H + T)
    enc = torch.randn(B, T, 2 * H, generator=g)
    W = torch.randn(H, 4 * H, generator=g) * (3.0 / (4 * H)) ** 0.5 * 4
    bias = torch.randn(H, generator=g) * 0.1
    v = torch.randn(1, H, generator=g) * (3.0 / H) ** 0.5 * 4
    hid = torch.randn(1, B, 2 * H, generator=g)
    from oracle import model_ref
    P = {"a.attn.weight": W, "a.attn.bias": bias, "a.v.weight": v}
    a_ref = model_ref.attention(hid, enc, P, "a")
    ctx_ref = torch.bmm(a_ref.unsqueeze(1), enc).squeeze(1)
    encd, Wd = enc.to(dev), W.to(dev)
    keys = torch.empty(B * T, H, device=dev)
    hip.gemm(encd, 2 * H, 1, Wd, 1, 4 * H, keys, H, B * T, H, 2 * H, b_off=2 * H, act=3)      # key image exp(2K), see include/a2s.h
    q = torch.empty(B, H, device=dev)
    hd, biasd, vd = hid[0].to(dev), bias.to(dev), v.to(dev)
    hip.gemm(hd, 2 * H, 1, Wd, 1, 4 * H, q, H, B, H, 2 * H, bias=biasd)
    ws = hip.attn_workspace(B, T, H, dev) if split else None
    previous = L.a2s_debug_get(b"attn_fused_combine")
    try:
        for fused in ((0, 1) if split else (0,)):           # split-T kernels: separate combine launch / merged by the last-arriving workgroup
            hip.check(L.a2s_debug_set(b"attn_fused_combine", fused), "debug_set")
            for rep in range(2):                            # twice: the arrival counters must be back at zero after a launch
                ctx = torch.full((B, 2 * H), 7.0, device=dev)
                attw = torch.full((B, T), 7.0, device=dev)
                hip.check(L.a2s_attn_step_fwd(hip.stream(), hip._p(keys), hip._p(encd), hip._p(q), C.c_long(H), hip._p(vd), hip._p(ctx),
                                              C.c_long(2 * H), C.c_void_p(0), C.c_long(0), hip._p(attw), B, T, H, C.c_void_p(0), 0, hip._p(ws)), "attn")
                torch.cuda.synchronize()
                e1, e2 = _rel(attw, a_ref), _rel(ctx, ctx_ref)
                if rep == 0:
                    _report(f"attention H{H} T{T} B{B} split={split} fused_combine={fused} weights", e1)
                    _report(f"attention H{H} T{T} B{B} split={split} fused_combine={fused} context", e2)
                assert e1 < 2e-5 and e2 < 2e-5, (fused, rep, e1, e2)
    finally:
        hip.check(L.a2s_debug_set(b"attn_fused_combine", previous), "debug_set")


def test_gru_sequence_both_directions(dev):
    from oracle import model_ref
    from piano_a2s_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(17)
    B, T, I, H = 3, 23, 32, 32
    x = torch.randn(B, T, I, generator=g)
    P = {}
    for sfx in ("l0", "l0_reverse"):
        P[f"g.weight_ih_{sfx}"] = torch.randn(3 * H, I, generator=g) * 0.3
        P[f"g.weight_hh_{sfx}"] = torch.randn(3 * H, H, generator=g) * 0.3
        P[f"g.bias_ih_{sfx}"] = torch.randn(3 * H, generator=g) * 0.1
        P[f"g.bias_hh_{sfx}"] = torch.randn(3 * H, generator=g) * 0.1
    of, hf = model_ref.gru_direction(x, P, "g", "l0")
    orr, hr = model_ref.gru_direction(x, P, "g", "l0_reverse", reverse=True)
    out = torch.empty(B, T, 2 * H, device=dev)
    xd = x.to(dev).reshape(B * T, I)
    hns = []
    Pd = {k: v.to(dev) for k, v in P.items()}
    for d, sfx in enumerate(("l0", "l0_reverse")):
        gi = hip.linear(xd, Pd[f"g.weight_ih_{sfx}"], Pd[f"g.bias_ih_{sfx}"])
        hbuf, gh, hn = torch.empty(2, B, H, device=dev), torch.empty(B, 3 * H, device=dev), torch.empty(B, H, device=dev)
        hip.check(L.a2s_gru_seq_fwd(hip.stream(), hip._p(gi), C.c_long(T * 3 * H), C.c_long(3 * H), hip._p(Pd[f"g.weight_hh_{sfx}"]),
                                    hip._p(Pd[f"g.bias_hh_{sfx}"]), C.c_void_p(out.data_ptr() + 4 * d * H), C.c_long(T * 2 * H),
                                    C.c_long(2 * H), hip._p(hbuf), hip._p(gh), C.c_void_p(0), hip._p(hn), B, T, H, d, C.c_void_p(0), C.c_size_t(0)), "gru_seq")
        hns.append(hn)
    torch.cuda.synchronize()
    e = max(_rel(out[..., :H], of), _rel(out[..., H:], orr), _rel(hns[0], hf), _rel(hns[1], hr))
    _report("gru_seq bidir", e)
    assert e < 1e-5, e


def test_staff_embedding_ragged_lengths(dev):
    from oracle import model_ref
    from piano_a2s_amd import engine, spec
    cfg = spec.default_cfg(freq_bins=24, conv_feature_size=32, hidden_size=32, max_length=(12, 8))
    st = spec.procedural_state(cfg, 11)
    P, _ = spec.split_state(st)
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 173, (4, 12), generator=g)
    lengths = torch.tensor([1, 12, 5, 7])
    ref = model_ref._staff_token(ids, lengths, P).squeeze(1)
    eng = engine.Engine(cfg)
    S = {k: v.to(dev) for k, v in st.items()}
    out = torch.zeros(4, 64, device=dev)
    eng._staff_token(S, ids.to(dev), lengths.to(dev), 1, out, 0, 12, 12, True)
    out32 = torch.zeros(4, 64, device=dev)
    eng._staff_token(S, ids.to(torch.int32).to(dev), lengths.to(dev), 1, out32, 0, 12, 12, False)
    torch.cuda.synchronize()
    e = max(_rel(out, ref), _rel(out32, ref))
    _report("staff_emb ragged", e)
    assert e < 1e-5, e


def test_gemm_with_operand_batchnorm_relu(dev):
    """a2s_gemm_f32_affine: BatchNorm+ReLU of an operand formed while it is staged -- the forward Linear form (A k-contiguous,
    channel = k // period) and the weight-gradient form (B n-contiguous, channel = n // period), ragged sizes (zero padding must stay 0)."""
    from piano_a2s_amd import hip
    g = torch.Generator().manual_seed(5)
    M, N, period, chans = 203, 37, 12, 7
    K = period * chans
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.2
    scale, shift = torch.rand(chans, generator=g) + 0.5, torch.randn(chans, generator=g) * 0.3
    act = torch.relu(x * scale.repeat_interleave(period) + shift.repeat_interleave(period))
    xd, Wd, sc, sh = x.to(dev), W.to(dev), scale.to(dev), shift.to(dev)
    y = hip.linear(xd, Wd, x_affine=(sc, sh, period))
    e1 = _rel(y, act @ W.t())
    dy = torch.randn(M, N, generator=g)
    dW = torch.zeros(N, K, device=dev)
    dyd = dy.to(dev)
    hip.gemm(dyd, 1, N, xd, K, 1, dW, K, N, K, M, beta=1.0, b_affine=(sc, sh, period))            # dW = dy^T act(x)
    e2 = _rel(dW, dy.t() @ act)
    dW2 = torch.zeros(N, K, device=dev)
    hip.gemm(dyd, 1, N, xd, K, 1, dW2, K, N, K, M, beta=1.0, splitk=3, b_affine=(sc, sh, period))
    e3 = _rel(dW2, dy.t() @ act)
    torch.cuda.synchronize()
    _report("gemm operand affine forward", e1)
    _report("gemm operand affine wgrad", max(e2, e3))
    assert max(e1, e2, e3) < 2e-5, (e1, e2, e3)


@pytest.mark.parametrize("M,period,chans,K", [(203, 128, 3, 40), (300, 480, 2, 64), (130, 132, 5, 16)])
def test_gemm_with_batchnorm_backward_statistics_epilogue(dev, M, period, chans, K):
    """a2s_gemm_f32_bnstats: C = A B plus, in the epilogue, the per-channel sums  sum g', sum g' xhat  of the output read as the gradient
    wrt relu(bn(y)) (channel = column // period; channels straddle the 128-column tiles), in the partial layout bn_bwd_finalize reads."""
    from piano_a2s_amd import hip
    L = hip.lib()
    g0 = torch.Generator().manual_seed(M + period)
    N = period * chans
    A = torch.randn(M, K, generator=g0)
    Bm = torch.randn(K, N, generator=g0) * 0.2
    y = torch.randn(M, N, generator=g0)
    mean, invstd = torch.randn(chans, generator=g0) * 0.1, torch.rand(chans, generator=g0) + 0.5
    scale, shift = torch.rand(chans, generator=g0) + 0.5, torch.randn(chans, generator=g0) * 0.3
    Cref = A @ Bm
    ch = torch.arange(N) // period
    mask = (y * scale[ch] + shift[ch]) > 0
    gm = torch.where(mask, Cref, torch.zeros(()))
    xhat = (y - mean[ch]) * invstd[ch]
    s1 = torch.stack([gm[:, ch == c].sum() for c in range(chans)])
    s2 = torch.stack([(gm * xhat)[:, ch == c].sum() for c in range(chans)])
    Ad, Bd, yd = A.to(dev), Bm.to(dev), y.to(dev)
    md, isd, scd, shd = mean.to(dev), invstd.to(dev), scale.to(dev), shift.to(dev)      # named: raw pointers keep nothing alive
    Cd = torch.empty(M, N, device=dev)
    nblk = L.a2s_gemm_bnstats_blocks(M, period)
    part = torch.full((nblk, chans, 2), 7.0, device=dev)
    hip.check(L.a2s_gemm_f32_bnstats(hip.stream(), M, N, K, hip._p(Ad), C.c_long(K), C.c_long(1), hip._p(Bd), C.c_long(N), C.c_long(1), hip._p(Cd), C.c_long(N),
                                     hip._p(yd), hip._p(md), hip._p(isd), hip._p(scd), hip._p(shd), period,
                                     hip._p(part)), "gemm bnstats")
    torch.cuda.synchronize()
    tot = part.double().sum(0).cpu()
    e_c = _rel(Cd, Cref)
    e1 = float((tot[:, 0] - s1.double()).abs().max() / s1.abs().max())
    e2 = float((tot[:, 1] - s2.double()).abs().max() / s2.abs().max())
    _report(f"gemm bn-stats epilogue M{M} period{period} C", e_c)
    _report(f"gemm bn-stats epilogue M{M} period{period} sums", max(e1, e2))
    assert e_c < 2e-5 and e1 < 2e-5 and e2 < 2e-5, (e_c, e1, e2)
