"""Data gradient of the 19200 -> 256 Linear on its own kernel (csrc/a2s_linear.hip: weight pre-split into fp16 term planes, a workgroup's rows of
dz resident in LDS, barrier-free sweep over the column tiles) against float64 and against the generic two-term GEMM tile it replaces:
the product itself and the BatchNorm-backward statistics of its epilogue (reference models.py:68, backward of y = relu(bn4(y4)) W^T)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _run(L, hip, dev, dz, Wt, y, mean, invstd, scale, shift, period, new):
    M, K = dz.shape
    N = Wt.shape[0]
    da = torch.full((M, N), float("nan"), device=dev)
    dmax, wmax = hip.absmax(dz), hip.absmax(Wt)
    if new:
        nblk = L.a2s_linear_dgrad_blocks(M)
        part = torch.full((nblk, N // period, 2), float("nan"), device=dev)
        nb = L.a2s_linear_dgrad_ws_bytes(N, K)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=dev)
        damax = torch.zeros(1, device=dev)
        hip.check(L.a2s_linear_dgrad_bnstats(hip.stream(), M, N, K, hip._p(dz), C.c_long(K), hip._p(Wt), hip._p(da), C.c_long(N), hip._p(y), hip._p(mean),
                                             hip._p(invstd), hip._p(scale), hip._p(shift), period, hip._p(part), hip._p(dmax), hip._p(wmax), hip._p(ws),
                                             C.c_size_t(nb), hip._p(damax)), "linear_dgrad")
    else:
        nblk = L.a2s_gemm_bnstats_blocks(M, period)
        part = torch.full((nblk, N // period, 2), float("nan"), device=dev)
        hip.check(L.a2s_gemm_f32_bnstats_scaled(hip.stream(), M, N, K, hip._p(dz), C.c_long(K), C.c_long(1), hip._p(Wt), C.c_long(1), C.c_long(K), hip._p(da),
                                                C.c_long(N), hip._p(y), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), period, hip._p(part),
                                                hip._p(dmax), hip._p(wmax)), "gemm bnstats")
    torch.cuda.synchronize()
    if new:
        assert float(damax) == float(da.abs().max()), "max |da| written beside the product"
    return da, part.double().sum(0)


@pytest.mark.parametrize("M,channels,period,gscale", [(300, 40, 128, 1e-5), (128, 10, 160, 1.0), (1201 * 2, 40, 480, 3e-7), (517, 3, 128, 1e3)])
def test_linear_dgrad_kernel(dev, M, channels, period, gscale):
    from piano_a2s_amd import hip
    L = hip.lib()
    K, N = 256, channels * period
    assert L.a2s_linear_dgrad_eligible(M, N, K, period) == 1
    g = torch.Generator().manual_seed(M + N)
    dz = (torch.randn(M, K, generator=g) * gscale).to(dev)
    dz[M // 2] *= 2.0 ** -12                                     # a row far below the tensor's maximum
    Wt = (torch.randn(N, K, generator=g) * 0.007).to(dev)
    y = torch.randn(M, N, generator=g).to(dev)
    mean = (torch.randn(channels, generator=g) * 0.1).to(dev)
    invstd = (torch.rand(channels, generator=g) + 0.5).to(dev)
    scale = (torch.randn(channels, generator=g)).to(dev)         # both signs: the mask is bn(y) > 0, not y > mean
    shift = (torch.randn(channels, generator=g) * 0.3).to(dev)
    da_new, st_new = _run(L, hip, dev, dz, Wt, y, mean, invstd, scale, shift, period, True)
    da_old, st_old = _run(L, hip, dev, dz, Wt, y, mean, invstd, scale, shift, period, False)
    ref = dz.double() @ Wt.double().t()
    bound = (dz.double().abs() @ Wt.double().abs().t())          # sum |a||b|: the scale of the rounding error of any fp32-level product
    assert torch.isfinite(da_new).all()
    err_new = float(((da_new.double() - ref).abs() / bound).max())
    err_old = float(((da_old.double() - ref).abs() / bound).max())
    assert err_new < 1e-6, f"product: {err_new:.3e} of sum|a||b| (generic tile: {err_old:.3e})"
    assert err_new <= 2.0 * err_old + 1e-7, f"{err_new:.3e} vs the generic tile's {err_old:.3e}"
    # statistics from the float64 product
    ch = torch.arange(N, device=dev) // period
    z = y.double() * scale.double()[ch] + shift.double()[ch]
    gm = torch.where(z > 0, ref, torch.zeros_like(ref))
    xhat = (y.double() - mean.double()[ch]) * invstd.double()[ch]
    s1 = gm.sum(0).view(channels, period).sum(1)
    s2 = (gm * xhat).sum(0).view(channels, period).sum(1)
    a1 = gm.abs().sum(0).view(channels, period).sum(1)
    a2 = (gm * xhat).abs().sum(0).view(channels, period).sum(1)
    for name, got, want, mag in (("sum g'", st_new[:, 0], s1, a1), ("sum g' xhat", st_new[:, 1], s2, a2)):
        e = float(((got - want).abs() / mag.clamp_min(1e-300)).max())
        assert e < 2e-6, f"{name}: {e:.3e} of the sum of magnitudes"
    for k in range(2):
        mag = a1 if k == 0 else a2
        assert float(((st_new[:, k] - st_old[:, k]).abs() / mag.clamp_min(1e-300)).max()) < 2e-6


@pytest.mark.parametrize("M,channels,period", [(300, 40, 128), (77, 6, 96), (1201 * 2, 40, 480), (128, 4, 32)])
def test_linear_forward_kernel(dev, M, channels, period):
    """z = relu(bn(y)) W^T on the kernel of csrc/a2s_linear.hip against float64 and against the generic two-term tile (an odd number of 64-k blocks in one case, channels that
    straddle a block in two, rows that are not a multiple of 128)."""
    import os
    from piano_a2s_amd import hip
    L = hip.lib()
    K, N = channels * period, 256
    assert L.a2s_linear_fwd_eligible(M, N, K, period) == 1
    g = torch.Generator().manual_seed(M + K)
    y = (torch.randn(M, K, generator=g) * 3.0).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.007).to(dev)
    scale = (torch.randn(channels, generator=g)).to(dev)
    shift = (torch.randn(channels, generator=g) * 0.3).to(dev)
    ymax = y.view(M, channels, period).abs().amax(dim=(0, 2)).contiguous()
    bound, wmax = hip.act_bound(scale, shift, ymax), hip.absmax(W)
    z_new = hip.linear_forward(y, W, (scale, shift, period), bound, wmax)
    hip.LINEAR_KERNELS = False                    # (the generic two-term GEMM tiles)
    try:
        z_old = hip.linear_forward(y, W, (scale, shift, period), bound, wmax)
    finally:
        hip.LINEAR_KERNELS = True
    torch.cuda.synchronize()
    ch = torch.arange(K, device=dev) // period
    a = torch.relu(y.double() * scale.double()[ch] + shift.double()[ch])
    ref = a @ W.double().t()
    mag = a.abs() @ W.double().abs().t()
    assert torch.isfinite(z_new).all()
    err_new = float(((z_new.double() - ref).abs() / mag.clamp_min(1e-300)).max())
    err_old = float(((z_old.double() - ref).abs() / mag.clamp_min(1e-300)).max())
    assert err_new < 1e-6, f"{err_new:.3e} of sum|a||w| (generic tile: {err_old:.3e})"
    assert err_new <= 2.0 * err_old + 1e-7, f"{err_new:.3e} vs the generic tile's {err_old:.3e}"


@pytest.mark.parametrize("M,channels,period,gscale", [(300, 40, 128, 1e-5), (64 * 37 + 5, 6, 64, 1.0), (1201 * 2, 40, 480, 3e-7), (2048 + 64, 1, 128, 1e2)])
def test_linear_weight_gradient_kernel(dev, M, channels, period, gscale):
    """G += dz^T relu(bn(y)) on the kernel of csrc/a2s_linear.hip (dz as transposed fp16 term planes, activations staged transposed, split-K over
    the rows with alternating accumulation sign) against float64 and against the generic split-K two-term tile; G starts non-zero."""
    import os
    from piano_a2s_amd import hip
    L = hip.lib()
    K, N = channels * period, 256
    assert L.a2s_linear_wgrad_eligible(M, N, K, period) == 1
    g = torch.Generator().manual_seed(M + K)
    y = (torch.randn(M, K, generator=g) * 2.0).to(dev)
    dz = (torch.randn(M, N, generator=g) * gscale).to(dev)
    dz[M // 3] *= 2.0 ** -12
    scale = (torch.randn(channels, generator=g)).to(dev)
    shift = (torch.randn(channels, generator=g) * 0.3).to(dev)
    ymax = y.view(M, channels, period).abs().amax(dim=(0, 2)).contiguous()
    bound, dmax = hip.act_bound(scale, shift, ymax), hip.absmax(dz)
    G0 = (torch.randn(N, K, generator=g) * gscale).to(dev)
    G_new = G0.clone()
    assert hip.linear_wgrad(dz, y, (scale, shift, period), dmax, bound, G_new)
    G_old = G0.clone()
    sk = L.a2s_gemm_pick_splitk(N, K, M, 1)
    hip.gemm(dz, 1, N, y, K, 1, G_old, K, N, K, M, beta=1.0, splitk=sk, b_affine=(scale, shift, period), two_term=(dmax, bound))
    torch.cuda.synchronize()
    ch = torch.arange(K, device=dev) // period
    a = torch.relu(y.double() * scale.double()[ch] + shift.double()[ch])
    ref = G0.double() + dz.double().t() @ a
    mag = G0.double().abs() + dz.double().abs().t() @ a.abs()
    assert torch.isfinite(G_new).all()
    err_new = float(((G_new.double() - ref).abs() / mag.clamp_min(1e-300)).max())
    err_old = float(((G_old.double() - ref).abs() / mag.clamp_min(1e-300)).max())
    assert err_new < 1e-6, f"{err_new:.3e} of sum|dz||a| (generic tile: {err_old:.3e})"
    assert err_new <= 2.0 * err_old + 1e-7, f"{err_new:.3e} vs the generic tile's {err_old:.3e}"
