"""The first ConvStack layer's compile-time-shaped kernels (csrc/a2s_conv.hip: conv3x3_c1_fixed, conv3x3_wgrad_c1_stream; round 6) against the
generic Cin = 1 kernels they replace at the model's shape (20 channels x 480 bins): the same order of operations per output, per statistic and
per tap sum, so every result is BIT-identical -- output, batch-statistics partials, per-channel max |y|, weight gradient, the BatchNorm input
gradient written out.  Sizes where workgroups walk several rows (the streaming weight gradient prefetches the next row while it sums the
current one).  Reference: nn.Conv2d(1, 20, 3, padding=1) + BatchNorm2d of ConvStack, models.py:525-534."""
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


def _both(L, hip, run):
    prev = L.a2s_debug_get(b"conv_c1_fast")
    out = []
    try:
        for fast in (0, 1):
            hip.check(L.a2s_debug_set(b"conv_c1_fast", fast), "debug_set")
            out.append(run())
    finally:
        hip.check(L.a2s_debug_set(b"conv_c1_fast", prev), "debug_set")
    return out


@pytest.mark.parametrize("B,T", [(1, 5), (3, 301), (16, 1201)])
def test_first_layer_forward_fixed_shape_kernel_is_bit_identical(dev, B, T):
    from piano_a2s_amd import hip
    L = hip.lib()
    F, Cout = 480, 20
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B, T, 1, F, generator=g).to(dev)
    w = (torch.randn(Cout, 1, 3, 3, generator=g) * 0.3).to(dev)
    nblk = L.a2s_conv3x3_stat_blocks(B, T, F, 1)

    def run():
        y = torch.full((B, T, Cout, F), 7.0, device=dev)
        part = torch.zeros(nblk, Cout, 2, device=dev)
        amax = torch.zeros(Cout, device=dev)
        hip.check(L.a2s_conv3x3_ranged(hip.stream(), hip._p(x), hip._p(w), hip._p(y), NULL, NULL, NULL, hip._p(part), hip._p(amax), B, T, F, 1, Cout, NULL),
                  "conv3x3_ranged")
        torch.cuda.synchronize()
        return y, part, amax
    (y0, p0, a0), (y1, p1, a1) = _both(L, hip, run)
    ref = torch.nn.functional.conv2d(x.permute(0, 2, 1, 3), w, padding=1).permute(0, 2, 1, 3)
    assert float((y1 - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert torch.equal(y0, y1) and torch.equal(p0, p1) and torch.equal(a0, a1)
    assert torch.equal(a1, y1.abs().amax(dim=(0, 1, 3)))


@pytest.mark.parametrize("B,T,write_dy", [(1, 7, True), (3, 301, True), (5, 401, False)])
def test_first_layer_weight_gradient_streaming_kernel_is_bit_identical(dev, B, T, write_dy):
    """768 persistent workgroups walk the B * T rows: 903 / 2005 rows = two and three rows for some of them."""
    from piano_a2s_amd import hip
    L = hip.lib()
    F, Cout = 480, 20
    g0 = torch.Generator().manual_seed(B * 77 + T)
    gact = torch.randn(B, T, Cout, F, generator=g0).to(dev)
    y = torch.randn(B, T, Cout, F, generator=g0).to(dev)
    x = torch.randn(B, T, 1, F, generator=g0).to(dev)
    mean, invstd = (torch.randn(Cout, generator=g0) * 0.1).to(dev), (torch.rand(Cout, generator=g0) + 0.5).to(dev)
    gamma, beta = (torch.rand(Cout, generator=g0) + 0.5).to(dev), (torch.randn(Cout, generator=g0) * 0.1).to(dev)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    c12 = (torch.randn(2 * Cout, generator=g0) * 0.01).to(dev)
    nb = L.a2s_conv3x3_wgrad_workspace_bytes(1, Cout)
    ws = torch.empty(nb // 4, device=dev)

    def run():
        dy = torch.full_like(gact, 7.0) if write_dy else None
        dW = torch.zeros(Cout, 1, 3, 3, device=dev)
        hip.check(L.a2s_conv3x3_wgrad_bn(hip.stream(), hip._p(gact), hip._p(y), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), hip._p(c12),
                                         hip._p(dy), hip._p(x), NULL, NULL, hip._p(dW), hip._p(ws), C.c_size_t(nb), B, T, F, 1, Cout), "wgrad_bn")
        torch.cuda.synchronize()
        return dW, dy
    (w0, d0), (w1, d1) = _both(L, hip, run)
    assert torch.equal(w0, w1)
    if write_dy:
        assert torch.equal(d0, d1)
    # and against the definition, in double
    gm = torch.where(y * scale.view(1, 1, -1, 1) + shift.view(1, 1, -1, 1) > 0, gact, torch.zeros_like(gact)).double()
    xhat = (y.double() - mean.double().view(1, 1, -1, 1)) * invstd.double().view(1, 1, -1, 1)
    dz = scale.double().view(1, 1, -1, 1) * (gm - c12[0::2].double().view(1, 1, -1, 1) - xhat * c12[1::2].double().view(1, 1, -1, 1))
    xp = torch.nn.functional.pad(x[:, :, 0].double(), (1, 1, 1, 1))
    ref = torch.stack([torch.stack([(dz * xp[:, dt:dt + T, None, df:df + F]).sum(dim=(0, 1, 3)) for df in range(3)], -1) for dt in range(3)], -2)
    assert float((w1[:, 0].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
