"""The split-operand (16-bit matrix pipe) kernels under NON-benign operand scales (reference models.py:499-505:
BatchNorm gamma is a free parameter, so the activations a trained net hands to a convolution can sit anywhere in fp32's range).

Every case runs the product's default path and the fp32-input MFMA kernel (`conv_bf16x3` = 0 / `gemm_bf16x3` = 0: bit-for-bit an fmaf
chain) on the same data and measures both against float64, relative to sum |a||b| -- what an fp32 dot product may lose.  The bar: the
default path's error is at most 3x the fp32-input kernel's (+ a 1e-7 allowance), whatever the scale.  Cases: BatchNorm scale 1e-3 and
1e+3, activations beyond fp16's 65504, one clip whose gradient is 1e6 times the others' (error measured on the SMALL clips alone)."""
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


def _report(name, err):
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/robustness_errors.txt", "a") as f:
        f.write(f"{name}: {err:.3e}\n")


class _switch:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        from piano_a2s_amd import hip
        self.L = hip.lib()
        self.prev = {k: self.L.a2s_debug_get(k.encode()) for k in self.kv}
        for k, v in self.kv.items():
            hip.check(self.L.a2s_debug_set(k.encode(), v), "debug_set")

    def __exit__(self, *a):
        for k, v in self.prev.items():
            self.L.a2s_debug_set(k.encode(), v)


def _conv_fwd(dev, x, w, scale, shift):
    from piano_a2s_amd import hip
    L = hip.lib()
    B, T, Cin, F = x.shape
    Cout = w.shape[0]
    y = torch.full((B, T, Cout, F), float("nan"), device=dev)
    part = torch.zeros(L.a2s_conv3x3_stat_blocks(B, T, F, Cin), Cout, 2, device=dev)
    xd, wd, scd, shd = x.to(dev), w.to(dev), scale.to(dev), shift.to(dev)
    cws = hip.conv_workspace(Cin, dev)
    hip.conv3x3_forward(xd, wd, y, scd, shd, part, cws)
    torch.cuda.synchronize()
    return y.cpu().double(), part.cpu().double().sum(0)


@pytest.mark.parametrize("Cin,Cout", [(20, 20), (20, 40), (40, 40)])
@pytest.mark.parametrize("bn_scale", [1e-3, 1.0, 1e3, 3e5])
def test_conv_forward_under_batchnorm_scales(dev, Cin, Cout, bn_scale):
    """relu(bn(x)) with gamma ~ bn_scale: activations of order 1e-3 ... 3e5 (the last beyond fp16's largest number); per-channel scales
    spread over a further factor 64 (channels of a trained net do not share one magnitude)."""
    g = torch.Generator().manual_seed(Cin + Cout)
    B, T, F = 2, 13, 132
    x = torch.randn(B, T, Cin, F, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    spread = torch.exp2(torch.randint(-3, 4, (Cin,), generator=g).float())
    scale = bn_scale * spread * (torch.rand(Cin, generator=g) + 0.5)
    shift = bn_scale * spread * torch.randn(Cin, generator=g) * 0.3
    a64 = torch.relu(x.double().permute(0, 2, 1, 3) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    ref = torch.nn.functional.conv2d(a64, w.double(), padding=1).permute(0, 2, 1, 3)
    mag = torch.nn.functional.conv2d(a64.abs(), w.double().abs(), padding=1).permute(0, 2, 1, 3) + 1e-300
    errs = {}
    for name, sw in (("fp32_input", dict(conv_bf16x3=0)), ("default", {})):
        with _switch(**sw):
            y, sums = _conv_fwd(dev, x, w, scale, shift)
        assert torch.isfinite(y).all(), name
        errs[name] = float(((y - ref).abs() / mag).max())
        s_err = float((sums[:, 0] - ref.sum(dim=(0, 1, 3))).abs().max() / mag.sum(dim=(0, 1, 3)).max())
        assert s_err < 1e-6, (name, "batch-statistics sums", s_err)
    _report(f"conv fwd {Cin}->{Cout} bn_scale {bn_scale:g}: fp32-input / default", errs["fp32_input"])
    _report(f"conv fwd {Cin}->{Cout} bn_scale {bn_scale:g}: default", errs["default"])
    assert errs["default"] <= 3 * errs["fp32_input"] + 1e-7, errs


@pytest.mark.parametrize("Cin,Cout", [(40, 40), (40, 20), (20, 20)])
def test_conv_data_gradient_with_one_outlier_clip(dev, Cin, Cout):
    """dy of clip 0 is 1e6 x the other clips': the error of the OTHER clips' data gradient, each relative to its own sum |dy||w|."""
    from piano_a2s_amd import hip
    g = torch.Generator().manual_seed(Cin * 3 + Cout)
    B, T, F = 3, 11, 100
    dy = 1e-6 * torch.randn(B, T, Cin, F, generator=g) * torch.exp(torch.randn(B, T, Cin, F, generator=g))
    dy[0] *= 1e6
    w = torch.randn(Cin, Cout, 3, 3, generator=g) * 0.1                       # (layer Cout = dy channels, layer Cin = dx channels)
    w64 = w.double().transpose(0, 1).flip(2, 3)
    ref = torch.nn.functional.conv2d(dy.double().permute(0, 2, 1, 3), w64, padding=1).permute(0, 2, 1, 3)
    mag = torch.nn.functional.conv2d(dy.double().abs().permute(0, 2, 1, 3), w64.abs(), padding=1).permute(0, 2, 1, 3) + 1e-300
    yl = torch.randn(B, T, Cout, F, generator=g)
    errs = {}
    for name, sw in (("fp32_input", dict(conv_bf16x3=0)), ("default", {})):
        with _switch(**sw):
            dx = hip.conv3x3_dgrad_for_test(dy.to(dev), w.to(dev), yl.to(dev))
        dx = dx.cpu().double()
        assert torch.isfinite(dx).all()
        errs[name] = [float(((dx[b] - ref[b]).abs() / mag[b]).max()) for b in range(B)]
    for b in range(B):
        _report(f"conv dgrad {Cin}->{Cout} outlier clip: clip {b} fp32-input", errs["fp32_input"][b])
        _report(f"conv dgrad {Cin}->{Cout} outlier clip: clip {b} default", errs["default"][b])
    for b in range(B):
        assert errs["default"][b] <= 3 * errs["fp32_input"][b] + 1e-7, (b, errs)


@pytest.mark.parametrize("Cin,Cout", [(40, 40), (20, 40), (20, 20)])
@pytest.mark.parametrize("bn_scale", [1e-3, 1e3])
def test_conv_weight_gradient_with_outlier_clip_and_batchnorm_scales(dev, Cin, Cout, bn_scale):
    """dW = sum over clips; clip 0's dy is 1e6 x the others', the activations are of order bn_scale.  The error of the SMALL clips'
    contribution is measured by running the weight gradient on those clips alone (the sum over all clips is dominated by clip 0 in any
    arithmetic); the full sum is held to the same bar relative to its own sum |dy||a|."""
    from piano_a2s_amd import hip
    g = torch.Generator().manual_seed(Cin + 7 * Cout)
    B, T, F = 3, 13, 100
    x = torch.randn(B, T, Cin, F, generator=g)
    dy = 1e-6 * torch.randn(B, T, Cout, F, generator=g) * torch.exp(torch.randn(B, T, Cout, F, generator=g))
    dy[0] *= 1e6
    scale = bn_scale * (torch.rand(Cin, generator=g) + 0.5)
    shift = bn_scale * torch.randn(Cin, generator=g) * 0.3
    a64 = torch.relu(x.double().permute(0, 2, 1, 3) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    g64 = dy.double().permute(0, 2, 1, 3)
    for name, sl in (("all clips", slice(0, B)), ("small clips", slice(1, B))):
        ref = torch.nn.grad.conv2d_weight(a64[sl], (Cout, Cin, 3, 3), g64[sl], padding=1)
        mag = torch.nn.grad.conv2d_weight(a64[sl].abs(), (Cout, Cin, 3, 3), g64[sl].abs(), padding=1) + 1e-300
        errs = {}
        for kname, sw in (("fp32_input", dict(wgrad_bf16x3=0, wgrad_f16x2=0)), ("default", {})):
            with _switch(**sw):
                dW = hip.conv3x3_wgrad_for_test(dy[sl].contiguous().to(dev), x[sl].contiguous().to(dev), scale.to(dev), shift.to(dev))
            dW = dW.cpu().double()
            assert torch.isfinite(dW).all()
            errs[kname] = float(((dW - ref).abs() / mag).max())
        _report(f"conv wgrad {Cin}->{Cout} bn_scale {bn_scale:g} {name}: fp32-input", errs["fp32_input"])
        _report(f"conv wgrad {Cin}->{Cout} bn_scale {bn_scale:g} {name}: default", errs["default"])
        assert errs["default"] <= 3 * errs["fp32_input"] + 1e-7, (name, errs)


@pytest.mark.parametrize("bn_scale", [1e-3, 1e3, 3e5])
def test_linear_forward_under_batchnorm_scales(dev, bn_scale):
    """The 19200 -> 256 Linear's forward form (operand BatchNorm+ReLU while staging) with activations of order bn_scale."""
    from piano_a2s_amd import hip
    g = torch.Generator().manual_seed(11)
    M, K, N, period = 12800 + 37, 1920, 256, 48
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.007).to(dev)
    aff = ((bn_scale * (torch.rand(K // period, generator=g) + 0.5)).to(dev), (bn_scale * torch.randn(K // period, generator=g) * 0.3).to(dev), period)
    xa = torch.relu(x.double() * aff[0].double().repeat_interleave(period) + aff[1].double().repeat_interleave(period))
    ref, mag = xa @ w.double().t(), xa.abs() @ w.double().abs().t() + 1e-300
    errs = {}
    for name, sw in (("fp32_input", dict(gemm_bf16x3=0)), ("default", {})):
        with _switch(**sw):
            y = hip.linear_forward_for_test(x, w, aff)
        torch.cuda.synchronize()
        assert torch.isfinite(y).all()
        errs[name] = float(((y.double() - ref).abs() / mag).max())
    _report(f"linear fwd bn_scale {bn_scale:g}: fp32-input", errs["fp32_input"])
    _report(f"linear fwd bn_scale {bn_scale:g}: default", errs["default"])
    assert errs["default"] <= 3 * errs["fp32_input"] + 1e-7, errs
