"""The non-GEMM layers of the depth network (csrc/nn_misc.hip) against plain torch fp32 / fp64 compositions of the same ops:
stem max-pool forward + index-free backward (incl. the ties a ReLU produces), eval-mode BatchNorm with a trainable affine
behind a convolution (the `downsample.1` case of online_adaption.py:182-184 and the general unfrozen case with residual +
ReLU), the single-channel scale layer, and the stand-alone upsample + concat helper."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CL = torch.channels_last


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 24, 32), (1, 16, 7, 9), (1, 64, 240, 320)])
def test_maxpool_fwd_bwd_matches_torch(B, C, H, W):
    from e2ehip import nn_ops
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(B, C, H, W, generator=g))               # many exact-zero ties, as behind the stem's ReLU
    x[:, :, ::5] = x[:, :, 1::5][:, :, : x[:, :, ::5].shape[2]]          # equal non-zero neighbours too
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.to(DEV).contiguous(memory_format=CL).requires_grad_(True)
    yd = nn_ops.max_pool_3x3_s2(xd)
    yd.backward(gy.to(DEV))
    assert torch.equal(yd.cpu(), yr.detach())
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=0, atol=0)    # sums of at most 4 identical addends in window order... exact here


@pytest.mark.parametrize("B,C,H,W,acc,relu", [(2, 64, 24, 32, 0, 1), (1, 16, 7, 9, 1, 0), (1, 64, 240, 320, 0, 1)])
def test_maxpool_with_kept_positions_matches_torch(B, C, H, W, acc, relu):
    """e2e_maxpool3x3s2_fwd_idx / _bwd_idx (the launch plan's pair): same values and gradients as ATen, with the accumulate flag and the
    fused ReLU mask of the producer (d/d pre-activation of the stem)."""
    from e2ehip import _lib as L
    g = torch.Generator().manual_seed(3)
    x = torch.relu(torch.randn(B, C, H, W, generator=g))
    x[:, :, ::5] = x[:, :, 1::5][:, :, : x[:, :, ::5].shape[2]]
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    want = xr.grad * (x > 0) if relu else xr.grad
    base = torch.randn(x.shape, generator=g)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)                      # NHWC
    Ho, Wo = yr.shape[2:]
    yd = torch.empty(B, Ho, Wo, C, device=DEV)
    idx = torch.empty(B, Ho, Wo, C, device=DEV, dtype=torch.uint8)
    dx = base.permute(0, 2, 3, 1).contiguous().to(DEV)
    L.call("e2e_maxpool3x3s2_fwd_idx", L.ptr(xd), L.ptr(yd), L.ptr(idx), B, H, W, C, L.stream())
    L.call("e2e_maxpool3x3s2_bwd_idx", L.ptr(xd), L.ptr(idx), L.ptr(gy.permute(0, 2, 3, 1).contiguous().to(DEV)), L.ptr(dx), B, H, W, C, acc, relu, L.stream())
    assert torch.equal(yd.permute(0, 3, 1, 2).cpu(), yr.detach())
    torch.testing.assert_close(dx.permute(0, 3, 1, 2).cpu(), want + base if acc else want, rtol=0, atol=1e-6 if acc else 0)


@pytest.mark.parametrize("relu,res", [(False, False), (True, True), (True, False)])
def test_trainable_eval_bn_behind_conv(relu, res):
    """BN(conv(x)) [+ residual] [ReLU] with eval statistics and TRAINABLE gamma / beta: values and all four gradients."""
    from e2ehip import nn_ops
    g = torch.Generator().manual_seed(2)
    B, Cin, Cout, H, W, s = 2, 64, 128, 13, 18, 2
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    rm, rv = torch.randn(Cout, generator=g) * 0.2, torch.rand(Cout, generator=g) + 0.3
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    r = torch.randn(B, Cout, Ho, Wo, generator=g) if res else None
    gy = torch.randn(B, Cout, Ho, Wo, generator=g)

    def run(dev, dt):
        t = [v.to(dev, dt).requires_grad_(True) for v in (x, w, gamma, beta)] + ([r.to(dev, dt).requires_grad_(True)] if res else [])
        if dev == "cpu":
            y = F.batch_norm(F.conv2d(t[0], t[1], None, s, 0), rm.to(dt), rv.to(dt), t[2], t[3], False, 0.0, 1e-5)
            if res:
                y = y + t[4]
            y = F.relu(y) if relu else y
        else:
            y = nn_ops.conv2d(t[0].contiguous(memory_format=CL), t[1], None, s, 0, "zeros", "relu" if relu else None,
                              (t[2], t[3], rm.to(dev), rv.to(dev), 1e-5), residual=t[4] if res else None)
        y.backward(gy.to(dev, dt))
        return [y.detach()] + [v.grad for v in t]
    ref, got = run("cpu", torch.float64), run(DEV, torch.float32)
    for a, b, name in zip(got, ref, ["y", "dx", "dw", "dgamma", "dbeta", "dres"]):
        scale = float(b.abs().max()) + 1e-12
        assert float((a.cpu().double() - b).abs().max()) / scale < 2e-5, name


def test_scale_layers_match_torch():
    from depth_estimation.networks import Conv1x1, ScaleLayer
    g = torch.Generator().manual_seed(3)
    d = torch.rand(2, 1, 24, 40, generator=g) + 0.5
    gy = torch.randn(2, 1, 24, 40, generator=g)
    for make in (lambda: ScaleLayer(0.7), lambda: Conv1x1(1, 1, init_value=0.6, bias=True), lambda: Conv1x1(1, 1, init_value=0.6, bias=False)):
        m = make().to(DEV)
        xd = d.to(DEV).requires_grad_(True)
        y = m(xd)
        y.backward(gy.to(DEV))
        ps = list(m.parameters())
        w = ps[0].detach().cpu().reshape(())
        b = ps[1].detach().cpu().reshape(()) if len(ps) > 1 else torch.zeros(())
        torch.testing.assert_close(y.detach().cpu(), d * w + b, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(xd.grad.cpu(), gy * w, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(ps[0].grad.cpu().reshape(()), (gy.double() * d.double()).sum().float(), rtol=1e-5, atol=1e-5)
        if len(ps) > 1:
            torch.testing.assert_close(ps[1].grad.cpu().reshape(()), gy.double().sum().float(), rtol=1e-5, atol=1e-5)


def test_unsupported_convolution_raises_instead_of_falling_back():
    from e2ehip import nn_ops
    x = torch.randn(1, 8, 8, 8, device=DEV)
    with pytest.raises(NotImplementedError):
        nn_ops.conv2d(x, torch.randn(5, 8, 3, 3, device=DEV), None, 1, 1)       # Cout = 5: no kernel, and no library fallback


def test_upsample_helper_matches_torch():
    from depth_estimation.networks import upsample
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 32, 6, 10, generator=g)
    xd = x.to(DEV).requires_grad_(True)
    y = upsample(xd)
    gy = torch.randn(2, 32, 12, 20, generator=g)
    y.backward(gy.to(DEV))
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode="nearest")
    yr.backward(gy)
    assert torch.equal(y.detach().cpu(), yr.detach())
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-6, atol=1e-6)
