"""Off-by-default losses against the CPU oracle: values at 1e-5 relative, gradients against the
oracle's autograd; the three losses the reference defines in loss/losses.py are additionally pinned by the golden
fixture g5_aux (tests/test_oracle_golden.py checks the oracle against it)."""
import pytest
import torch

from oracle import warp_loss as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, seed=0, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return lo + (hi - lo) * torch.rand(*shape, generator=g)


@pytest.mark.parametrize("B,H,W", [(1, 48, 64), (2, 37, 53), (1, 480, 640)])
def test_smoothness_value_and_grad(B, H, W):
    from e2ehip import ops
    disp = _rand(B, 1, H, W, seed=1, lo=0.05, hi=2.0)
    img = _rand(B, H, W, 3, seed=2).permute(0, 3, 1, 2)                  # the reference's NCHW view of NHWC memory
    dr = disp.clone().requires_grad_(True)
    ref = O.smoothness(dr, img)
    ref.backward()
    dg = disp.to(DEV).requires_grad_(True)
    out = ops.smoothness(dg, img.to(DEV))
    (out * 3.0).backward()
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(dg.grad.cpu() / 3.0, dr.grad, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("n_valid", [20000, 5000])
def test_geometric_consistency_gate_value_and_grad(n_valid):
    from e2ehip import ops
    H, W = 120, 200
    a = _rand(1, 1, H, W, seed=3, lo=0.5, hi=4.0)
    b = a + 0.3 * (_rand(1, 1, H, W, seed=4) - 0.5)
    b[0, 0, :4] = 9.0 * a[0, 0, :4]                                     # |a-b|/(a+b) = 0.8: inside the clamp
    b[0, 0, 4:6] = -0.5 * a[0, 0, 4:6]                                  # |a-b|/(a+b) = 3: clamped to 1, zero gradient
    valid = torch.zeros(1, 1, H, W)
    valid.view(-1)[:n_valid] = 1.0
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = O.geometric_consistency(ar, br, valid)
    ag, bg = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    out = ops.geometric_consistency(ag, bg, valid.to(DEV))
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-8)
    out.backward()
    if n_valid > 10000:
        ref.backward()
        torch.testing.assert_close(ag.grad.cpu(), ar.grad, rtol=1e-4, atol=1e-10)
        torch.testing.assert_close(bg.grad.cpu(), br.grad, rtol=1e-4, atol=1e-10)
    else:                                                                 # gate closed: constant 0, no gradient
        assert float(out.detach()) == 0.0 and float(ag.grad.abs().max()) == 0.0 and float(bg.grad.abs().max()) == 0.0


def test_depth_gt_loss_value_and_grad():
    from e2ehip import ops
    H, W = 240, 320
    p = _rand(1, 1, H, W, seed=5, lo=0.3, hi=5.0)
    mask = (_rand(H, W, seed=6) < 0.1).float()
    gt = (p[0, 0] + 0.2 * (_rand(H, W, seed=7) - 0.5)) * mask
    pr = p.clone().requires_grad_(True)
    ref = O.depth_gt(pr, gt, mask)
    ref.backward()
    pg = p.to(DEV).requires_grad_(True)
    out = ops.masked_l1(pg, gt.to(DEV), mask.to(DEV))
    out.backward()
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(pg.grad.cpu(), pr.grad, rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("C", [1, 2, 4])
def test_min_reprojection_value_and_grad(C):
    from e2ehip import ops
    e = _rand(2, C, 60, 80, seed=8 + C)
    if C > 1:
        e[0, 0, :5] = -1.0
        e[0, 1, :5] = -1.0                                               # ties below everything else: the FIRST minimal channel takes the gradient
    er = e.clone().requires_grad_(True)
    ref = O.min_reprojection(er)
    eg = e.to(DEV).requires_grad_(True)
    out = ops.min_reprojection(eg)
    out.backward()
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-8)
    g = eg.grad.cpu()
    n = 2 * 60 * 80
    assert torch.allclose(g.sum(1), torch.full((2, 60, 80), 1.0 / n))   # one channel per pixel carries 1/n
    amin = e.min(1, keepdim=True)[0]
    assert bool(((g > 0) <= (e == amin)).all())                          # ... and it is a minimal one
    if C > 1:
        assert float(g[0, 1, :5].abs().max()) == 0.0 and bool((g[0, 0, :5] > 0).all())


def test_process_disparity_value_and_grad():
    from e2ehip import ops
    d = _rand(2, 1, 48, 64, seed=20, lo=0.01, hi=10.0)
    dr = d.clone().requires_grad_(True)
    ref = O.process_disparity(dr)
    w = _rand(1, 1, 48, 64, seed=21)
    (ref * w).sum().backward()
    dg = d.to(DEV).requires_grad_(True)
    out = ops.process_disparity(dg)
    (out * w.to(DEV)).sum().backward()
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(dg.grad.cpu(), dr.grad, rtol=1e-5, atol=1e-7)


def test_dropin_losses_module_uses_the_kernels():
    import loss.losses as LL
    H, W = 40, 56
    disp = _rand(1, 1, H, W, seed=30, lo=0.1, hi=1.0).to(DEV).requires_grad_(True)
    img = _rand(1, 3, H, W, seed=31).to(DEV)
    v = LL.disparity_smoothness_loss(disp, img)
    v.backward()
    assert disp.grad is not None and torch.isfinite(disp.grad).all()
    outs = {("warped_depth", 1): _rand(1, 1, 200, 100, seed=32, lo=1, hi=2).to(DEV), ("interpolated_depth", 1): _rand(1, 1, 200, 100, seed=33, lo=1, hi=2).to(DEV),
            ("valid_mask", 1): torch.ones(1, 1, 200, 100, device=DEV)}
    g = LL.geometric_consistency_loss(outs, 1, DEV)
    ref = O.geometric_consistency(outs[("warped_depth", 1)].cpu(), outs[("interpolated_depth", 1)].cpu(), outs[("valid_mask", 1)].cpu())
    torch.testing.assert_close(g.cpu(), ref, rtol=1e-5, atol=1e-8)
    with pytest.raises(Exception):
        LL.disparity_smoothness_loss(disp.detach().cpu(), img.cpu())     # no CPU fallback
