"""median / scale chain / regulariser / metrics / Adam kernels against the oracle (torch CPU)."""
import pytest
import torch

from oracle import warp_loss

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 614400])
def test_median_lower_exact(n):
    from e2ehip import ops
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, generator=g) * 3 + 0.5
    if n > 10:
        x[: n // 3] = x[0]                  # heavy ties
        x[5] = -2.0
        x[6] = 0.0
    got = ops.median_lower(x.to(DEV)).cpu()
    assert torch.equal(got, torch.median(x)), (got, torch.median(x))


def test_depth_scale_chain_vs_oracle():
    from e2ehip import ops
    g = torch.Generator().manual_seed(0)
    F_, H, W = 2, 40, 56
    disp = torch.rand(F_, 1, H, W, generator=g) * 0.5 + 0.3
    gt = torch.rand(1, F_, H, W, 1, generator=g) * 2 + 1
    wgt = torch.rand(F_, 1, H, W, generator=g)
    dc = disp.clone().requires_grad_(True)
    depths = [1 / dc[i:i + 1] for i in range(F_)]
    scaled, ratio = warp_loss.median_scale(depths, gt)
    (torch.cat(scaled, 0) * wgt).sum().backward()
    dg = disp.to(DEV).requires_grad_(True)
    mgt = ops.median_lower(gt.to(DEV))
    assert torch.equal(mgt.cpu(), torch.median(gt))
    depth, delta, r = ops.depth_from_disp_median_scaled(dg, mgt)
    (depth * wgt.to(DEV)).sum().backward()
    torch.testing.assert_close(r.cpu(), ratio.detach(), rtol=1e-6, atol=0)
    torch.testing.assert_close(depth.detach().cpu(), torch.cat(scaled, 0).detach(), rtol=1e-6, atol=0)
    torch.testing.assert_close(delta.cpu(), (1 / disp), rtol=1e-6, atol=0)
    err = (dg.grad.cpu() - dc.grad).abs().max() / dc.grad.abs().max()
    assert err < 1e-5, err
    # the median element carries the extra term: it must be the largest-magnitude difference from rho*g chain
    k = int(torch.argmax((dc.grad + (1 / disp) ** 2 * ratio.detach() * wgt).abs()))
    assert (dg.grad.cpu().flatten()[k] - dc.grad.flatten()[k]).abs() < 1e-4 * dc.grad.abs().max()


@pytest.mark.parametrize("kind", ["l1", "l2"])
def test_mean_diff_vs_golden(golden, kind):
    from e2ehip import ops
    g = golden("g5_aux")
    d1 = g["d1"].to(DEV).requires_grad_(True)
    out = ops.mean_diff(g["d0"].to(DEV), d1, kind)
    torch.testing.assert_close(out.cpu(), g[f"reg_{kind}"], rtol=1e-5, atol=1e-7)
    out.backward()
    if kind == "l2":
        torch.testing.assert_close(d1.grad.cpu(), g["greg_l2"], rtol=1e-5, atol=1e-9)
    with pytest.raises(ValueError):
        ops.mean_diff(g["d0"].to(DEV), d1, "huber")


def test_depth_metrics_vs_golden(golden):
    from e2ehip import ops
    g = golden("g5_aux")
    torch.testing.assert_close(ops.depth_metrics(g["gt"].to(DEV), g["pred"].to(DEV), False).cpu(), g["metrics_icl"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(ops.depth_metrics(g["gt_holes"].to(DEV), g["pred"].to(DEV), True).cpu(), g["metrics_tum"], rtol=1e-5, atol=1e-7)


def test_fused_adam_vs_torch():
    from e2ehip.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(64, 3, 7, 7), (64,), (128, 64, 3, 3), (5,), (1, 16, 3, 3)]
    ref = [torch.randn(s).requires_grad_(True) for s in shapes]
    frozen = torch.randn(7)                        # requires_grad False: never touched
    nograd = torch.randn(9).requires_grad_(True)   # grad None: skipped like torch.optim.Adam does
    mine = [r.detach().clone().to(DEV).requires_grad_(True) for r in ref]
    fz, ng = frozen.clone().to(DEV), nograd.detach().clone().to(DEV).requires_grad_(True)
    o_ref = torch.optim.Adam(ref + [nograd], lr=1e-5)
    o_me = FusedAdam(mine + [fz, ng], lr=1e-5)
    for step in range(4):
        o_ref.zero_grad(); o_me.zero_grad()
        gs = [torch.randn(s) * (0.1 + step) for s in shapes]
        for p, q, gg in zip(ref, mine, gs):
            (p * gg).sum().backward()
            (q * gg.to(DEV)).sum().backward()
        o_ref.step(); o_me.step()
        for p, q in zip(ref, mine):
            torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=1e-6, atol=1e-9)
    assert torch.equal(fz.cpu(), frozen) and torch.equal(ng.detach().cpu(), nograd.detach())
    # the first update is lr * sign(g): a direct check of the bias-corrected step size
    p0 = torch.randn(10).to(DEV).requires_grad_(True)
    start = p0.detach().clone()
    o = FusedAdam([p0], lr=1e-3)
    (p0 * torch.arange(1, 11, device=DEV).float()).sum().backward()
    o.step()
    torch.testing.assert_close(start - p0.detach(), torch.full((10,), 1e-3, device=DEV), rtol=1e-4, atol=0)
