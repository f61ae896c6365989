"""HIP warp / photometric kernels (through the C ABI) against the CPU oracle and the golden vectors."""
import pytest
import torch

from oracle import warp_loss
from synth import grad_mismatch, make_pair

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# fp32 tolerance named by BASELINE.json north_star: 1e-4 relative on depth / loss tensors
RTOL = 1e-4


def _ops():
    from e2ehip import ops
    return ops


def _close(a, b, rtol=RTOL, atol=1e-6, what=""):
    torch.testing.assert_close(a.detach().cpu(), b.detach().cpu(), rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")


def _img_close(a, b, what=""):
    """image-valued tensors in [0,1]: 1e-4 of the tensor's scale (bilinear taps at ix ~ 1e2..1e3 carry
    fp32 coordinate rounding of ~1e-5 px times the local image slope)."""
    a, b = a.detach().cpu(), b.detach().cpu()
    err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)
    assert err < RTOL, f"{what}: max err / max|ref| = {err:.3e}"


def _grad_close(a, b, rel=1e-3, what=""):
    """gradients: compare against the tensor's own scale (tiny entries carry cancellation noise)."""
    a, b = a.detach().cpu(), b.detach().cpu()
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item() / scale
    assert err < rel, f"{what}: max err / max|ref| = {err:.3e}"


def _gdepth_close(a, b, grid, what=""):
    err, skipped = grad_mismatch(a, b, grid)
    assert err < 1e-3 and skipped < 0.01, f"{what}: max err / max|ref| = {err:.3e} ({skipped:.2%} boundary pixels skipped)"


def _mask_close(a, b, grid, what=""):
    """0/1 validity masks must agree except where |grid| sits within rounding of the boundary 1.0."""
    a, b = a.detach().cpu(), b.detach().cpu()
    diff = (a != b)
    if diff.any():
        edge = (grid.abs().max(-1)[0] - 1).abs().unsqueeze(1)
        assert (edge[diff] < 1e-5).all(), f"{what}: mask differs away from the boundary"
        assert diff.sum() <= 4


@pytest.mark.parametrize("tag", ["a", "b"])
def test_modular_ops_vs_golden(golden, tag):
    ops = _ops()
    g = golden(f"g1_warp_{tag}")
    H, W = g["depth"].shape[2:]
    d = g["depth"].to(DEV).requires_grad_(True)
    cam = ops.backproject(d, g["invK"].to(DEV))
    _close(cam, g["cam"], what="cam")
    grid, valid = ops.project3d(cam, g["K"].to(DEV), g["T"].to(DEV), H, W)
    _close(grid, g["grid"], atol=2e-6, what="grid")
    _mask_close(valid, g["valid"], g["grid"], "valid")
    src = g["src"].to(DEV).permute(0, 3, 1, 2)
    tgt = g["tgt"].to(DEV).permute(0, 3, 1, 2)
    for pad in ("border", "zeros"):
        synth = ops.grid_sample(src, grid, padding_mode=pad, align_corners=False)
        _img_close(synth, g[f"{pad}_synth"], what=f"synth {pad}")
        pm = ops.photometric(synth * valid, tgt * valid)
        _img_close(pm, g[f"{pad}_pmap"], what=f"pmap {pad}")
        loss = pm.mean(1, keepdim=True).mean()
        _close(loss, g[f"{pad}_loss"], what=f"loss {pad}")
        gd, = torch.autograd.grad(loss, d, retain_graph=True)
        _gdepth_close(gd, g[f"{pad}_gdepth"], g["grid"], what=f"gdepth {pad}")
    gridg, zg, _ = ops.project3d(cam, g["K"].to(DEV), g["T"].to(DEV), H, W, geometric=True)
    _close(zg, g["geo_depth"], what="geo depth")
    _close(ops.grid_sample(src, gridg, padding_mode="border", align_corners=True), g["geo_synth_ac"], atol=1e-5, what="align_corners")


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("pad", ["border", "zeros"])
def test_fused_vs_golden(golden, tag, pad):
    ops = _ops()
    g = golden(f"g1_warp_{tag}")
    d = g["depth"].to(DEV).requires_grad_(True)
    src, tgt = g["src"].to(DEV).permute(0, 3, 1, 2), g["tgt"].to(DEV).permute(0, 3, 1, 2)
    out = ops.warp_photometric(d, src, tgt, g["K"].to(DEV), g["invK"].to(DEV), g["T"].to(DEV), padding_mode=pad, want_pmap=True)
    _img_close(out["synth"], g[f"{pad}_synth"], what="synth")
    _mask_close(out["valid"], g["valid"], g["grid"], "valid")
    _img_close(out["pmap"], g[f"{pad}_pmap"], what="pmap")
    _close(out["photometric"], g[f"{pad}_loss"], what="loss")
    out["photometric"].backward()
    _gdepth_close(d.grad, g[f"{pad}_gdepth"], g["grid"], what="gdepth")


def test_fused_nomask_vs_golden(golden):
    ops = _ops()
    g = golden("g1_warp_b")
    d = g["depth"].to(DEV).requires_grad_(True)
    src, tgt = g["src"].to(DEV).permute(0, 3, 1, 2), g["tgt"].to(DEV).permute(0, 3, 1, 2)
    out = ops.warp_photometric(d, src, tgt, g["K"].to(DEV), g["invK"].to(DEV), g["T"].to(DEV), use_mask=False)
    _close(out["photometric"], g["nomask_loss"], what="loss")
    out["photometric"].backward()
    _gdepth_close(d.grad, g["nomask_gdepth"], g["grid"], what="gdepth")


def test_ssim_photometric_vs_golden(golden):
    ops = _ops()
    g = golden("g3_ssim")
    x = g["x"].to(DEV).requires_grad_(True)
    y = g["y"].to(DEV)
    _close(ops.ssim(x, y), g["ssim"], atol=1e-5, what="ssim")
    p = ops.photometric(x, y)
    _close(p, g["pmap"], atol=1e-5, what="pmap")
    gx, = torch.autograd.grad(p.mean(), x)
    _grad_close(gx, g["gx"], what="gx")
    # gradient wrt the second argument and through the per-channel SSIM map, vs the oracle's autograd
    xc, yc = g["x"].clone().requires_grad_(True), g["y"].clone().requires_grad_(True)
    w = torch.rand(g["ssim"].shape, generator=torch.Generator().manual_seed(0))
    (warp_loss.ssim(xc, yc) * w).sum().backward()
    xg, yg = g["x"].to(DEV).requires_grad_(True), g["y"].to(DEV).requires_grad_(True)
    (ops.ssim(xg, yg) * w.to(DEV)).sum().backward()
    _grad_close(xg.grad, xc.grad, what="ssim gx")
    _grad_close(yg.grad, yc.grad, what="ssim gy")


@pytest.mark.parametrize("H,W,B,pad,reg", [(37, 53, 1, "border", "l2"), (64, 96, 2, "zeros", "l1"), (8, 33, 1, "border", None),
                                           (3, 3, 1, "border", "l2")])
def test_fused_vs_oracle_ragged(H, W, B, pad, reg):
    """Sizes that are not multiples of the 32x8 tile, batch > 1, both paddings, with the regulariser."""
    ops = _ops()
    s = make_pair(H, W, seed=H * 1000 + W, B=B, rz=3.0, ry=2.0, t=(0.2, 0.05, 0.1))
    gen = torch.Generator().manual_seed(5)
    dsrc = s["depth"] + 0.1 * torch.rand(s["depth"].shape, generator=gen)
    it = s["depth"] + 0.05 * torch.rand(s["depth"].shape, generator=gen)
    is_ = dsrc + 0.05 * torch.rand(s["depth"].shape, generator=gen)
    # oracle
    dc, dsc = s["depth"].clone().requires_grad_(True), dsrc.clone().requires_grad_(True)
    synth, valid, grid = warp_loss.inverse_warp(dc, s["src"].permute(0, 3, 1, 2), s["K"], s["invK"], s["T"], pad)
    lp, pmap = warp_loss.masked_photometric_mean(synth, s["tgt"].permute(0, 3, 1, 2), valid)
    tot = lp * 1.7
    if reg:
        lr = warp_loss.depth_regularizer(it, dc, reg) + warp_loss.depth_regularizer(is_, dsc, reg)
        tot = tot + 0.3 * lr
    tot.backward()
    # HIP
    d, ds = s["depth"].to(DEV).requires_grad_(True), dsrc.to(DEV).requires_grad_(True)
    out = ops.warp_photometric(d, s["src"].to(DEV).permute(0, 3, 1, 2), s["tgt"].to(DEV).permute(0, 3, 1, 2),
                               s["K"].to(DEV), s["invK"].to(DEV), s["T"].to(DEV), padding_mode=pad,
                               depth_src=ds, init_tgt=it.to(DEV), init_src=is_.to(DEV), reg_kind=reg, want_pmap=True)
    t2 = out["photometric"] * 1.7
    if reg:
        _close(out["reg"], lr, what="reg")
        t2 = t2 + 0.3 * out["reg"]
    t2.backward()
    _img_close(out["synth"], synth, what="synth")
    _mask_close(out["valid"], valid, grid, "valid")
    _img_close(out["pmap"], pmap, what="pmap")
    _close(out["photometric"], lp, what="photometric")
    _gdepth_close(d.grad, dc.grad, grid, what="g depth_tgt")
    if reg:
        _grad_close(ds.grad, dsc.grad, what="g depth_src")


def test_full_size_checksums(golden):
    """480x640 (BASELINE size): the reference's checksums + sampled pixels, and run-to-run bit equality."""
    ops = _ops()
    g = golden("g9_full_checksums")
    s = make_pair(480, 640, seed=int(g["seed"]))
    d = s["depth"].to(DEV).requires_grad_(True)
    args = (s["src"].to(DEV).permute(0, 3, 1, 2), s["tgt"].to(DEV).permute(0, 3, 1, 2), s["K"].to(DEV), s["invK"].to(DEV), s["T"].to(DEV))
    out = ops.warp_photometric(d, *args, want_pmap=True)
    out["photometric"].backward()
    _close(out["photometric"], g["loss"], what="loss")
    assert abs(out["valid"].sum().item() - g["valid_sum"].item()) <= 4
    assert abs(out["synth"].double().sum().item() - g["synth_sum"].item()) < 1e-5 * g["synth_abs"].item()
    assert abs(d.grad.double().sum().item() - g["gdepth_sum"].item()) < 1e-3 * g["gdepth_abs"].item()
    assert abs(d.grad.double().abs().sum().item() - g["gdepth_abs"].item()) < 1e-3 * g["gdepth_abs"].item()
    pick = g["pick"].to(DEV)
    _img_close(out["synth"][0].reshape(3, -1)[:, pick], g["synth_pick"], what="synth pick")
    _img_close(out["pmap"].view(-1)[pick], g["pmap_pick"], what="pmap pick")
    _grad_close(d.grad.view(-1)[pick], g["gdepth_pick"], what="gdepth pick")
    d2 = s["depth"].to(DEV).requires_grad_(True)
    out2 = ops.warp_photometric(d2, *args)
    out2["photometric"].backward()
    assert torch.equal(out2["photometric"], out["photometric"]) and torch.equal(d2.grad, d.grad)


def test_errors():
    ops = _ops()
    from e2ehip import E2EError
    s = make_pair(16, 16)
    with pytest.raises(E2EError):          # CPU tensors are refused: no fallback
        ops.backproject(s["depth"], s["invK"])
    d = s["depth"].to(DEV)
    with pytest.raises(ValueError):
        ops.grid_sample(s["src"].to(DEV).permute(0, 3, 1, 2), torch.zeros(1, 16, 16, 2, device=DEV), padding_mode="reflection")
    with pytest.raises(ValueError):
        ops.warp_photometric(d, s["src"].to(DEV).permute(0, 3, 1, 2), s["tgt"].to(DEV).permute(0, 3, 1, 2), s["K"].to(DEV),
                             s["invK"].to(DEV), s["T"].to(DEV), reg_kind="huber")


@pytest.mark.parametrize("H,W,B,pad,reg,nhwc", [(48, 64, 1, "border", "l2", True), (37, 53, 2, "border", "l1", True),
                                                (64, 96, 2, "zeros", "l2", True), (20, 70, 1, "border", None, False),
                                                (3, 3, 1, "border", "l2", True), (480, 640, 1, "border", "l2", True)])
def test_lossgrad_single_launch_vs_oracle(H, W, B, pad, reg, nhwc):
    """e2e_warp_photo_lossgrad (loss + gradient in one pass) against the oracle's autograd."""
    from e2ehip.fused import LossGradPlan
    big = H >= 480
    s = make_pair(H, W, seed=H * 1000 + W, B=B) if big else make_pair(H, W, seed=H * 1000 + W, B=B, rz=3.0, ry=2.0, t=(0.2, 0.05, 0.1))
    gen = torch.Generator().manual_seed(5)
    dsrc = s["depth"] + 0.1 * torch.rand(s["depth"].shape, generator=gen)
    it = s["depth"] + 0.05 * torch.rand(s["depth"].shape, generator=gen)
    is_ = dsrc + 0.05 * torch.rand(s["depth"].shape, generator=gen)
    wp, wr = 1.3, 0.07
    dc, dsc = s["depth"].clone().requires_grad_(True), dsrc.clone().requires_grad_(True)
    synth, valid, grid = warp_loss.inverse_warp(dc, s["src"].permute(0, 3, 1, 2), s["K"], s["invK"], s["T"], pad)
    lp, _ = warp_loss.masked_photometric_mean(synth, s["tgt"].permute(0, 3, 1, 2), valid)
    tot = wp * lp
    if reg:
        lr = warp_loss.depth_regularizer(it, dc, reg) + warp_loss.depth_regularizer(is_, dsc, reg)
        tot = tot + wr * lr
    tot.backward()
    if nhwc:
        src, tgt = s["src"].to(DEV).permute(0, 3, 1, 2), s["tgt"].to(DEV).permute(0, 3, 1, 2)
    else:
        src, tgt = s["src"].permute(0, 3, 1, 2).contiguous().to(DEV), s["tgt"].permute(0, 3, 1, 2).contiguous().to(DEV)
    plan = LossGradPlan(B, H, W, torch.device(DEV), pad, True, reg, wp, wr).bind(
        s["depth"].to(DEV), dsrc.to(DEV), it.to(DEV), is_.to(DEV), src, tgt, s["K"].to(DEV), s["invK"].to(DEV), s["T"].to(DEV))
    loss, gdt, gds = plan.step()
    _close(loss[0], lp, what="photometric")
    _gdepth_close(gdt, dc.grad, grid, what="g depth_tgt")
    if reg:
        _close(loss[1], lr, what="reg")
        _grad_close(gds, dsc.grad, what="g depth_src")
    l2, g2, _ = plan.step()
    assert torch.equal(l2, loss) and torch.equal(g2, gdt)       # bitwise reproducible


def test_lossgrad_host_geometry_matches_device_geometry():
    """The 12 geometry numbers as kernel arguments (computed on the host in fp64) vs derived in the kernel from K, inv_K, T."""
    from e2ehip.fused import LossGradPlan
    H, W = 96, 128
    s = make_pair(H, W, seed=5)
    gen = torch.Generator().manual_seed(5)
    dsrc = s["depth"] + 0.1 * torch.rand(s["depth"].shape, generator=gen)
    args = (s["depth"].to(DEV), dsrc.to(DEV), (s["depth"] + 0.05).to(DEV), (dsrc + 0.05).to(DEV), s["src"].to(DEV).permute(0, 3, 1, 2),
            s["tgt"].to(DEV).permute(0, 3, 1, 2), s["K"].to(DEV), s["invK"].to(DEV), s["T"].to(DEV))
    a = LossGradPlan(1, H, W, torch.device(DEV), "border", True, "l2", 1.0, 1e-2).bind(*args)
    la, ga, gsa = (t.clone() for t in a.step())
    b = LossGradPlan(1, H, W, torch.device(DEV), "border", True, "l2", 1.0, 1e-2).bind(*args).set_host_geometry(s["K"][0], s["invK"][0], s["T"][0])
    lb, gb, gsb = b.step()
    torch.testing.assert_close(lb, la, rtol=1e-5, atol=1e-8)
    dc = s["depth"].clone().requires_grad_(True)
    synth, valid, grid = warp_loss.inverse_warp(dc, s["src"].permute(0, 3, 1, 2), s["K"], s["invK"], s["T"], "border")
    lp, _ = warp_loss.masked_photometric_mean(synth, s["tgt"].permute(0, 3, 1, 2), valid)
    (lp + 1e-2 * (warp_loss.depth_regularizer(s["depth"] + 0.05, dc, "l2"))).backward()
    _close(lb[0], lp, what="photometric")
    _gdepth_close(gb, dc.grad, grid, what="g depth_tgt (host geometry)")
    assert torch.equal(gsb, gsa)


@pytest.mark.parametrize("B,reg,hostgeo", [(1, "l2", True), (2, "l1", False), (1, None, False)])
def test_lossgrad_chain_equals_two_kernel_form(B, reg, hostgeo):
    """Chained launches (one kernel per step, fixed-point slot sums finalised by the next launch / the flush) give the
    same gradients bit for bit and the same losses to 1e-6, for several steps with changing depth, and are reproducible."""
    from e2ehip.fused import LossGradPlan
    H, W = 75, 101
    s = make_pair(H, W, seed=9, B=B)
    gen = torch.Generator().manual_seed(9)
    dsrc = s["depth"] + 0.1 * torch.rand(s["depth"].shape, generator=gen)
    depth = s["depth"].to(DEV).clone()
    args = (depth, dsrc.to(DEV), (s["depth"] + 0.05).to(DEV), (dsrc + 0.05).to(DEV), s["src"].to(DEV).permute(0, 3, 1, 2),
            s["tgt"].to(DEV).permute(0, 3, 1, 2), s["K"].to(DEV), s["invK"].to(DEV), s["T"].to(DEV))
    mk = lambda: LossGradPlan(B, H, W, torch.device(DEV), "border", True, reg, 1.0, 1e-2).bind(*args)
    ref, ch = mk(), mk()
    if hostgeo:
        for p in (ref, ch):
            p.set_host_geometry(s["K"][0], s["invK"][0], s["T"][0])
    steps = 5
    ref_losses, ref_grads = [], []
    for k in range(steps):
        depth.copy_(s["depth"].to(DEV) * (1.0 + 0.01 * k))
        l, g, _ = ref.step()
        ref_losses.append(l.clone()); ref_grads.append(g.clone())
    for rep in range(2):                                            # twice: the slot sets must come back clean
        losses = [torch.full((2,), -1.0, device=DEV) for _ in range(steps)]
        for k in range(steps):
            depth.copy_(s["depth"].to(DEV) * (1.0 + 0.01 * k))
            g, _ = ch.step_chain(k % 3, (k - 1) % 3 if k else -1, losses[k - 1] if k else None)
            assert torch.equal(g, ref_grads[k])
        ch.flush_chain((steps - 1) % 3, losses[steps - 1])
        for k in range(steps):
            torch.testing.assert_close(losses[k][0], ref_losses[k][0], rtol=1e-6, atol=1e-9)
            if reg:
                torch.testing.assert_close(losses[k][1], ref_losses[k][1], rtol=1e-6, atol=1e-9)
        if rep == 0:
            first = [l.clone() for l in losses]
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, losses))   # bitwise reproducible (integer sums)
    with pytest.raises(Exception):
        ch.step_chain(1, 1, losses[0])                                  # sets must differ


def test_lossgrad_chain_keeps_non_finite_losses_visible_and_tiny_images():
    """A NaN in the inputs must surface as a NaN loss in the chained form too (fixed-point sums cannot carry it: a flag
    slot does); and the chain works on images smaller than one tile."""
    from e2ehip.fused import LossGradPlan
    for (H, W, poison) in ((3, 5, False), (40, 33, True)):
        s = make_pair(H, W, seed=4)
        tgt = s["tgt"].clone()
        if poison:
            tgt[0, 7, 9, 1] = float("nan")
        args = (s["depth"].to(DEV), None, None, None, s["src"].to(DEV).permute(0, 3, 1, 2), tgt.to(DEV).permute(0, 3, 1, 2),
                s["K"].to(DEV), s["invK"].to(DEV), s["T"].to(DEV))
        ref = LossGradPlan(1, H, W, torch.device(DEV), "border", True, None).bind(*args)
        l_ref, g_ref, _ = ref.step()
        ch = LossGradPlan(1, H, W, torch.device(DEV), "border", True, None).bind(*args)
        out = torch.zeros(2, device=DEV)
        g, _ = ch.step_chain(0)
        ch.flush_chain(0, out)
        if poison:
            assert torch.isnan(l_ref[0]) and torch.isnan(out[0])
        else:
            torch.testing.assert_close(out[0], l_ref[0], rtol=1e-6, atol=1e-9)
            assert torch.equal(g, g_ref)
        out2 = torch.zeros(2, device=DEV)                       # the flag and the slots were cleared by the flush
        ch.bind(*((s["depth"].to(DEV),) + args[1:5] + (s["tgt"].to(DEV).permute(0, 3, 1, 2),) + args[6:]))
        ch.step_chain(1)
        ch.flush_chain(1, out2)
        assert torch.isfinite(out2[0])


@pytest.mark.parametrize("pad,align", [("border", False), ("zeros", True)])
def test_grid_sample_input_gradient_matches_torch_and_is_bitwise_reproducible(pad, align):
    """d/d(sampled image) of grid_sample (the adjoint of a data-dependent bilinear gather = a scatter; LOSS.geometric's interpolated
    source depth, online_adaption.py:436-439): values against torch's CPU F.grid_sample backward, and two runs bit for bit -- the
    contributions are accumulated as fixed-point integers (e2e_grid_sample_bwd_exact), so arrival order does not matter."""
    import torch.nn.functional as F
    from e2ehip import ops
    g = torch.Generator().manual_seed(4)
    B, C, H, W = 2, 3, 40, 56
    img = torch.randn(B, C, H, W, generator=g)
    grid = (torch.rand(B, H, W, 2, generator=g) * 2.4 - 1.2)                 # incl. out-of-range samples
    gout = torch.randn(B, C, H, W, generator=g)
    ir, gr = img.clone().requires_grad_(True), grid.clone().requires_grad_(True)
    F.grid_sample(ir, gr, mode="bilinear", padding_mode=pad, align_corners=align).backward(gout)
    runs = []
    for _ in range(2):
        idv, gdv = img.to(DEV).requires_grad_(True), grid.to(DEV).requires_grad_(True)
        ops.grid_sample(idv, gdv, padding_mode=pad, align_corners=align).backward(gout.to(DEV))
        runs.append((idv.grad.clone(), gdv.grad.clone()))
    torch.testing.assert_close(runs[0][0].cpu(), ir.grad, rtol=1e-5, atol=1e-5)
    # d/d(grid): a sample within rounding of an integer coordinate (a different bilinear tap set) or of the border clamp is a kink of the
    # function -- two evaluations of ix = (x + 1) * W / 2 - 0.5 that differ in the last bit (with / without a fused multiply-add) take
    # different sides there, and the gradient differs by its own size.  All but 0.1 % of the elements at 1e-4.
    bad = ~torch.isclose(runs[0][1].cpu(), gr.grad, rtol=1e-4, atol=1e-4)
    assert int(bad.sum()) <= 1e-3 * bad.numel(), int(bad.sum())
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


def test_grid_sample_input_gradient_refuses_non_finite_and_huge_contributions():
    """The fixed-point accumulation of e2e_grid_sample_bwd_exact holds sums below 32768 with contributions below 4096: an upstream
    gradient that is NaN / Inf or far out of range must not come back as a finite, plausible number (a wrapped 64-bit integer) -- the
    whole image gradient is NaN then, loudly."""
    from e2ehip import ops
    g = torch.Generator().manual_seed(9)
    B, C, H, W = 1, 2, 24, 32
    img = torch.randn(B, C, H, W, generator=g)
    grid = torch.rand(B, H, W, 2, generator=g) * 1.6 - 0.8
    for bad in (float("nan"), float("inf"), 1.0e7):
        gout = torch.randn(B, C, H, W, generator=g)
        gout[0, 1, 5, 7] = bad
        idv, gdv = img.to(DEV).requires_grad_(True), grid.to(DEV).requires_grad_(True)
        ops.grid_sample(idv, gdv, padding_mode="border", align_corners=False).backward(gout.to(DEV))
        assert torch.isnan(idv.grad).all(), bad
    gout = torch.randn(B, C, H, W, generator=g) * 100.0                        # large but legal: unaffected
    idv, gdv = img.to(DEV).requires_grad_(True), grid.to(DEV).requires_grad_(True)
    ops.grid_sample(idv, gdv, padding_mode="border", align_corners=False).backward(gout.to(DEV))
    assert torch.isfinite(idv.grad).all()
