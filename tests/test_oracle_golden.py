"""The oracle (oracle/*.py) against the golden vectors captured from the reference's own files
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import depthnet, knn, pointfusion, poses, refine, warp_loss

TOL = dict(rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g1_backproject_project(golden, tag):
    g = golden(f"g1_warp_{tag}")
    H, W = g["depth"].shape[2:]
    cam = warp_loss.backproject(g["depth"], g["invK"])
    torch.testing.assert_close(cam, g["cam"], **TOL)
    grid, valid = warp_loss.project(cam, g["K"], g["T"], H, W)
    torch.testing.assert_close(grid, g["grid"], **TOL)
    assert torch.equal(valid, g["valid"])
    _, z, _ = warp_loss.project(cam, g["K"], g["T"], H, W, geometric=True)
    torch.testing.assert_close(z, g["geo_depth"], **TOL)


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("pad", ["border", "zeros"])
def test_g1_g4_warp_loss_and_grads(golden, tag, pad):
    g = golden(f"g1_warp_{tag}")
    d = g["depth"].clone().requires_grad_(True)
    src, tgt = g["src"].permute(0, 3, 1, 2), g["tgt"].permute(0, 3, 1, 2)
    synth, valid, _ = warp_loss.inverse_warp(d, src, g["K"], g["invK"], g["T"], pad)
    loss, pmap = warp_loss.masked_photometric_mean(synth, tgt, valid)
    gd, = torch.autograd.grad(loss, d, retain_graph=True)
    gs, = torch.autograd.grad(loss, synth)
    torch.testing.assert_close(synth, g[f"{pad}_synth"], **TOL)
    torch.testing.assert_close(pmap, g[f"{pad}_pmap"], **TOL)
    torch.testing.assert_close(loss, g[f"{pad}_loss"], **TOL)
    torch.testing.assert_close(gs, g[f"{pad}_gsynth"], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(gd, g[f"{pad}_gdepth"], rtol=1e-4, atol=1e-9)


def test_g1_nomask(golden):
    g = golden("g1_warp_b")
    d = g["depth"].clone().requires_grad_(True)
    synth, valid, _ = warp_loss.inverse_warp(d, g["src"].permute(0, 3, 1, 2), g["K"], g["invK"], g["T"], "border")
    loss, _ = warp_loss.masked_photometric_mean(synth, g["tgt"].permute(0, 3, 1, 2), valid, use_mask=False)
    gd, = torch.autograd.grad(loss, d)
    torch.testing.assert_close(loss, g["nomask_loss"], **TOL)
    torch.testing.assert_close(gd, g["nomask_gdepth"], rtol=1e-4, atol=1e-9)


def test_g3_ssim(golden):
    g = golden("g3_ssim")
    x = g["x"].clone().requires_grad_(True)
    torch.testing.assert_close(warp_loss.ssim(x, g["y"]), g["ssim"], **TOL)
    p = warp_loss.photometric(x, g["y"])
    torch.testing.assert_close(p, g["pmap"], **TOL)
    gx, = torch.autograd.grad(p.mean(), x)
    torch.testing.assert_close(gx, g["gx"], rtol=1e-5, atol=1e-9)


def test_g5_aux(golden):
    g = golden("g5_aux")
    disp = g["disp"].clone().requires_grad_(True)
    sm = warp_loss.smoothness(disp, g["img"])
    torch.testing.assert_close(sm, g["smooth"], **TOL)
    torch.testing.assert_close(torch.autograd.grad(sm, disp)[0], g["gsmooth"], **TOL)
    d1 = g["d1"].clone().requires_grad_(True)
    torch.testing.assert_close(warp_loss.depth_regularizer(g["d0"], d1, "l1"), g["reg_l1"], **TOL)
    r2 = warp_loss.depth_regularizer(g["d0"], d1, "l2")
    torch.testing.assert_close(r2, g["reg_l2"], **TOL)
    torch.testing.assert_close(torch.autograd.grad(r2, d1)[0], g["greg_l2"], **TOL)
    with pytest.raises(ValueError):
        warp_loss.depth_regularizer(g["d0"], d1, "huber")
    torch.testing.assert_close(torch.stack(warp_loss.depth_metrics("ICL", g["gt"], g["pred"])), g["metrics_icl"], **TOL)
    torch.testing.assert_close(torch.stack(warp_loss.depth_metrics("TUM", g["gt_holes"], g["pred"])), g["metrics_tum"], **TOL)
    torch.testing.assert_close(warp_loss.geometric_consistency(g["wd"], g["idp"], g["vm"]), g["geo"], **TOL)
    torch.testing.assert_close(warp_loss.depth_gt(g["d1"], g["d0"] * g["smask"], g["smask"]), g["gt_loss"], **TOL)


def test_g6_pose(golden):
    g = golden("g6_pose")
    torch.testing.assert_close(poses.poses_to_transforms(g["poses"]), g["transforms"], **TOL)
    torch.testing.assert_close(poses.inverse_T(g["poses"][0]), g["inverse"], **TOL)
    torch.manual_seed(5)
    md, mk = poses.sparse_sampling(0.3, g["sp_depth"])
    assert torch.equal(mk, g["sp_mask"]) and torch.equal(md, g["sp_masked"])
    torch.testing.assert_close(poses.disp_to_depth(torch.linspace(0, 1, 9), 0.1, 80.0), g["d2d"], **TOL)


def test_g7_network(golden):
    g = golden("g7_net")
    sd = depthnet.random_state_dict(0)
    assert list(sd.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    s = sum(float(v.double().abs().sum()) for v in sd.values() if v.dtype.is_floating_point)
    assert abs(s - float(g["sd_abs_sum"])) < 1e-6 * s, "seeded state dict drifted (torch RNG changed?)"
    tk = depthnet.trainable_keys(sd)
    for k in tk:
        sd[k].requires_grad_(True)
    feats = depthnet.encoder_forward(sd, g["img"])
    for i, f in enumerate(feats):
        torch.testing.assert_close(f, g[f"f{i}"], rtol=1e-4, atol=1e-5)
    disp = depthnet.decoder_forward(sd, feats)
    torch.testing.assert_close(disp, g["disp"], rtol=1e-4, atol=1e-5)
    (disp * g["wgt"]).sum().backward()
    names = list(g["grad_names"])
    for n, ref in zip(names, g["grad_norms"]):
        if ref < 0:                       # grad None in the reference: frozen "bn" names never appear here
            assert n not in tk or sd[n].grad is None or n.startswith("encoder.encoder.fc") or ".11." in n or ".12." in n or ".13." in n
            continue
        torch.testing.assert_close(sd[n].grad.norm(), ref, rtol=2e-4, atol=1e-7)
    assert sorted(tk) == sorted(n for n, r in zip(names, g["grad_norms"]) if r >= 0)
    for k in [k for k in g if k.startswith("gs_")]:
        name = [n for n in names if "gs_" + n.replace(".", "_") == k][0]
        torch.testing.assert_close(sd[name].grad.flatten()[:16], g[k], rtol=2e-3, atol=1e-6)


def test_g8_three_refinement_steps(golden):
    g = golden("g8_refine")
    cfg = refine.Config()
    r = refine.Refiner(depthnet.random_state_dict(0), cfg)
    recs = r.refine_pair(g["colors"], g["gt_depths"], g["poses"], g["K"])
    np.testing.assert_allclose([x["loss"] for x in recs], g["losses"].numpy(), rtol=1e-4)
    np.testing.assert_allclose([x["ratio"] for x in recs], g["ratios"].numpy(), rtol=1e-4)
    torch.testing.assert_close(recs[0]["depth1"], g["first_depth1"], rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        for i in range(2):
            torch.testing.assert_close(depthnet.disp_forward(r.sd, g["colors"][:, i]), g[f"final_disp{i}"], rtol=1e-4, atol=1e-5)
    # the map step ran: first keyframe -> prev frame appended whole, live frame fused
    H, W = g["colors"].shape[2:4]
    assert r.map["points"].shape[0] >= H * W and len(r.tables) == 2


def test_knn_c_matches_torch():
    torch.manual_seed(0)
    a, b = torch.rand(3000, 3), torch.rand(4111, 3)
    b[17] = b[5]                              # duplicate reference point: first index must win
    a[0] = b[17]
    d, i = knn.knn1(a, b)
    d2, i2 = knn.knn1_torch(a, b)
    assert torch.equal(d, d2) and torch.equal(i, i2) and i[0] == 5 and d[0] == 0


def test_pointfusion_oracle_properties():
    """Unpinned part: structural properties the recalled semantics imply."""
    torch.manual_seed(1)
    H, W = 24, 32
    K = torch.eye(4); K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 30.0, -30.0, 15.5, 11.5
    depth = 2.0 + 0.1 * torch.rand(H, W); depth[3, 4] = 0.0
    col = torch.rand(H, W, 3)
    pose = torch.eye(4)
    st, t = pointfusion.pointfusion_step(pointfusion.empty_state(), col, depth, K, pose)
    assert st["points"].shape[0] == H * W - 1 and t["active"].shape[0] == 0
    # same frame again: every surviving point projects onto its own pixel and fuses with itself
    st2, t2 = pointfusion.pointfusion_step(st, col, depth, K, pose)
    u = t2["unique"]
    assert u.shape[0] > 0.8 * (H * W)
    pix = u[:, 1] * W + u[:, 2]
    assert torch.equal(pix, torch.sort(pix)[0]) and pix.unique().numel() == pix.numel()
    assert st2["points"].shape[0] == st["points"].shape[0] + (H * W - 1 - u.shape[0])
    assert torch.all(st2["ccounts"][u[:, 0]] > st["ccounts"][u[:, 0]])
