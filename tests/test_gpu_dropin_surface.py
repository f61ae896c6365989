"""The reference's import surface (same module / class / function names) running on the HIP kernels."""
import pytest
import torch

from oracle import knn as oknn
from oracle import pointfusion as opf
from oracle import warp_loss
from synth import make_pair

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_reference_style_refinement_iteration():
    """The body of online_adaption.py:412-455 + :544-564 written exactly as the reference writes it, against the oracle."""
    from depth_estimation.view_synthesis import BackprojectDepth, Project3D
    from e2ehip import ops
    from loss.losses import SSIM, depth_reguralizer, photometric_loss
    H, W = 48, 64
    s = make_pair(H, W, seed=11)
    bp, pr, ssim = BackprojectDepth(1, H, W).to(DEV), Project3D(1, H, W).to(DEV), SSIM().to(DEV)
    colors = torch.stack([s["src"], s["tgt"]], 1).to(DEV)           # (1,2,H,W,3)
    src, tgt = colors[:, 0].permute(0, 3, 1, 2), colors[:, 1].permute(0, 3, 1, 2)
    depth = s["depth"].to(DEV).requires_grad_(True)
    K, T = s["K"].to(DEV), s["T"].to(DEV)
    cam = bp(depth, torch.pinverse(K))
    grid, valid = pr(points=cam, K=K, T=T, geometric=False)
    synth = ops.grid_sample(src, grid, padding_mode="border", align_corners=False)
    pm = photometric_loss(ssim=ssim, prediction=synth * valid, target=tgt * valid)
    init = (s["depth"] + 0.03).to(DEV)
    loss = pm.mean(1, keepdim=True).mean() + 1e-2 * depth_reguralizer(init, depth, "l2")
    loss.backward()
    dc = s["depth"].clone().requires_grad_(True)
    sy, va, _ = warp_loss.inverse_warp(dc, s["src"].permute(0, 3, 1, 2), s["K"], s["invK"], s["T"], "border")
    lo, _ = warp_loss.masked_photometric_mean(sy, s["tgt"].permute(0, 3, 1, 2), va)
    lo = lo + 1e-2 * warp_loss.depth_regularizer(s["depth"] + 0.03, dc, "l2")
    lo.backward()
    torch.testing.assert_close(loss.detach().cpu(), lo.detach(), rtol=1e-4, atol=1e-7)
    assert ((depth.grad.cpu() - dc.grad).abs().max() / dc.grad.abs().max()) < 2e-3
    with pytest.raises(ValueError):
        depth_reguralizer(init, depth, "l3")


def test_gradslam_surface_pointfusion_and_knn_loss():
    """online_adaption.py:347-363 (map steps), :461-469 (local cloud), :638-645 (3-D loss) through the gradslam /
    chamferdist names, against the oracle."""
    from gradslam import Pointclouds, RGBDImages
    from gradslam.geometry.geometryutils import transform_pointcloud
    from gradslam.slam import PointFusion
    from gradslam.slam.fusionutils import find_active_map_points
    from loss.losses import knn_points_loss
    from test_gpu_pointfusion_knn import _K, _pose, _scene
    H, W = 36, 48
    K = _K(H, W)
    d0, c0 = _scene(H, W, 5)
    d1, c1 = _scene(H, W, 5)
    p0, p1 = _pose(), _pose(0.4, 0.8, 0.2, (0.02, 0.0, -0.01))
    slam = PointFusion(odom="gt", dist_th=0.05, angle_th=20, sigma=0.6, device=DEV)

    def rgbd(c, d, p):
        return RGBDImages(c.to(DEV)[None, None], d.to(DEV)[None, None, ..., None], K.to(DEV)[None, None],
                          None if p is None else p.to(DEV)[None, None])

    cloud = Pointclouds(device=DEV)
    cloud, _ = slam.step(cloud, rgbd(c0, d0, p0), None)
    st, _ = opf.pointfusion_step(opf.empty_state(), c0, d0, K, p0)
    assert torch.equal(cloud.points_list[0].cpu(), st["points"])
    act = find_active_map_points(cloud, rgbd(c1, d1, p1))
    st2, tab = opf.pointfusion_step(st, c1, d1, K, p1)
    assert torch.equal(act[:, 1:].cpu(), tab["active"]) and int(act[:, 0].abs().sum()) == 0
    cloud, poses = slam.step(cloud, rgbd(c1, d1, p1), None)
    assert cloud.points_list[0].shape[0] == st2["points"].shape[0] and tuple(poses.shape) == (1, 1, 4, 4)
    torch.testing.assert_close(cloud.points_list[0].cpu(), st2["points"], rtol=1e-6, atol=1e-7)
    # 3-D loss of a (differentiable) local cloud against the map
    dpred = (d1 * 1.02).to(DEV).requires_grad_(True)
    local = RGBDImages(c1.to(DEV)[None, None], dpred[None, None, ..., None], K.to(DEV)[None, None], p1.to(DEV)[None, None])
    target_pc, _ = slam.step(Pointclouds(device=DEV), local, None)
    T = _pose(0.1, 0.1, 0.0, (0.001, 0.0, 0.0))
    moved = transform_pointcloud(target_pc.points_list[0], T.to(DEV))
    loss, idx = knn_points_loss(cloud.points_list[0].unsqueeze(0).detach(), moved.unsqueeze(0))
    loss.backward()
    dc = (d1 * 1.02).clone().requires_grad_(True)
    maps = opf.vertex_normal_maps(dc, K, p1)
    mv = opf.transform_pointcloud(maps["Vg"][maps["valid"]], T)
    lref, iref = oknn.knn_points_loss(st2["points"].unsqueeze(0), mv.unsqueeze(0))
    lref.backward()
    torch.testing.assert_close(loss.detach().cpu(), lref.detach(), rtol=1e-4, atol=1e-9)
    assert (idx.cpu() != iref).float().mean() < 1e-3          # map positions differ by exp() ulps -> a rare flipped neighbour
    assert ((dpred.grad.cpu() - dc.grad).abs().max() / dc.grad.abs().max()) < 1e-3
    with pytest.raises(ValueError):
        knn_points_loss(torch.zeros(2, 4, 3, device=DEV), torch.zeros(1, 4, 3, device=DEV))
    # the reference's default odometry (configs/config.yaml:30): frame-to-model GradICP from the previous frame's pose
    c2, est = PointFusion(odom="gradicp", device=DEV).step(cloud, rgbd(c1, d1, None), rgbd(c0, d0, p0))
    assert tuple(est.shape) == (1, 1, 4, 4) and torch.isfinite(est).all() and c2.points_list[0].shape[0] >= cloud.points_list[0].shape[0]


def test_image_recover_slam_and_icpslam():
    from gradslam import RGBDImages
    from gradslam.slam import ICPSLAM
    from slam.custom_slam import image_recover_slam
    from test_gpu_pointfusion_knn import _K, _pose, _scene
    H, W = 20, 28
    K = _K(H, W)
    ds, cs, ps = zip(*[(*_scene(H, W, 7 + i), _pose(0, 0.5 * i, 0, (0.01 * i, 0, 0))) for i in range(3)])
    depth = torch.stack([d for d, _ in zip(ds, cs)], 0)
    frames = RGBDImages(torch.stack(cs, 0).to(DEV)[None], depth.to(DEV)[None, ..., None], K.to(DEV)[None, None], torch.stack(ps, 0).to(DEV)[None])
    cloud = image_recover_slam(frames, ICPSLAM(odom="gt", device=DEV), DEV)
    assert cloud.points_list[0].shape[0] == int((depth != 0).sum())
    pc, poses = ICPSLAM(odom="gt", device=DEV)(frames)
    assert tuple(poses.shape) == (1, 3, 4, 4) and pc.points_list[0].shape[0] == cloud.points_list[0].shape[0]


def test_image_recover_slam_values_vs_oracle():
    """H15 (slam/custom_slam.py:6-35) with PointFusion(odom="gt"): the fused map after three frames -- point count, positions,
    normals, colours and confidences -- against the oracle's PointFusion chain, and the gradient of a loss on the map reaches
    the LAST frame's depth only (every earlier frame is detached by image_recover_slam)."""
    from gradslam import RGBDImages
    from gradslam.slam import PointFusion
    from oracle import pointfusion as opf
    from slam.custom_slam import image_recover_slam
    from test_gpu_pointfusion_knn import _K, _pose, _scene
    H, W = 40, 56
    K = _K(H, W)
    ds, cs, ps = [], [], []
    for i in range(3):
        d, c = _scene(H, W, 30)
        ds.append(d); cs.append(c); ps.append(_pose(0.3 * i, 0.6 * i, 0.0, (0.012 * i, 0.0, -0.01 * i)))
    st = opf.empty_state()
    for d, c, p in zip(ds, cs, ps):
        st, _ = opf.pointfusion_step(st, c, d, K, p)
    depth = torch.stack(ds, 0).to(DEV)[None, ..., None].requires_grad_(True)
    frames = RGBDImages(torch.stack(cs, 0).to(DEV)[None], depth, K.to(DEV)[None, None], torch.stack(ps, 0).to(DEV)[None])
    cloud = image_recover_slam(frames, PointFusion(odom="gt", device=DEV), DEV)
    P = cloud.points_list[0]
    assert P.shape[0] == st["points"].shape[0]
    torch.testing.assert_close(P.detach().cpu(), st["points"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(cloud.normals_list[0].detach().cpu(), st["normals"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(cloud.colors_list[0].detach().cpu(), st["colors"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(cloud.features_list[0].detach().cpu().reshape(-1), st["ccounts"], rtol=1e-5, atol=1e-6)
