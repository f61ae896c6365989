"""Frame-to-model ICP odometry on the GPU against the CPU oracle and the known camera motion."""
import numpy as np
import pytest
import torch

from oracle import icp as oicp
from oracle import pointfusion as opf

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(L=3, H=96, W=128, step=0.04):
    from e2ehip.synthetic import make_sequence
    colors, depths, K, poses = make_sequence(L, H, W, seed=3, step=step, noise=0.0, scene="corner")
    return colors[0], depths[0, ..., 0], K[0, 0], poses[0]


@pytest.mark.parametrize("mode", ["icp", "gradicp"])
def test_frame_to_model_matches_oracle_and_ground_truth(mode):
    from e2ehip import icp
    from e2ehip.fusionmap import FusionMap
    colors, depths, K, poses = _scene()
    H, W = depths.shape[1:]
    st, _ = opf.pointfusion_step(opf.empty_state(), colors[0], depths[0], K, poses[0])
    P_ref, tr_ref = oicp.frame_to_model(st["points"], st["normals"], depths[1], K, poses[0], mode=mode)
    fm = FusionMap(3 * H * W, H, W, DEV)
    fm.load_state(st["points"].to(DEV), st["normals"].to(DEV), st["colors"].to(DEV), st["ccounts"].to(DEV))
    P, tr = icp.frame_to_model(fm, depths[1].to(DEV), K.to(DEV), poses[0].to(DEV), mode=mode)
    assert len(tr) == len(tr_ref) and tr[0][0] == tr_ref[0][0]                  # same inlier count on the first iteration
    np.testing.assert_allclose(P.cpu().numpy(), P_ref, atol=2e-5, rtol=0)
    gt = poses[1].numpy()
    assert np.linalg.norm(P.cpu().numpy()[:3, 3] - gt[:3, 3]) < 3e-3            # 4 cm of motion recovered to a few mm
    assert np.linalg.norm(poses[0].numpy()[:3, 3] - gt[:3, 3]) > 3e-2


@pytest.mark.parametrize("mode", ["icp", "gradicp"])
def test_resident_odometry_equals_host_loop_and_oracle(mode):
    """e2ehip.icp.ResidentOdometry (round 4): selection of source / target clouds, the index over the targets and the 20 iterations with
    the 6x6 solve, the se(3) exponential and GradICP's damping update ON THE DEVICE (e2e_icp_update), no host round trip -- against the
    host-driven loop of the same kernels (frame_to_model: numpy solve per iteration) and against the CPU oracle: same inlier counts in
    every iteration, same pose.  Run twice (the second time as a captured graph replay would: same buffers) -> bit-identical."""
    from e2ehip import icp
    from e2ehip.fusionmap import FusionMap
    colors, depths, K, poses = _scene()
    H, W = depths.shape[1:]
    st, _ = opf.pointfusion_step(opf.empty_state(), colors[0], depths[0], K, poses[0])
    P_ref, tr_ref = oicp.frame_to_model(st["points"], st["normals"], depths[1], K, poses[0], mode=mode)
    fm = FusionMap(3 * H * W, H, W, DEV)
    fm.load_state(st["points"].to(DEV), st["normals"].to(DEV), st["colors"].to(DEV), st["ccounts"].to(DEV))
    d1, Kd, p0 = depths[1].to(DEV).contiguous(), K.to(DEV).contiguous(), poses[0].to(DEV).contiguous()
    P_host, tr_host = icp.frame_to_model(fm, d1, Kd, p0, mode=mode)
    odo = icp.ResidentOdometry(fm, dsratio=4, numiters=20, mode=mode)
    P1 = odo.run(d1, Kd, p0).clone()
    its, tr, ntgt, nact = odo.check()
    assert its == len(tr_host) == len(tr_ref) and ntgt == (nact + 3) // 4
    assert [c for c, _ in tr] == [c for c, _ in tr_host] == [c for c, _ in tr_ref]
    np.testing.assert_allclose([e for _, e in tr], [e for _, e in tr_host], rtol=1e-6)
    np.testing.assert_allclose(P1.cpu().numpy(), P_host.cpu().numpy(), atol=2e-6, rtol=0)
    np.testing.assert_allclose(P1.cpu().numpy(), P_ref, atol=2e-5, rtol=0)
    P2 = odo.run(d1, Kd, p0)
    assert torch.equal(P1, P2)
    # a hole under a selected pixel is reported, not silently mis-registered
    d_bad = d1.clone()
    d_bad[0, 0] = 0.0
    odo.run(d_bad, Kd, p0)
    with pytest.raises(RuntimeError, match="no depth"):
        odo.check()


def test_driver_gradicp_runs_resident_and_reports_ate():
    """configs/config.yaml:30 (odom: gradicp) through the driver's launch plan: the map step with odometry is one captured graph
    (RefineStepPlan.update_map_odom); the estimated poses stay on the device until the trajectory error is asked for."""
    from e2ehip.synthetic import make_sequence
    from online_adaption import SLAM, default_config
    from oracle import depthnet
    cfg = default_config(96, 128, 5)
    cfg.MODEL.odom = "gradicp"
    cfg.DEMO.frame_threshold = 0.0
    cfg.DEBUG.print_metrics = False
    seq = make_sequence(5, 96, 128, seed=3, step=0.03, noise=0.0, scene="corner")
    sd = depthnet.random_state_dict(0)
    sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * 40.0
    drv = SLAM(cfg, sequence=seq, state_dict=sd)
    drv.main()
    assert len(drv.estimated_poses) == 4 and all(p.is_cuda for p, _ in drv.estimated_poses)
    assert any(k[0] == "map_odom" for k in drv.step_plan._graphs if isinstance(k, tuple))
    ate = drv.absolute_trajectory_error()
    assert np.isfinite(ate) and ate < 0.25                   # random-weight depths: plumbing, not accuracy
    its, tr, ntgt, nact = drv._odo.check()
    assert its >= 1 and ntgt > 100
    drv.close()


def test_normal_equations_kernel_vs_numpy():
    from e2ehip import _lib as L
    from e2ehip import ops
    g = torch.Generator().manual_seed(0)
    src, tgt = torch.rand(5000, 3, generator=g), torch.rand(3000, 3, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(3000, 3, generator=g), dim=1)
    sd, td, nd = src.to(DEV), tgt.to(DEV), nrm.to(DEV)          # keep the device tensors alive across the raw-pointer call
    d, idx = ops.knn1(sd, td)
    out = torch.empty(29, device=DEV, dtype=torch.float64)
    ws = torch.empty(L.load().e2e_icp_workspace_bytes(), device=DEV, dtype=torch.uint8)
    for th in (None, 0.05):
        L.call("e2e_icp_normal_equations", L.ptr(sd), L.ptr(td), L.ptr(nd), L.ptr(idx), L.ptr(d),
               -1.0 if th is None else th, 5000, L.ptr(out), L.ptr(ws), L.stream())
        AtA, Atb, cnt, err = oicp.normal_equations(src, tgt, nrm, idx.cpu(), d.cpu(), th)
        v = out.cpu().numpy()
        k = 0
        for r in range(6):
            for c in range(r, 6):
                assert abs(v[k] - AtA[r, c]) < 1e-9 * max(1.0, abs(AtA[r, c]))
                k += 1
        np.testing.assert_allclose(v[21:27], Atb, rtol=1e-9, atol=1e-12)
        assert int(v[27]) == cnt and abs(v[28] - err) < 1e-9 * max(err, 1.0)


def test_pointfusion_step_with_icp_odometry_and_driver_ate():
    """gradslam surface: PointFusion(odom='icp').step(map, live, prev) localises, then fuses with the estimated pose;
    the driver reports the trajectory error of a short sequence."""
    from gradslam import Pointclouds, RGBDImages
    from gradslam.slam import PointFusion
    colors, depths, K, poses = _scene(L=4, H=96, W=128, step=0.03)

    def rgbd(i, pose):
        return RGBDImages(colors[i].to(DEV)[None, None], depths[i].to(DEV)[None, None, ..., None], K.to(DEV)[None, None],
                          None if pose is None else pose.to(DEV)[None, None])

    slam = PointFusion(odom="icp", numiters=20, device=DEV)
    cloud, _ = slam.step(Pointclouds(device=DEV), rgbd(0, poses[0]), None)
    prev = rgbd(0, poses[0])
    errs = []
    for i in (1, 2, 3):
        live = rgbd(i, None)
        cloud, est = slam.step(cloud, live, prev)
        errs.append(float((est[0, 0, :3, 3].cpu() - poses[i][:3, 3]).norm()))
        prev = live                                   # carries the ESTIMATED pose forward
    assert max(errs) < 6e-3, errs                     # drift over three 3 cm steps stays at the mm level
    assert cloud.points_list[0].shape[0] > colors.shape[1] * colors.shape[2]

    from online_adaption import SLAM, default_config
    from oracle import depthnet
    cfg = default_config(96, 128, 4)
    cfg.MODEL.odom = "icp"
    cfg.DEMO.frame_threshold = 0.0
    cfg.DEBUG.print_metrics = False
    cfg.OPTIMIZATION.refinement_steps = 1
    from e2ehip.synthetic import make_sequence
    seq = make_sequence(4, 96, 128, seed=3, step=0.03, noise=0.0, scene="corner")
    drv = SLAM(cfg, sequence=seq, state_dict=depthnet.random_state_dict(0))
    drv.main()
    assert len(drv.estimated_poses) == 3 and drv.absolute_trajectory_error() < 0.25   # random-weight depths: only checks plumbing
