"""Host-side pieces that need no GPU: PLY export round trip, and the N3/N4 entry points refusing CPU tensors (there is no
CPU fallback anywhere in the product path)."""
import numpy as np
import pytest
import torch


def test_ply_roundtrip_cpu(tmp_path):
    from utils.export import load_ply, save_ply
    g = torch.Generator().manual_seed(2)
    pts = torch.randn(257, 3, generator=g)
    col = torch.rand(257, 3, generator=g)                               # 0..1 colours are scaled to 0..255
    rec = load_ply(save_ply(str(tmp_path / "m.ply"), pts, col))
    assert rec.dtype.names == ("x", "y", "z", "red", "green", "blue") and rec.shape[0] == 257
    np.testing.assert_array_equal(np.stack([rec["x"], rec["y"], rec["z"]], 1), pts.numpy())
    np.testing.assert_array_equal(rec["green"], np.clip(np.rint(col.numpy()[:, 1] * 255.0), 0, 255).astype(np.uint8))
    with pytest.raises(ValueError):
        save_ply(str(tmp_path / "bad.ply"), torch.zeros(4, 2))


def test_new_entry_points_have_no_cpu_fallback():
    from e2ehip import ops
    from e2ehip._lib import E2EError
    from e2ehip.tensor_refine import learn_depth_scale, refine_depth_tensor
    d = torch.rand(1, 1, 8, 8) + 0.5
    img = torch.rand(1, 3, 8, 8)
    K = torch.eye(4).reshape(1, 4, 4)
    for call in (lambda: ops.smoothness(d, img), lambda: ops.geometric_consistency(d, d, torch.ones_like(d)),
                 lambda: ops.masked_l1(d, d[0, 0], torch.ones(8, 8)), lambda: ops.min_reprojection(torch.rand(1, 2, 8, 8)),
                 lambda: ops.process_disparity(torch.rand(2, 1, 8, 8)), lambda: ops.KnnIndex(torch.rand(10, 3), 4),
                 lambda: refine_depth_tensor(d, img, img, K, K), lambda: learn_depth_scale(d, img, img, K, K, steps=1)):
        with pytest.raises(E2EError):
            call()


def test_product_training_utils_vs_golden_g6(golden):
    """H14: the PRODUCT's utils.training_utils (not the oracle twin, which tests/test_oracle_golden.py::test_g6_pose pins) against
    fixture g6, captured from the reference's own utils/training_utils.py:106-118,130-140,176-216."""
    from utils import training_utils as tu
    g = golden("g6_pose")
    tol = dict(rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(tu.torch_poses_to_transforms(g["poses"]), g["transforms"], **tol)
    torch.testing.assert_close(tu.inverse_T_matrix(g["poses"][0]), g["inverse"], **tol)
    torch.manual_seed(5)
    md, mk = tu.sparse_sampling("random", 0.3, g["sp_depth"])
    assert torch.equal(mk, g["sp_mask"]) and torch.equal(md, g["sp_masked"])
    with pytest.raises(ValueError):
        tu.sparse_sampling("grid", 0.3, g["sp_depth"])
    torch.testing.assert_close(tu.convert_disp_to_depth(torch.linspace(0, 1, 9), 0.1, 80.0), g["d2d"], **tol)


def _closed_form_T12(P1, P2):
    """The reference's own known answer, pose_checker.py:57-82: T_12 = [R1^T R2 | R1^T (t2 - t1)] "should match Transform 2"
    (the loader's frame-to-frame transform)."""
    R1, t1, R2, t2 = P1[:3, :3], P1[:3, 3], P2[:3, :3], P2[:3, 3]
    T = np.eye(4)
    T[:3, :3] = R1.T @ R2
    T[:3, 3] = R1.T @ (t2 - t1)
    return T


def test_transforms_match_the_pose_checker_closed_form(tmp_path):
    """The one known answer the reference holds for the gradslam half of the path (pose_checker.py:57-82): the loader's `transforms`
    (and training_utils.torch_poses_to_transforms on its `poses`, online_adaption.py:270) equal R1^T R2 | R1^T (t2 - t1) for rigid
    poses -- checked for the product's ICL loader, the product's pose helper and the oracle's twin of it."""
    import os
    from PIL import Image
    from gradslam.datasets import ICL
    from oracle import poses as oposes
    from utils.training_utils import torch_poses_to_transforms
    rng = np.random.default_rng(3)
    n, H, W = 9, 12, 16

    def rigid(i):
        a, b, c = 0.07 * i, -0.04 * i, 0.02 * i * i
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(c), -np.sin(c)], [0, np.sin(c), np.cos(c)]])
        T = np.eye(4)
        T[:3, :3] = Rz @ Ry @ Rx
        T[:3, 3] = (0.3 * i, 0.1 * np.sin(i), -0.2 * i + 0.05 * rng.random())
        return T
    P = [rigid(i) for i in range(n)]
    traj = tmp_path / "ICL" / "living_room_traj1_frei_png"
    os.makedirs(traj / "rgb")
    os.makedirs(traj / "depth")
    for i in range(n):
        Image.fromarray(np.full((H, W, 3), 7 * i, dtype=np.uint8)).save(traj / "rgb" / f"{i}.png")
        Image.fromarray(np.full((H, W), 5000, dtype=np.uint16)).save(traj / "depth" / f"{i}.png")
    with open(traj / "associations.txt", "w") as f:
        for i in range(n):
            f.write(f"{i} depth/{i}.png {i} rgb/{i}.png\n")
    with open(traj / "livingRoom1n.gt.sim", "w") as f:
        for i in range(n):
            for r in P[i][:3]:
                f.write(" ".join(f"{v:.10f}" for v in r) + "\n")
            f.write("\n")
    # the reference's call: seqlen 3, a dilation, a start (pose_checker.py:43)
    ds = ICL(str(tmp_path / "ICL"), trajectories=("living_room_traj1_frei_png",), seqlen=3, height=H, width=W, dilation=2, start=1)
    _, _, _, poses, transforms, *_ = ds[0]
    want = torch.from_numpy(_closed_form_T12(poses[1].double().numpy(), poses[2].double().numpy())).float()
    torch.testing.assert_close(transforms[2], want, rtol=1e-5, atol=1e-5)                                    # "should match Transform 2"
    torch.testing.assert_close(transforms[1], torch.from_numpy(_closed_form_T12(poses[0].double().numpy(), poses[1].double().numpy())).float(),
                               rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(transforms[0], torch.eye(4), rtol=0, atol=1e-6)
    for fn in (torch_poses_to_transforms, oposes.poses_to_transforms):                                       # online_adaption.py:270
        T = fn(poses[None])[0]
        torch.testing.assert_close(T[2], want, rtol=1e-5, atol=1e-5)
    # and directly on the absolute poses of the file (frames 1, 4, 7)
    torch.testing.assert_close(transforms[2].double(), torch.from_numpy(_closed_form_T12(P[4], P[7])), rtol=1e-5, atol=1e-5)
