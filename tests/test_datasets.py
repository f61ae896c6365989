"""ICL / TUM loaders on tiny synthetic directories written with Pillow: file formats, depth scale, intrinsics resize,
dilation / stride / start slicing, relative poses and frame-to-frame transforms, TUM timestamp association."""
import os

import numpy as np
import pytest
import torch
from PIL import Image


def _write_frames(d, n, H=12, W=16):
    os.makedirs(os.path.join(d, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(d, "depth"), exist_ok=True)
    for i in range(n):
        rgb = np.full((H, W, 3), i * 10, dtype=np.uint8)
        rgb[0, 0] = (255, 0, 7)
        Image.fromarray(rgb).save(os.path.join(d, "rgb", f"{i}.png"))
        dep = np.full((H, W), 5000 + 500 * i, dtype=np.uint16)          # 1.0 m + 0.1 m per frame
        dep[1, 1] = 0
        Image.fromarray(dep).save(os.path.join(d, "depth", f"{i}.png"))


def _pose(i):
    T = np.eye(4)
    a = 0.1 * i
    T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    T[:3, 3] = (0.5 * i, 0.1, -0.2 * i)
    return T


def test_icl_loader(tmp_path):
    from gradslam.datasets import ICL
    traj = tmp_path / "ICL" / "living_room_traj1_frei_png"
    n = 12
    _write_frames(str(traj), n)
    with open(traj / "associations.txt", "w") as f:
        for i in range(n):
            f.write(f"{i} depth/{i}.png {i} rgb/{i}.png\n")
    with open(traj / "livingRoom1n.gt.sim", "w") as f:
        for i in range(n):
            for r in _pose(i)[:3]:
                f.write(" ".join(f"{v:.8f}" for v in r) + "\n")
            f.write("\n")
    ds = ICL(str(tmp_path / "ICL"), seqlen=3, dilation=1, stride=2, start=2, height=12, width=16)
    assert len(ds) == 3                                     # starts 2, 4, 6 (span 5 frames)
    color, depth, K, pose, tr, names = ds[1]
    assert color.shape == (3, 12, 16, 3) and depth.shape == (3, 12, 16, 1) and K.shape == (1, 4, 4) and pose.shape == (3, 4, 4)
    assert [os.path.basename(x) for x in names] == ["4.png", "6.png", "8.png"]
    assert float(color[0, 5, 5, 0]) == 40.0 and tuple(color[0, 0, 0].tolist()) == (255.0, 0.0, 7.0)
    assert abs(float(depth[1, 3, 3, 0]) - 1.6) < 1e-6 and float(depth[0, 1, 1, 0]) == 0.0
    torch.testing.assert_close(K[0, :2, :3], torch.tensor([[481.2 * 16 / 640, 0, 319.5 * 16 / 640], [0, -480.0 * 12 / 480, 239.5 * 12 / 480]]))
    torch.testing.assert_close(pose[0], torch.eye(4), atol=1e-6, rtol=0)
    ref = torch.from_numpy(np.linalg.inv(_pose(4)) @ _pose(8)).float()
    torch.testing.assert_close(pose[2], ref, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(tr[2], torch.from_numpy(np.linalg.inv(_pose(6)) @ _pose(8)).float(), atol=1e-5, rtol=1e-5)
    big = ICL(str(tmp_path / "ICL"), seqlen=2, height=24, width=32)            # resize path: bilinear colour, nearest depth
    c2, d2, K2, *_ = big[0]
    assert c2.shape == (2, 24, 32, 3) and abs(float(d2[0].max()) - 1.0) < 1e-6 and abs(float(K2[0, 0, 0]) - 481.2 * 32 / 640) < 1e-4
    with pytest.raises(ValueError):
        ICL(str(tmp_path / "nope"))
    with pytest.raises(ValueError):
        ICL(str(tmp_path / "ICL"), seqlen=50)


def test_tum_loader(tmp_path):
    from gradslam.datasets import TUM
    seq = tmp_path / "TUM" / "rgbd_dataset_freiburg1_xyz"
    n = 8
    _write_frames(str(seq), n)
    with open(seq / "rgb.txt", "w") as f:
        f.write("# color images\n")
        for i in range(n):
            f.write(f"{100.0 + i * 0.033:.6f} rgb/{i}.png\n")
    with open(seq / "depth.txt", "w") as f:
        for i in range(n):
            if i != 3:                                       # a dropped depth frame: RGB frame 3 has no partner within 0.02 s
                f.write(f"{100.004 + i * 0.033:.6f} depth/{i}.png\n")
    with open(seq / "groundtruth.txt", "w") as f:
        f.write("# timestamp tx ty tz qx qy qz qw\n")
        for i in range(n * 4):
            t = 100.0 + i * 0.00825
            a = 0.05 * i
            f.write(f"{t:.6f} {0.01 * i:.6f} 0.0 0.0 0.0 0.0 {np.sin(a / 2):.8f} {np.cos(a / 2):.8f}\n")
    ds = TUM(str(tmp_path / "TUM"), seqlen=3, dilation=0, height=12, width=16)
    color, depth, K, pose, tr, names, stamps = ds[0]
    assert [os.path.basename(x) for x in names] == ["0.png", "1.png", "2.png"]
    assert len(ds) == 2 and [os.path.basename(x) for x in ds[1][5]] == ["4.png", "5.png", "6.png"]       # frame 3 skipped
    assert abs(float(K[0, 0, 0]) - 525.0 * 16 / 640) < 1e-4 and float(K[0, 1, 1]) > 0
    torch.testing.assert_close(pose[0], torch.eye(4), atol=1e-6, rtol=0)
    # frame 1 (t = 100.033) matches ground-truth sample 4 (t = 100.033): yaw 0.2 rad, x = 0.04
    assert abs(float(pose[1, 0, 3]) - 0.04) < 1e-5 and abs(float(pose[1, 0, 0]) - np.cos(0.2)) < 1e-5
    assert abs(float(depth[2, 4, 4, 0]) - 1.2) < 1e-6
