"""N2 on the GPU path: a sequence that comes out of the ICL / TUM loaders (PNG files on disk, written here with Pillow) through
`SLAM.dataset_init`'s loader branch -- the reference's `ICL(...)` / `TUM(...)` + `DataLoader` + whole-sequence upload + `colors /= 255`
(online_adaption.py:59-96, :212-220) -- gives the same run, bit for bit, as handing `SLAM` the loader's tensors directly."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import depthnet

pytestmark = pytest.mark.gpu
H, W, L = 64, 96, 3


def _frames():
    from e2ehip.synthetic import make_sequence
    colors, depths, K, poses = make_sequence(L, H, W, seed=17)
    rgb = (colors[0] * 255).round().clamp(0, 255).to(torch.uint8).numpy()                      # (L,H,W,3)
    dep = (depths[0, ..., 0] * 5000).round().clamp(0, 65535).to(torch.int32).numpy().astype(np.uint16)
    return rgb, dep, poses[0].double().numpy()


def _write_icl(root):
    rgb, dep, poses = _frames()
    traj = os.path.join(root, "ICL", "living_room_traj1_frei_png")
    os.makedirs(os.path.join(traj, "rgb")); os.makedirs(os.path.join(traj, "depth"))
    with open(os.path.join(traj, "associations.txt"), "w") as fa, open(os.path.join(traj, "livingRoom1n.gt.sim"), "w") as fp:
        for i in range(L):
            Image.fromarray(rgb[i]).save(os.path.join(traj, "rgb", f"{i}.png"))
            Image.fromarray(dep[i]).save(os.path.join(traj, "depth", f"{i}.png"))
            fa.write(f"{i} depth/{i}.png {i} rgb/{i}.png\n")
            for r in poses[i][:3]:
                fp.write(" ".join(f"{v:.9f}" for v in r) + "\n")
            fp.write("\n")


def _quat(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    return (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w


def _write_tum(root):
    rgb, dep, poses = _frames()
    seq = os.path.join(root, "TUM", "rgbd_dataset_freiburg1_xyz")
    os.makedirs(os.path.join(seq, "rgb")); os.makedirs(os.path.join(seq, "depth"))
    with open(os.path.join(seq, "rgb.txt"), "w") as fr, open(os.path.join(seq, "depth.txt"), "w") as fd, open(os.path.join(seq, "groundtruth.txt"), "w") as fg:
        fg.write("# timestamp tx ty tz qx qy qz qw\n")
        for i in range(L):
            Image.fromarray(rgb[i]).save(os.path.join(seq, "rgb", f"{i}.png"))
            Image.fromarray(dep[i]).save(os.path.join(seq, "depth", f"{i}.png"))
            t = 100.0 + 0.1 * i
            fr.write(f"{t:.6f} rgb/{i}.png\n")
            fd.write(f"{t + 0.003:.6f} depth/{i}.png\n")
            qx, qy, qz, qw = _quat(poses[i][:3, :3])
            fg.write(f"{t:.6f} {poses[i][0, 3]:.9f} {poses[i][1, 3]:.9f} {poses[i][2, 3]:.9f} {qx:.9f} {qy:.9f} {qz:.9f} {qw:.9f}\n")


@pytest.mark.parametrize("name", ["ICL", "TUM"])
def test_loader_fed_run_equals_tensor_fed_run(tmp_path, name):
    from gradslam.datasets import ICL, TUM
    from online_adaption import SLAM, default_config
    (_write_icl if name == "ICL" else _write_tum)(str(tmp_path))
    sd = depthnet.random_state_dict(0)
    sd["decoder.decoder.10.conv.weight"] = sd["decoder.decoder.10.conv.weight"] * 40.0

    def cfg():
        c = default_config(H, W, L)
        c.DATA.name, c.DATA.data_path = name, str(tmp_path)
        c.DATA.dilation, c.DATA.stride, c.DATA.start = 0, 1, 0
        c.DEMO.frame_threshold = 0.0
        return c
    # (a) through dataset_init's loader branch
    a = SLAM(cfg(), state_dict=sd)
    assert a.colors.shape == (1, L, H, W, 3) and float(a.colors.max()) <= 1.0 and float(a.colors.max()) > 0.5
    a.main()
    # (b) the same tensors handed over directly
    ds = (ICL if name == "ICL" else TUM)(basedir=os.path.join(str(tmp_path), name), seqlen=L, height=H, width=W, dilation=0, stride=1, start=0)
    colors, depths, K, poses = ds[0][:4]
    b = SLAM(cfg(), sequence=(colors[None] / 255.0, depths[None], K[None], poses[None]), state_dict=sd)
    b.main()
    la, lb = torch.stack(a.log), torch.stack(b.log)
    assert la.shape == (3 * (L - 1), 12) and torch.equal(la, lb)
    assert a.map.M == b.map.M and a.map.M > H * W
    for x, y in zip(a.map.live(), b.map.live()):
        assert torch.equal(x, y)
    if name == "TUM":
        assert float(a.intrinsics[0, 0, 1, 1]) > 0                  # TUM's positive fy (ICL: negative)
    # the refinement did something on this sequence: the photometric loss of the last step is below the first one's
    assert float(la[-1, 1]) < float(la[0, 1]) * 1.2 and torch.isfinite(la).all()
