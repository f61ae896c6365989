"""TUM RGB-D loader with gradslam.datasets.TUM's contract (reference call site: online_adaption.py:77-84).
Layout under `basedir` (README.md:76-88): <basedir>/rgbd_dataset_freiburg1_xyz/{rgb.txt, depth.txt, groundtruth.txt, rgb/, depth/}.
RGB, depth and pose streams are associated by nearest timestamp (max difference 0.02 s), depth = png / 5000 m, the
ROS-default pinhole fx = fy = 525, cx = 319.5, cy = 239.5."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import datautils


def _read_list(path):
    out = []
    for line in open(path):
        line = line.strip()
        if line and not line.startswith("#"):
            p = line.split()
            out.append((float(p[0]), p[1:]))
    return out


def _associate(a, b, max_dt):
    """greedy nearest-timestamp matching of two sorted (t, payload) lists -> [(ia, ib)]"""
    tb = np.array([t for t, _ in b])
    pairs, used = [], set()
    for ia, (t, _) in enumerate(a):
        ib = int(np.argmin(np.abs(tb - t)))
        if abs(tb[ib] - t) < max_dt and ib not in used:
            used.add(ib)
            pairs.append((ia, ib))
    return pairs


class TUM(Dataset):
    DEPTH_SCALE = 5000.0

    def __init__(self, basedir, sequences=None, seqlen=4, dilation=None, stride=None, start=None, end=None, height=480, width=640,
                 channels_first=False, normalize_color=False, return_depth=True, return_intrinsics=True, return_pose=True,
                 return_transform=True, return_names=True, return_timestamps=True, max_dt=0.02):
        if channels_first:
            raise NotImplementedError("the reference uses channels-last frames only")
        if not os.path.isdir(basedir):
            raise ValueError(f"Base directory {basedir} does not exist")
        self.height, self.width, self.seqlen, self.normalize_color = int(height), int(width), int(seqlen), normalize_color
        if sequences is None:
            sequences = sorted(d for d in os.listdir(basedir) if os.path.isfile(os.path.join(basedir, d, "rgb.txt")))
        elif isinstance(sequences, str):
            sequences = [sequences]
        if not sequences:
            raise ValueError(f"No TUM sequences found under {basedir}")
        K = torch.eye(4)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 525.0, 525.0, 319.5, 239.5
        self.intrinsics = datautils.scale_intrinsics(K, self.height / 480.0, self.width / 640.0).unsqueeze(0)
        self.items, self.frames = [], {}
        for seq in sequences:
            sdir = os.path.join(basedir, seq)
            rgb, dep, gt = (_read_list(os.path.join(sdir, f)) for f in ("rgb.txt", "depth.txt", "groundtruth.txt"))
            frames = []
            dmatch = dict(_associate(rgb, dep, max_dt))
            pmatch = dict(_associate(rgb, gt, max_dt))
            for i, (t, payload) in enumerate(rgb):
                if i in dmatch and i in pmatch:
                    pose = datautils.quaternion_pose(*[float(v) for v in gt[pmatch[i]][1][:7]])
                    frames.append((os.path.join(sdir, payload[0]), os.path.join(sdir, dep[dmatch[i]][1][0]), pose, t))
            self.frames[seq] = frames
            starts, step = datautils.sequence_starts(len(frames), self.seqlen, dilation, stride, start, end)
            for s in starts:
                self.items.append((seq, [s + i * step for i in range(self.seqlen)]))
        if not self.items:
            raise ValueError("seqlen / dilation / start leave no complete sequence")

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        seq, ids = self.items[idx]
        fr = self.frames[seq]
        color = np.stack([datautils.read_color(fr[i][0], self.height, self.width) for i in ids])
        depth = np.stack([datautils.read_depth(fr[i][1], self.height, self.width, self.DEPTH_SCALE) for i in ids])
        if self.normalize_color:
            color = color / 255.0
        pose = datautils.relative_poses(torch.from_numpy(np.stack([fr[i][2] for i in ids])).float())
        names = [os.path.join(seq, os.path.basename(fr[i][0])) for i in ids]
        stamps = [fr[i][3] for i in ids]
        return (torch.from_numpy(color), torch.from_numpy(depth), self.intrinsics.clone(), pose, datautils.poses_to_transforms(pose), names, stamps)
