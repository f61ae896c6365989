"""ICL-NUIM loader with gradslam.datasets.ICL's contract (reference call site: online_adaption.py:69-75):
item = (colors (L,H,W,3) float 0..255, depths (L,H,W,1) metres, intrinsics (1,4,4), poses (L,4,4) relative to the
item's first frame, transforms (L,4,4), names).  Layout expected under `basedir` (README.md:58-72):
    <basedir>/living_room_traj1_frei_png/{rgb/*.png, depth/*.png, associations.txt, livingRoom1n.gt.sim}
"""
import glob
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import datautils


class ICL(Dataset):
    DEPTH_SCALE = 5000.0

    def __init__(self, basedir, trajectories=None, seqlen=4, dilation=None, stride=None, start=None, end=None, height=480, width=640,
                 channels_first=False, normalize_color=False, return_depth=True, return_intrinsics=True, return_pose=True,
                 return_transform=True, return_names=True):
        if channels_first:
            raise NotImplementedError("the reference uses channels-last frames only")
        if not os.path.isdir(basedir):
            raise ValueError(f"Base directory {basedir} does not exist")
        self.height, self.width, self.seqlen, self.normalize_color = int(height), int(width), int(seqlen), normalize_color
        if trajectories is None:
            trajectories = sorted(d for d in os.listdir(basedir) if os.path.isdir(os.path.join(basedir, d)) and d.endswith("frei_png"))
        elif isinstance(trajectories, str):
            trajectories = [trajectories]
        if not trajectories:
            raise ValueError(f"No ICL trajectories found under {basedir}")
        K = torch.eye(4)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 481.20, -480.0, 319.5, 239.5
        self.intrinsics = datautils.scale_intrinsics(K, self.height / 480.0, self.width / 640.0).unsqueeze(0)
        self.items = []          # (trajectory, [frame indices])
        self.frames = {}
        for traj in trajectories:
            tdir = os.path.join(basedir, traj)
            frames = self._read_associations(tdir)
            poses = self._read_poses(tdir, len(frames))
            self.frames[traj] = (tdir, frames, poses)
            starts, step = datautils.sequence_starts(len(frames), self.seqlen, dilation, stride, start, end)
            for s in starts:
                self.items.append((traj, [s + i * step for i in range(self.seqlen)]))
        if not self.items:
            raise ValueError("seqlen / dilation / start leave no complete sequence")

    @staticmethod
    def _read_associations(tdir):
        assoc = os.path.join(tdir, "associations.txt")
        out = []
        if os.path.isfile(assoc):
            for line in open(assoc):
                p = line.split()
                if len(p) >= 4:
                    out.append((os.path.join(tdir, p[3]), os.path.join(tdir, p[1])))      # (rgb, depth)
        else:
            rgbs = sorted(glob.glob(os.path.join(tdir, "rgb", "*.png")), key=lambda f: int(os.path.splitext(os.path.basename(f))[0]))
            out = [(r, os.path.join(tdir, "depth", os.path.basename(r))) for r in rgbs]
        return out

    @staticmethod
    def _read_poses(tdir, n):
        sims = glob.glob(os.path.join(tdir, "*.gt.sim"))
        if not sims:
            raise ValueError(f"No ground-truth pose file (*.gt.sim) in {tdir}")
        vals = np.loadtxt(sims[0]).reshape(-1, 3, 4)
        poses = np.tile(np.eye(4), (vals.shape[0], 1, 1))
        poses[:, :3, :] = vals
        if poses.shape[0] < n:
            raise ValueError(f"{sims[0]} holds {poses.shape[0]} poses for {n} frames")
        return torch.from_numpy(poses[:n]).float()

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        traj, ids = self.items[idx]
        _, frames, poses = self.frames[traj]
        color = np.stack([datautils.read_color(frames[i][0], self.height, self.width) for i in ids])
        depth = np.stack([datautils.read_depth(frames[i][1], self.height, self.width, self.DEPTH_SCALE) for i in ids])
        if self.normalize_color:
            color = color / 255.0
        pose = datautils.relative_poses(poses[ids])
        names = [os.path.join(traj, os.path.basename(frames[i][0])) for i in ids]
        return (torch.from_numpy(color), torch.from_numpy(depth), self.intrinsics.clone(), pose, datautils.poses_to_transforms(pose), names)
