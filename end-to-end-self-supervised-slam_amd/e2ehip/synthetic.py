"""Synthetic RGB-D sequences with exact geometry (there are no datasets on the build / GPU machines).

A textured plane seen by a moving pinhole camera: per pixel the ray/plane intersection gives the metric depth
and the world point, whose analytic texture gives the colour -- so consecutive frames are photometrically
consistent under the true poses, which is what the refinement's warp + photometric loss needs.
Intrinsics default to gradslam's ICL values (fx=481.2, fy=-480, cx=319.5, cy=239.5 at 640x480)."""
import math

import torch


def icl_intrinsics(H, W):
    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 481.2 * W / 640, -480.0 * H / 480, 319.5 * W / 640, 239.5 * H / 480
    return K


def tum_intrinsics(H, W):
    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 525.0 * W / 640, 525.0 * H / 480, 319.5 * W / 640, 239.5 * H / 480
    return K


def _pose(i, step):
    """camera-to-world pose of frame i: sideways translation `step` m per frame plus a slow pan."""
    a = math.radians(0.35 * i)
    R = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1.0, 0], [-math.sin(a), 0, math.cos(a)]])
    T = torch.eye(4)
    T[:3, :3] = R
    T[:3, 3] = torch.tensor([step * i, 0.01 * math.sin(0.3 * i), 0.0])
    return T


def make_sequence(L, H, W, seed=1234, step=0.06, K=None, noise=0.01, holes=0.0, scene="plane"):
    """-> colors (1,L,H,W,3) in [0,1], depths (1,L,H,W,1) metres, intrinsics (1,1,4,4), poses (1,L,4,4).
    scene: "plane" (one slanted textured plane) or "corner" (three planes meeting in a room corner: constrains all six
    degrees of freedom, which frame-to-model ICP needs -- a single plane lets the estimate slide)."""
    g = torch.Generator().manual_seed(seed)
    K = icl_intrinsics(H, W) if K is None else K
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    vs, us = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    ray_c = torch.stack([(us - cx) / fx, (vs - cy) / fy, torch.ones(H, W)], -1)
    planes = [(torch.tensor([0.15, -0.1, -1.0]), -2.2)]          # plane n.X = d, about 2.2 m in front of the first camera
    if scene == "corner":
        planes += [(torch.tensor([1.0, 0.1, -0.35]), -1.6), (torch.tensor([0.05, 1.0, -0.3]), -1.3)]
    elif scene != "plane":
        raise ValueError(f"unknown scene {scene}")
    planes = [(n / n.norm(), d / float(n.norm())) for n, d in planes]
    ph = torch.rand(6, generator=g) * 6.28
    colors, depths, poses = [], [], []
    for i in range(L):
        T = _pose(i, step)
        ray_w = ray_c @ T[:3, :3].T
        s = torch.full((H, W), float("inf"))
        for n, d_plane in planes:                                  # nearest positive ray/plane intersection
            sk = (d_plane - (n * T[:3, 3]).sum()) / (ray_w * n).sum(-1)
            s = torch.where((sk > 0.05) & (sk < s), sk, s)
        X = T[:3, 3] + s.unsqueeze(-1) * ray_w
        tex = [0.5 + 0.22 * torch.sin(X[..., 0] * (7 + 2 * c) + ph[c]) * torch.cos(X[..., 1] * (5 + 3 * c) + ph[3 + c])
               + 0.18 * torch.sin((X[..., 0] + X[..., 1]) * (17 + 5 * c)) for c in range(3)]
        col = (torch.stack(tex, -1) + noise * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
        dep = s.clone()
        if holes > 0:
            dep[torch.rand(H, W, generator=g) < holes] = 0.0
        colors.append(col)
        depths.append(dep.unsqueeze(-1))
        poses.append(T)
    return torch.stack(colors)[None], torch.stack(depths)[None], K[None, None], torch.stack(poses)[None]
