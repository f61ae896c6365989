"""Drop-in for depth_estimation/view_synthesis.py: same class names, constructor arguments, call
signatures and outputs; the arithmetic runs in the HIP kernels behind include/e2eslam.h.

reference: depth_estimation/view_synthesis.py:7-40 (BackprojectDepth), :42-78 (Project3D).
"""
import torch.nn as nn

from e2ehip import ops


class BackprojectDepth(nn.Module):
    """depth (B,1,H,W) + inverse intrinsics (B,4,4) -> homogeneous camera points (B,4,H*W).
    The reference pre-builds a (B,3,H*W) pixel grid parameter; here pixel coordinates are generated in
    the kernel, so the module holds no buffers."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth, inv_K):
        if depth.shape[0] != self.batch_size or tuple(depth.shape[-2:]) != (self.height, self.width):
            raise RuntimeError(f"BackprojectDepth built for ({self.batch_size},1,{self.height},{self.width}), got {tuple(depth.shape)}")
        return ops.backproject(depth.reshape(self.batch_size, 1, self.height, self.width), inv_K)


class Project3D(nn.Module):
    """camera points (B,4,H*W), K, T (B,4,4) -> sampling grid (B,H,W,2) [+ depth (B,1,H,W)] + valid mask (B,1,H,W)."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        if eps != 1e-7:
            raise ValueError("the HIP kernel fixes eps = 1e-7 (the reference's default, view_synthesis.py:45)")
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T, geometric):
        return ops.project3d(points, K, T, self.height, self.width, geometric)
