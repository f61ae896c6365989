"""CPU restatement of chamferdist.knn_points (K=1, D=3) and the 3-D point loss built on it.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: chamferdist is not vendored /
installed (README.md:17-19); semantics per SURVEY.md Appendix A and the call site
loss/losses.py:57-61 (squared distances, int64 indices, first minimum wins).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle_knn.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.knn1_brute.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                    ctypes.c_void_p, ctypes.c_void_p]
        _LIB.knn1_brute.restype = None
    return _LIB


def knn1(p1, p2):
    """p1 (P1,3), p2 (P2,3) fp32 -> (dists (P1,) squared, idx (P1,) int64). No autograd."""
    a = np.ascontiguousarray(p1.detach().numpy(), dtype=np.float32)
    b = np.ascontiguousarray(p2.detach().numpy(), dtype=np.float32)
    d = np.empty(a.shape[0], dtype=np.float32)
    i = np.empty(a.shape[0], dtype=np.int64)
    _lib().knn1_brute(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], d.ctypes.data, i.ctypes.data)
    return torch.from_numpy(d), torch.from_numpy(i)


def knn1_torch(p1, p2, chunk=2048):
    """Same thing with torch ops (cross-check of the C loop)."""
    ds, ids = [], []
    for s in range(0, p1.shape[0], chunk):
        q = p1[s:s + chunk, None, :] - p2[None, :, :]
        d = (q[..., 0] * q[..., 0] + q[..., 1] * q[..., 1]) + q[..., 2] * q[..., 2]
        m = d.min(dim=1)[0]
        first = (d == m[:, None]).float().argmax(dim=1)
        ds.append(m); ids.append(first)
    return torch.cat(ds), torch.cat(ids)


def knn_points_loss(gt_points, noisy_points):
    """gt (1,M,3), noisy (1,P,3) -> (mean squared NN distance, idx (1,P)).
    reference: loss/losses.py:39-63; autograd: d/dp1 = 2 g (p1 - p2[idx])."""
    if gt_points.shape[0] != noisy_points.shape[0]:
        raise ValueError("Pointclouds must have the same batch dimension")
    if gt_points.shape[2] != noisy_points.shape[2]:
        raise ValueError("Number of axes is not the same in both pointclouds")
    _, idx = knn1(noisy_points[0], gt_points[0])
    q = noisy_points[0] - gt_points[0][idx]
    d = (q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]
    return d.mean(), idx.unsqueeze(0)


def chamfer_distance(source, target, bidirectional=False, reverse=False, reduction="mean"):
    """chamferdist.ChamferDistance.forward as SURVEY.md Appendix A restates it (un-vendored, PARITY UNPINNED): the `reduction` of the
    squared nearest-neighbour distances source -> target (+ target -> source when bidirectional, only that when reverse); call site
    train_depth.py:690-692.  source (1,P,3), target (1,M,3).  Gradients reach BOTH clouds (knn_points' backward: d p1 = 2 g (p1 - p2[idx]),
    d p2 = the scatter of the negative), which autograd derives here from the gathered difference."""
    red = {"mean": torch.mean, "sum": torch.sum}[reduction]

    def one_way(a, b):
        _, idx = knn1(a[0], b[0])
        q = a[0] - b[0][idx]
        return red((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2])
    fwd = one_way(source, target)
    if bidirectional:
        return fwd + one_way(target, source)
    return one_way(target, source) if reverse else fwd


def color_points_loss(gt_color, noisy_color, indexes):
    """loss/losses.py:65-82: mean |noisy colour - colour of its nearest ground-truth point|; (1,M,3), (1,P,3), idx (1,P)."""
    if gt_color.shape[2] != noisy_color.shape[2]:
        raise ValueError("Number of axes is not the same in both pointclouds")
    return torch.mean(torch.abs(noisy_color[0] - gt_color[0, indexes[0].long()]))
