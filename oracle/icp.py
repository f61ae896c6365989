"""CPU restatement of gradslam's frame-to-model ICP odometry (MODEL.odom: icp / gradicp, configs/config.yaml:30;
reached through PointFusion.step at online_adaption.py:362-363 and train_depth.py:378-382).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: gradslam is an un-vendored, un-pinned dependency;
this follows the recalled structure of gradslam/odometry/{icp,gradicp,icputils}.py:
  * source  = the live frame's valid vertices, every `dsratio`-th pixel in both directions, placed in the world with the
              PREVIOUS frame's pose (initial guess);
  * target  = the map points that project into the previous frame (find_active_map_points), every `dsratio`-th of them,
              with their normals;
  * numiters Gauss-Newton steps of point-to-plane ICP: nearest target point for every source point, rows
              A_i = [n_i, s_i x n_i], b_i = n_i . (t_i - s_i), solve (A^T A + damp I) xi = A^T b, T <- exp(xi) T;
  * live pose = T . previous pose.
"gradicp" uses the same residuals with a Levenberg-Marquardt damping that is updated by a smooth (generalised logistic)
function of the error change instead of a hard accept/reject, so that the pose stays differentiable; the reference never
differentiates through the pose (it is discarded at online_adaption.py:362), so only the forward value matters here.
All linear algebra is float64 (both here and in the HIP path's host step) to keep the comparison meaningful.
"""
import numpy as np
import torch

from . import knn, pointfusion


def so3_hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def se3_exp(xi):
    """xi = (v, omega) -> 4x4 (Rodrigues + the V matrix), float64."""
    v, w = np.asarray(xi[:3], np.float64), np.asarray(xi[3:], np.float64)
    th = np.linalg.norm(w)
    W = so3_hat(w)
    if th < 1e-8:
        R = np.eye(3) + W + 0.5 * W @ W
        V = np.eye(3) + 0.5 * W + W @ W / 6.0
    else:
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W @ W
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


def normal_equations(src, tgt, tgt_n, idx, dists, dist_thresh):
    """27 sums of the point-to-plane system in float64 + (count, sum of squared residuals)."""
    s = src.double().numpy()
    t = tgt.double().numpy()[idx.numpy()]
    n = tgt_n.double().numpy()[idx.numpy()]
    keep = np.ones(len(s), bool) if dist_thresh is None else (dists.double().numpy() < float(dist_thresh) ** 2)
    s, t, n = s[keep], t[keep], n[keep]
    A = np.concatenate([n, np.cross(s, n)], 1)
    b = (n * (t - s)).sum(1)
    return A.T @ A, A.T @ b, int(keep.sum()), float((b * b).sum())


def point_to_plane_icp(src, tgt, tgt_n, numiters=20, damp=1e-8, dist_thresh=None, mode="icp", lambda_max=2.0, B=1.0, B2=1.0, nu=200.0):
    """src (Ns,3), tgt / tgt_n (Nt,3) float32 tensors -> 4x4 float64 transform aligning src to tgt, plus a trace."""
    T = np.eye(4)
    lam = float(damp)
    trace = []
    for _ in range(numiters):
        cur = (src.double() @ torch.from_numpy(T[:3, :3]).T + torch.from_numpy(T[:3, 3])).float()
        d, idx = knn.knn1(cur, tgt)
        AtA, Atb, cnt, err = normal_equations(cur, tgt, tgt_n, idx, d, dist_thresh)
        if cnt < 6:
            break
        xi = np.linalg.solve(AtA + lam * np.eye(6), Atb)
        step = se3_exp(xi)
        if mode == "gradicp":
            nxt = (cur.double() @ torch.from_numpy(step[:3, :3]).T + torch.from_numpy(step[:3, 3])).float()
            d2, idx2 = knn.knn1(nxt, tgt)
            _, _, cnt2, err2 = normal_equations(nxt, tgt, tgt_n, idx2, d2, dist_thresh)
            # generalised logistic damping update: grows towards lambda_max when the step made things worse,
            # shrinks towards 1/lambda_max when it helped (smooth in the error change)
            delta = (err2 / max(cnt2, 1)) - (err / max(cnt, 1))
            q = 1.0 / lambda_max + (lambda_max - 1.0 / lambda_max) / (1.0 + B * np.exp(-B2 * nu * delta)) ** (1.0 / 1.0)
            lam = lam * q
            gate = 1.0 / (1.0 + np.exp(np.clip(nu * delta, -60, 60)))          # ~1 when the error decreased
            step = se3_exp(gate * xi)
        T = step @ T
        trace.append((cnt, err))
    return T, trace


def frame_to_model(map_points, map_normals, depth, K, prev_pose, dsratio=4, **kw):
    """The odometry of PointFusion._localize: returns the live pose (4,4) float64."""
    H, W = depth.shape
    maps = pointfusion.vertex_normal_maps(depth, K, prev_pose)
    sub = torch.zeros(H, W, dtype=torch.bool)
    sub[::dsratio, ::dsratio] = True
    src = maps["Vg"][maps["valid"] & sub]
    active = pointfusion.find_active_map_points(map_points, K, prev_pose, H, W)
    sel = active[::dsratio, 0]
    T, trace = point_to_plane_icp(src, map_points[sel], map_normals[sel], **kw)
    return T @ prev_pose.double().numpy(), trace
