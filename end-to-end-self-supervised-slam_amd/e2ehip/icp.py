"""Frame-to-model ICP odometry (gradslam's "icp" / "gradicp" odometry providers).

Per Gauss-Newton iteration the GPU transforms the source cloud, finds exact nearest neighbours in the active map
points and folds the N x 6 point-to-plane system into 29 numbers (csrc/icp.hip); the 6x6 solve and the se(3)
exponential are float64 on the host (one 232-byte copy per iteration; odometry runs once per keyframe, not per
refinement step)."""
import numpy as np
import torch

from . import _lib as L
from . import ops


def _so3_hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def se3_exp(xi):
    v, w = np.asarray(xi[:3], np.float64), np.asarray(xi[3:], np.float64)
    th = np.linalg.norm(w)
    W = _so3_hat(w)
    if th < 1e-8:
        R, V = np.eye(3) + W + 0.5 * W @ W, np.eye(3) + 0.5 * W + W @ W / 6.0
    else:
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W @ W
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, V @ v
    return T


def _unpack(v):
    AtA = np.zeros((6, 6))
    k = 0
    for r in range(6):
        for c in range(r, 6):
            AtA[r, c] = AtA[c, r] = v[k]
            k += 1
    return AtA, v[21:27], int(round(v[27])), float(v[28])


def _reduce(cur, tgt, tgt_n, dist_thresh, out, ws, index=None):
    d, idx = ops.knn1(cur, index if index is not None else tgt)
    L.call("e2e_icp_normal_equations", L.ptr(cur), L.ptr(tgt), L.ptr(tgt_n), L.ptr(idx), L.ptr(d),
           -1.0 if dist_thresh is None else float(dist_thresh), cur.shape[0], L.ptr(out), L.ptr(ws), L.stream())
    return _unpack(out.cpu().numpy())


def point_to_plane_icp(src, tgt, tgt_n, numiters=20, damp=1e-8, dist_thresh=None, mode="icp", lambda_max=2.0, B=1.0, B2=1.0, nu=200.0):
    """src (Ns,3), tgt / tgt_n (Nt,3) device tensors -> 4x4 float64 numpy transform aligning src to tgt, plus a trace."""
    if mode not in ("icp", "gradicp"):
        raise ValueError(f"unknown odometry mode {mode}")
    src, tgt, tgt_n = (L.dev(t, n).contiguous() for t, n in ((src, "src"), (tgt, "tgt"), (tgt_n, "tgt_normals")))
    dev = src.device
    out = torch.empty(29, device=dev, dtype=torch.float64)
    ws = torch.empty(L.load().e2e_icp_workspace_bytes(), device=dev, dtype=torch.uint8)
    T = np.eye(4)
    lam = float(damp)
    trace = []
    index = ops.KnnIndex(tgt, src.shape[0])               # the target cloud is fixed: one grid for all iterations
    for _ in range(numiters):
        cur = ops.transform_points(src, torch.from_numpy(T).float().to(dev))
        AtA, Atb, cnt, err = _reduce(cur, tgt, tgt_n, dist_thresh, out, ws, index)
        if cnt < 6:
            break
        xi = np.linalg.solve(AtA + lam * np.eye(6), Atb)
        step = se3_exp(xi)
        if mode == "gradicp":
            nxt = ops.transform_points(cur, torch.from_numpy(step).float().to(dev))
            _, _, cnt2, err2 = _reduce(nxt, tgt, tgt_n, dist_thresh, out, ws, index)
            delta = (err2 / max(cnt2, 1)) - (err / max(cnt, 1))
            lam = lam * (1.0 / lambda_max + (lambda_max - 1.0 / lambda_max) / (1.0 + B * np.exp(-B2 * nu * delta)))
            step = se3_exp(xi / (1.0 + np.exp(np.clip(nu * delta, -60, 60))))
        T = step @ T
        trace.append((cnt, err))
    return T, trace


def frame_to_model(fmap, depth, K, prev_pose, dsratio=4, **kw):
    """PointFusion._localize: pose of the live frame (depth (H,W)) given the resident map `fmap` (e2ehip.FusionMap)
    and the previous frame's pose.  Returns a (4,4) float32 device tensor and the iteration trace."""
    H, W = fmap.H, fmap.W
    if fmap.M == 0:
        raise ValueError("frame-to-model odometry needs a non-empty map")
    with torch.no_grad():
        maps = fmap.frame_maps(depth, K, prev_pose)
        sub = torch.zeros(H, W, dtype=torch.bool, device=depth.device)
        sub[::dsratio, ::dsratio] = True
        src = maps["Vg"][0][maps["valid"][0] & sub]
        fmap.associate(maps, K, prev_pose)
        sel = fmap.table("active")[::dsratio, 0]
        if sel.numel() < 6 or src.shape[0] < 6:
            raise RuntimeError("too few points for frame-to-model ICP (no overlap between the live frame and the map)")
        T, trace = point_to_plane_icp(src, fmap.points[sel], fmap.normals[sel], **kw)
        pose = torch.from_numpy(T @ prev_pose.detach().double().cpu().numpy()).float().to(depth.device)
    return pose, trace
