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


class ResidentOdometry:
    """PointFusion._localize for the driver's resident map WITHOUT a host round trip: source / target selection, the nearest-neighbour
    index over the targets and the `numiters` Gauss-Newton (icp) or Levenberg-Marquardt (gradicp) iterations are a fixed sequence of
    launches over buffers allocated once -- sizes that depend on the map (live points, active points, targets) are device data -- so a
    keyframe's odometry can sit inside the captured map-update graph (RefineStepPlan.update_map_odom).  Same arithmetic as
    frame_to_model / point_to_plane_icp above (which remain for ad-hoc clouds and as the cross-check) with the 6x6 solve, the se(3)
    exponential and the damping update in the e2e_icp_update kernel instead of numpy.

    Requires a live frame whose selected pixels all have depth (network predictions 1 / disp do); a hole raises `status`, read by
    check() together with the target-capacity overflow flag wherever the host synchronises anyway."""

    def __init__(self, fmap, dsratio=4, numiters=20, mode="gradicp", damp=1e-8, dist_thresh=None, lambda_max=2.0, B=1.0, B2=1.0, nu=200.0,
                 target_capacity=None, grid_cells=256):
        if mode not in ("icp", "gradicp"):
            raise ValueError(f"unknown odometry mode {mode}")
        self.map, self.ds, self.numiters, self.mode = fmap, int(dsratio), int(numiters), mode
        self.damp, self.dist_thresh = float(damp), dist_thresh
        self.lm = (float(lambda_max), float(B), float(B2), float(nu))
        H, W, dev = fmap.H, fmap.W, fmap.device
        self.n_src = ((H + self.ds - 1) // self.ds) * ((W + self.ds - 1) // self.ds)
        # active points are map points that project into ONE frame: a few per pixel at most
        self.tcap = int(target_capacity or min(fmap.cap, 8 * H * W) // self.ds + 1)
        f = dict(device=dev, dtype=torch.float32)
        lib = L.load()
        self.Vg, self.Ng = torch.empty(1, H, W, 3, **f), torch.empty(1, H, W, 3, **f)
        self.alpha = torch.empty(1, H, W, **f)
        self.src, self.cur, self.nxt = (torch.empty(self.n_src, 3, **f) for _ in range(3))
        self.tgt, self.tgt_n = torch.empty(self.tcap, 3, **f), torch.empty(self.tcap, 3, **f)
        self.tcount = torch.zeros(3, device=dev, dtype=torch.int64)
        self.status = torch.zeros(1, device=dev, dtype=torch.int32)
        # a coarse grid: 19 200 queries of a sparse target set, decimetres away while the pose is still wrong (e2e_knn1_index_*_res)
        self.cells = int(grid_cells)
        self.index = torch.empty(lib.e2e_knn1_index_capacity_bytes_res(self.n_src, self.tcap, self.cells), device=dev, dtype=torch.uint8)
        self.d = torch.empty(self.n_src, **f)
        self.idx = torch.empty(self.n_src, device=dev, dtype=torch.int64)
        self.out29 = torch.empty(29, device=dev, dtype=torch.float64)
        self.ws = torch.empty(lib.e2e_icp_workspace_bytes(), device=dev, dtype=torch.uint8)
        self.state = torch.zeros(lib.e2e_icp_state_doubles(), device=dev, dtype=torch.float64)
        self.T32, self.step32 = torch.eye(4, **f), torch.eye(4, **f)
        self.pose = torch.eye(4, **f)                       # the result: live pose (4,4), rewritten by every run()
        from .ops import fusion_alpha_den
        self._alpha_den = float(fusion_alpha_den(fmap.sigma))

    def _search_reduce_update(self, pts, st, warm, phase, prev_pose):
        # every search after the first of a keyframe starts from the previous search's neighbours (same source points, moved by one small
        # step; same targets): a real target point's distance bounds the ball from the start -- exact for any candidate
        L.call("e2e_knn1_index_query_dev_res", L.ptr(pts), self.n_src, L.ptr(self.tgt) if warm else None, L.ptr(self.idx) if warm else None, self.tcap,
               self.n_src, L.ptr(self.index), self.cells, L.ptr(self.d), L.ptr(self.idx), st)
        L.call("e2e_icp_reduce_update", L.ptr(pts), L.ptr(self.tgt), L.ptr(self.tgt_n), L.ptr(self.idx), L.ptr(self.d),
               -1.0 if self.dist_thresh is None else float(self.dist_thresh), self.n_src, L.ptr(self.ws), L.ptr(self.state), L.ptr(self.T32),
               L.ptr(self.step32), L.ptr(prev_pose), L.ptr(self.pose), 1 if self.mode == "gradicp" else 0, phase, *self.lm, st)

    def run(self, depth, K, prev_pose):
        """depth (H,W) of the live frame, K (4,4), prev_pose (4,4): contiguous device tensors (resident buffers when this is captured).
        Leaves the estimated live pose in self.pose and returns it.  No host synchronisation, no allocation."""
        m, st = self.map, L.stream()
        H, W = m.H, m.W
        for n, t in (("depth", depth), ("K", K), ("prev_pose", prev_pose)):
            if not L.dev(t, n).is_contiguous():
                raise ValueError(f"ResidentOdometry.run: {n} must be contiguous")
        # the live frame placed with the PREVIOUS pose (initial guess), and the map points active in that view
        L.call("e2e_vertex_normal_maps", L.ptr(depth), L.ptr(K), L.ptr(prev_pose), self._alpha_den, None, None, L.ptr(self.Vg), L.ptr(self.Ng),
               L.ptr(self.alpha), 1, H, W, st)
        L.call("e2e_pf_associate_dev", L.ptr(m.points), L.ptr(m.normals), L.ptr(m.ccounts), L.ptr(m.count), L.ptr(K), L.ptr(prev_pose),
               L.ptr(self.Vg), L.ptr(self.Ng), m.dist_th, m.dot_th, L.ptr(m.ws), m.cap, H, W, st)
        L.call("e2e_pf_active_subsample_dev", L.ptr(m.points), L.ptr(m.normals), L.ptr(m.count), m.cap, L.ptr(m.ws), H, W, self.ds,
               L.ptr(self.tgt), L.ptr(self.tgt_n), L.ptr(self.tcount), self.tcap, st)
        L.call("e2e_icp_source_subsample", L.ptr(self.Vg), L.ptr(depth), H, W, self.ds, L.ptr(self.src), L.ptr(self.status), st)
        L.call("e2e_knn1_index_build_dev_res", L.ptr(self.tgt), L.ptr(self.tcount), self.tcap, self.n_src, L.ptr(self.index), self.cells, st)
        L.call("e2e_icp_state_init", L.ptr(self.state), L.ptr(self.T32), L.ptr(self.step32), L.ptr(prev_pose), L.ptr(self.pose), self.damp, st)
        for it in range(self.numiters):
            L.call("e2e_transform_points", L.ptr(self.src), L.ptr(self.T32), L.ptr(self.cur), self.n_src, 0, st)
            self._search_reduce_update(self.cur, st, it > 0, 0, prev_pose)
            if self.mode == "gradicp":
                L.call("e2e_transform_points", L.ptr(self.cur), L.ptr(self.step32), L.ptr(self.nxt), self.n_src, 0, st)
                self._search_reduce_update(self.nxt, st, True, 1, prev_pose)
        m._assoc_M = None
        return self.pose

    def check(self):
        """Host read (one sync) of the two error flags of all runs so far; raises if either fired.  Returns (iterations done by the last run,
        its trace [(inliers, sum r^2)], targets, active points)."""
        st, tc, status = self.state.cpu(), self.tcount.cpu(), int(self.status.item())
        if status:
            raise RuntimeError("resident odometry: a selected pixel of the live frame had no depth (use icp.frame_to_model for depth maps with holes)")
        if int(tc[2]):
            raise RuntimeError(f"resident odometry: more than {self.tcap} target points; raise target_capacity")
        it = int(st[25])
        return it, [(int(st[32 + 2 * k]), float(st[33 + 2 * k])) for k in range(min(it, 64))], int(tc[0]), int(tc[1])
