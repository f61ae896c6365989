"""CPU restatement (torch, fp32 / int64) of the PointFusion map step and the RGB-D unprojection.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: the arithmetic lives in `gradslam` (README.md:9-21, no version pin; era v0.1.0),
which is neither vendored under /root/reference nor installed.  The restatement follows
SURVEY.md Appendix A (recalled semantics of gradslam/structures/rgbdimages.py,
gradslam/slam/fusionutils.py) and is anchored on the reference's call sites:
online_adaption.py:347-363 (map step), :461-469 (local cloud), slam/custom_slam.py:21-34.

Because the index tables / masks are bit-exact targets, every fp32 expression that feeds a
comparison is written with an EXPLICIT operation order (no matmul/einsum whose summation
order is a library detail, no fused multiply-add); the HIP kernels evaluate the same
expressions in the same order with -ffp-contract=off.

Single-sequence form: batch B = 1, frame sequence length 1 (every reference call site).
"""
import math

import torch


def sqrt_rn(x):
    """IEEE (correctly rounded) fp32 square root.  torch's CPU float sqrt is a vectorised approximation that
    is NOT always correctly rounded (7.6e3 of 1e6 random inputs differ from the exact result), so it cannot
    serve as a bit-exact specification; sqrt in fp64 rounded once to fp32 is exact (53 >= 2*24+2)."""
    return torch.sqrt(x.double()).to(x.dtype)


# ----------------------------------------------------------------------------------------
# RGBDImages maps (gradslam.structures.rgbdimages, SURVEY Appendix A "RGBDImages")
# ----------------------------------------------------------------------------------------
def intrinsics_inverse(K):
    """Closed-form inverse of a pinhole K (4,4): [1/fx, 0, -cx/fx; 0, 1/fy, -cy/fy; 0 0 1]."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    Ki = torch.zeros(3, 3, dtype=K.dtype)
    Ki[0, 0] = 1.0 / fx
    Ki[1, 1] = 1.0 / fy
    Ki[0, 2] = -cx / fx
    Ki[1, 2] = -cy / fy
    Ki[2, 2] = 1.0
    return Ki


def vertex_normal_maps(depth, K, pose):
    """depth (H,W) fp32, K (4,4), pose (4,4) camera-to-world -> dict of (H,W,3) maps:
       V  = (Kinv . [w,h,1]) * d * valid              local vertex map
       n  = normalize(cross(dW V, dH V)) * valid      local normal map (last col/row diff = 0)
       Vg = (R V + t) * valid ;  ng = R n             global maps
    Explicit order:  rx = Ki00*w + Ki02 ; ry = Ki11*h + Ki12 ; V = (rx*d, ry*d, d)
                     Vg_i = ((R_i0*Vx + R_i1*Vy) + R_i2*Vz) + t_i
    """
    H, W = depth.shape
    Ki = intrinsics_inverse(K)
    hs, ws = torch.meshgrid(torch.arange(H, dtype=torch.float32),
                            torch.arange(W, dtype=torch.float32), indexing="ij")
    valid = depth != 0
    vf = valid.to(depth.dtype)
    rx = Ki[0, 0] * ws + Ki[0, 2]
    ry = Ki[1, 1] * hs + Ki[1, 2]
    V = torch.stack([rx * depth, ry * depth, depth], -1) * vf.unsqueeze(-1)
    dh = torch.zeros_like(V)
    dv = torch.zeros_like(V)
    dh[:, :-1] = V[:, 1:] - V[:, :-1]
    dv[:-1, :] = V[1:, :] - V[:-1, :]
    cx = dh[..., 1] * dv[..., 2] - dh[..., 2] * dv[..., 1]
    cy = dh[..., 2] * dv[..., 0] - dh[..., 0] * dv[..., 2]
    cz = dh[..., 0] * dv[..., 1] - dh[..., 1] * dv[..., 0]
    nrm = sqrt_rn((cx * cx + cy * cy) + cz * cz)
    den = torch.where(nrm == 0, torch.ones_like(nrm), nrm)
    n = torch.stack([cx / den, cy / den, cz / den], -1) * vf.unsqueeze(-1)

    def rot(R, v):
        return torch.stack([(R[i, 0] * v[..., 0] + R[i, 1] * v[..., 1]) + R[i, 2] * v[..., 2] for i in range(3)], -1)

    R, t = pose[:3, :3], pose[:3, 3]
    Vg = (rot(R, V) + t) * vf.unsqueeze(-1)
    ng = rot(R, n)
    return {"valid": valid, "V": V, "n": n, "Vg": Vg, "ng": ng}


def rigid_inverse(T):
    """[R^T | -R^T t] with explicit order (gradslam inverse_transformation)."""
    R, t = T[:3, :3], T[:3, 3]
    Ti = torch.eye(4, dtype=T.dtype)
    Rt = R.t()
    Ti[:3, :3] = Rt
    for i in range(3):
        Ti[i, 3] = -((Rt[i, 0] * t[0] + Rt[i, 1] * t[1]) + Rt[i, 2] * t[2])
    return Ti


# ----------------------------------------------------------------------------------------
# fusionutils (SURVEY Appendix A "fusionutils")
# ----------------------------------------------------------------------------------------
def find_active_map_points(points, K, pose, H, W):
    """Map points (M,3) -> (P,3) int64 rows [n, h, w] of the points that project inside the live
    frame, ascending n.  p' = T^-1 p ; front = z'>0 ; u = (fx*x' + cx*z')/z' ; v = (fy*y' + cy*z')/z' ;
    in = u>-1e-3 & u<W-0.999 & v>-1e-3 & v<H-0.999 & front ; (h,w) = round-half-even, clamped."""
    M = points.shape[0]
    if M == 0:
        return torch.zeros(0, 3, dtype=torch.int64)
    Ti = rigid_inverse(pose)
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    xc = ((Ti[0, 0] * x + Ti[0, 1] * y) + Ti[0, 2] * z) + Ti[0, 3]
    yc = ((Ti[1, 0] * x + Ti[1, 1] * y) + Ti[1, 2] * z) + Ti[1, 3]
    zc = ((Ti[2, 0] * x + Ti[2, 1] * y) + Ti[2, 2] * z) + Ti[2, 3]
    front = zc > 0
    u = (K[0, 0] * xc + K[0, 2] * zc) / zc
    v = (K[1, 1] * yc + K[1, 2] * zc) / zc
    inside = (u > -1e-3) & (u < W - 0.999) & (v > -1e-3) & (v < H - 0.999) & front
    n = torch.nonzero(inside).squeeze(1)
    h = torch.round(v[n]).long().clamp(0, H - 1)
    w = torch.round(u[n]).long().clamp(0, W - 1)
    return torch.stack([n, h, w], 1)


def find_similar_map_points(points, normals, maps, pc2im, dist_th, dot_th):
    """Keep rows whose map point is within dist_th of the frame's global vertex at (h,w) and whose
    normals agree: ||Vg-p|| < dist_th  and  ng.nm > dot_th."""
    if pc2im.shape[0] == 0:
        return pc2im, torch.zeros(0, dtype=torch.bool)
    n, h, w = pc2im[:, 0], pc2im[:, 1], pc2im[:, 2]
    fp, fn = maps["Vg"][h, w], maps["ng"][h, w]
    mp, mn = points[n], normals[n]
    d = fp - mp
    dist = sqrt_rn((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    dot = (fn[:, 0] * mn[:, 0] + fn[:, 1] * mn[:, 1]) + fn[:, 2] * mn[:, 2]
    keep = (dist < dist_th) & (dot > dot_th)
    return pc2im[keep], keep


def find_best_unique_correspondences(points, ccounts, maps, pc2im):
    """Per pixel keep ONE map point: max confidence (min 1/(c+1e-20)), then min squared distance,
    then min index.  Output rows [n,h,w] sorted by (h,w)."""
    if pc2im.shape[0] == 0:
        return pc2im
    n, h, w = pc2im[:, 0], pc2im[:, 1], pc2im[:, 2]
    inv_c = 1.0 / (ccounts[n] + 1e-20)
    d = points[n] - maps["Vg"][h, w]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    W = maps["Vg"].shape[1]
    pix = h * W + w
    # lexicographic minimum of (1/c, d2, n) per pixel == first row of each pixel's run after a row-lexicographic sort by
    # (pixel, 1/c, d2, n) (gradslam: torch.unique(rows, dim=0)); built from stable sorts, least significant key first
    # (rows arrive in ascending n, so n is already in order)
    order = torch.arange(pix.shape[0])
    for key in (d2, inv_c, pix):
        order = order[torch.sort(key[order], stable=True).indices]
    ps = pix[order]
    first = torch.ones_like(ps, dtype=torch.bool)
    first[1:] = ps[1:] != ps[:-1]
    return pc2im[order[first]]


def fusion_alpha(V, sigma, eps=1e-7):
    """alpha = exp(-|V_local|^2 / (2 sigma^2 + eps)) per pixel (gradslam get_alpha)."""
    den = torch.tensor(2 * (sigma ** 2) + eps, dtype=torch.float32)
    s = (V[..., 0] * V[..., 0] + V[..., 1] * V[..., 1]) + V[..., 2] * V[..., 2]
    return torch.exp(-s / den)


def fuse_with_map(state, maps, colors, pc2im, sigma):
    """state: dict points/normals/colors (M,3), ccounts (M,).  Returns the new state.
    Matched rows: c' = c + a ; X' = (c*X + a*X_f) / where(c'==0,1,c').  As in the padded-tensor
    formulation every map point goes through that expression (a = 0, X_f = 0 when unmatched).
    New points = valid-depth pixels with no correspondence, row-major (h,w) order, ccount = alpha."""
    H, W = maps["valid"].shape
    alpha = fusion_alpha(maps["V"], sigma)
    M = state["points"].shape[0]
    new_mask = maps["valid"].clone()
    out = {}
    if M > 0 and pc2im.shape[0] > 0:
        n, h, w = pc2im[:, 0], pc2im[:, 1], pc2im[:, 2]
        fa = torch.zeros(M)
        fa[n] = alpha[h, w]
        c = state["ccounts"]
        cn = c + fa
        den = torch.where(cn == 0, torch.ones_like(cn), cn)
        for name, src in (("points", maps["Vg"]), ("normals", maps["ng"]), ("colors", colors)):
            f = torch.zeros(M, 3)
            f[n] = src[h, w]
            out[name] = (c.unsqueeze(1) * state[name] + fa.unsqueeze(1) * f) / den.unsqueeze(1)
        out["ccounts"] = cn
        new_mask[h, w] = False
    else:
        out = {k: v.clone() for k, v in state.items()}
    out["points"] = torch.cat([out["points"], maps["Vg"][new_mask]], 0)
    out["normals"] = torch.cat([out["normals"], maps["ng"][new_mask]], 0)
    out["colors"] = torch.cat([out["colors"], colors[new_mask]], 0)
    out["ccounts"] = torch.cat([out["ccounts"], alpha[new_mask]], 0)
    return out


def empty_state():
    z = torch.zeros(0, 3)
    return {"points": z.clone(), "normals": z.clone(), "colors": z.clone(), "ccounts": torch.zeros(0)}


def pointfusion_step(state, colors, depth, K, pose, dist_th=0.05, angle_th=20.0, sigma=0.6):
    """PointFusion.step with a known pose (odom='gt' or prev_frame None) = update_map_fusion.
    colors (H,W,3), depth (H,W).  Returns (new_state, tables) where tables holds the three
    index tables for bit-exact comparison."""
    H, W = depth.shape
    dot_th = torch.tensor(math.cos(angle_th * math.pi / 180.0), dtype=torch.float32)
    dth = torch.tensor(dist_th, dtype=torch.float32)
    maps = vertex_normal_maps(depth, K, pose)
    active = find_active_map_points(state["points"], K, pose, H, W)
    similar, _ = find_similar_map_points(state["points"], state["normals"], maps, active, dth, dot_th)
    unique = find_best_unique_correspondences(state["points"], state["ccounts"], maps, similar)
    new_state = fuse_with_map(state, maps, colors, unique, sigma)
    return new_state, {"active": active, "similar": similar, "unique": unique, "maps": maps}


def transform_pointcloud(points, T):
    """(N,3) -> (R p^T + t)^T, explicit order.  gradslam.geometry.geometryutils.transform_pointcloud."""
    R, t = T[:3, :3], T[:3, 3]
    return torch.stack([((R[i, 0] * points[:, 0] + R[i, 1] * points[:, 1]) + R[i, 2] * points[:, 2]) + t[i]
                        for i in range(3)], 1)
