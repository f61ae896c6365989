"""torch.autograd glue over the C ABI (include/e2eslam.h).  Device memory, streams and autograd
bookkeeping come from PyTorch-ROCm; every computation below is a hand-written HIP kernel."""
import torch
from torch.autograd.function import once_differentiable

from . import _lib as L

PADDING = {"zeros": 0, "border": 1}


def _padding(mode):
    if mode not in PADDING:
        raise ValueError(f"padding_mode '{mode}' is not supported by the HIP path (zeros | border)")
    return PADDING[mode]


def _mat(t, name, B):
    t = L.dev(t, name)
    if t.shape != (B, 4, 4):
        raise ValueError(f"{name}: expected shape ({B},4,4), got {tuple(t.shape)}")
    return t.contiguous()


# ---------------------------------------------------------------------------------------------
class _Backproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, inv_K):
        B, _, H, W = depth.shape
        d = L.dev(depth, "depth").contiguous()
        ik = _mat(inv_K, "inv_K", B)
        out = torch.empty(B, 4, H * W, device=d.device, dtype=torch.float32)
        L.call("e2e_backproject_fwd", L.ptr(d), L.ptr(ik), L.ptr(out), B, H, W, L.stream())
        ctx.save_for_backward(ik)
        ctx.dims = (B, H, W)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        ik, = ctx.saved_tensors
        B, H, W = ctx.dims
        g = g.contiguous()
        gd = torch.empty(B, 1, H, W, device=g.device, dtype=torch.float32)
        L.call("e2e_backproject_bwd", L.ptr(g), L.ptr(ik), L.ptr(gd), B, H, W, L.stream())
        return gd, None


def backproject(depth, inv_K):
    """BackprojectDepth.forward -- view_synthesis.py:34-40."""
    if depth.dim() != 4 or depth.shape[1] != 1:
        raise ValueError(f"depth: expected (B,1,H,W), got {tuple(depth.shape)}")
    return _Backproject.apply(depth, inv_K)


class _Project3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, K, T, H, W, geometric):
        B = points.shape[0]
        p = L.dev(points, "points").contiguous()
        K, T = _mat(K, "K", B), _mat(T, "T", B)
        grid = torch.empty(B, H, W, 2, device=p.device, dtype=torch.float32)
        valid = torch.empty(B, 1, H, W, device=p.device, dtype=torch.float32)
        z = torch.empty(B, 1, H, W, device=p.device, dtype=torch.float32) if geometric else None
        L.call("e2e_project3d_fwd", L.ptr(p), L.ptr(K), L.ptr(T), L.ptr(grid), L.ptr(valid), L.ptr(z), B, H, W, L.stream())
        ctx.save_for_backward(p, K, T)
        ctx.dims = (B, H, W)
        ctx.mark_non_differentiable(valid)
        if geometric:
            return grid, z, valid
        return grid, valid

    @staticmethod
    @once_differentiable
    def backward(ctx, g_grid, *rest):
        p, K, T = ctx.saved_tensors
        B, H, W = ctx.dims
        g_z = rest[0] if len(rest) == 2 else None
        g_grid = (g_grid if g_grid is not None else torch.zeros(B, H, W, 2, device=p.device)).contiguous()
        g_z = g_z.contiguous() if g_z is not None else None
        gp = torch.empty_like(p)
        L.call("e2e_project3d_bwd", L.ptr(p), L.ptr(K), L.ptr(T), L.ptr(g_grid), L.ptr(g_z), L.ptr(gp), B, H, W, L.stream())
        return gp, None, None, None, None, None


def project3d(points, K, T, height, width, geometric=False):
    """Project3D.forward -- view_synthesis.py:54-78."""
    if points.dim() != 3 or points.shape[1] != 4 or points.shape[2] != height * width:
        raise ValueError(f"points: expected (B,4,{height * width}), got {tuple(points.shape)}")
    return _Project3D.apply(points, K, T, height, width, bool(geometric))


class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, grid, pad, align):
        B, C, Hi, Wi = inp.shape
        _, Ho, Wo, _ = grid.shape
        inp = L.dev(inp, "input")
        grid = L.dev(grid, "grid").contiguous()
        out = torch.empty(B, C, Ho, Wo, device=inp.device, dtype=torch.float32)
        L.call("e2e_grid_sample_fwd", L.ptr(inp), L.strides4(inp), L.ptr(grid), L.ptr(out), B, C, Hi, Wi, Ho, Wo, pad, int(align), L.stream())
        ctx.save_for_backward(inp, grid)
        ctx.cfg = (pad, int(align))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        inp, grid = ctx.saved_tensors
        pad, align = ctx.cfg
        B, C, Hi, Wi = inp.shape
        _, Ho, Wo, _ = grid.shape
        g = g.contiguous()
        gg = torch.empty_like(grid)
        if ctx.needs_input_grad[0]:      # gradient wrt the sampled image: a scatter -- fixed-point integer accumulation, bitwise reproducible
            gi = torch.empty(B, C, Hi, Wi, device=g.device, dtype=torch.float32)
            fx = torch.empty(B * C * Hi * Wi + 1, device=g.device, dtype=torch.int64)      # + the poison word (non-finite / out-of-range contributions)
            L.call("e2e_grid_sample_bwd_exact", L.ptr(inp), L.strides4(inp), L.ptr(grid), L.ptr(g), L.ptr(gg), L.ptr(fx), L.ptr(gi),
                   B, C, Hi, Wi, Ho, Wo, pad, align, L.stream())
            return gi, gg, None, None
        L.call("e2e_grid_sample_bwd", L.ptr(inp), L.strides4(inp), L.ptr(grid), L.ptr(g), L.ptr(gg), None,
               B, C, Hi, Wi, Ho, Wo, pad, align, L.stream())
        return None, gg, None, None


def grid_sample(input, grid, mode="bilinear", padding_mode="zeros", align_corners=False):
    """F.grid_sample as the reference calls it (online_adaption.py:431-439,450-453)."""
    if mode != "bilinear":
        raise ValueError("only bilinear sampling is on the reference's path")
    if input.dim() != 4 or grid.dim() != 4 or grid.shape[-1] != 2 or grid.shape[0] != input.shape[0]:
        raise ValueError(f"grid_sample: bad shapes input {tuple(input.shape)} grid {tuple(grid.shape)}")
    return _GridSample.apply(input, grid, _padding(padding_mode), bool(align_corners))


class _Photometric(torch.autograd.Function):
    """returns (ssim (B,C,H,W), pmap (B,1,H,W)); either may be unused by the caller."""

    @staticmethod
    def forward(ctx, x, y, want_ssim, want_pmap):
        B, C, H, W = x.shape
        x, y = L.dev(x, "prediction"), L.dev(y, "target")
        ssim = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32) if want_ssim else None
        pmap = torch.empty(B, 1, H, W, device=x.device, dtype=torch.float32) if want_pmap else None
        L.call("e2e_photometric_fwd", L.ptr(x), L.strides4(x), L.ptr(y), L.strides4(y), L.ptr(ssim), L.ptr(pmap), B, C, H, W, L.stream())
        ctx.save_for_backward(x, y)
        return ssim, pmap

    @staticmethod
    @once_differentiable
    def backward(ctx, g_ssim, g_pmap):
        x, y = ctx.saved_tensors
        B, C, H, W = x.shape
        g_ssim = g_ssim.contiguous() if g_ssim is not None else None
        g_pmap = g_pmap.contiguous() if g_pmap is not None else None
        gx = gy = None
        if g_ssim is None and g_pmap is None:
            return None, None, None, None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
            L.call("e2e_photometric_bwd", L.ptr(x), L.strides4(x), L.ptr(y), L.strides4(y), L.ptr(g_pmap), L.ptr(g_ssim), L.ptr(gx), B, C, H, W, L.stream())
        if ctx.needs_input_grad[1]:
            gy = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
            L.call("e2e_photometric_bwd", L.ptr(y), L.strides4(y), L.ptr(x), L.strides4(x), L.ptr(g_pmap), L.ptr(g_ssim), L.ptr(gy), B, C, H, W, L.stream())
        return gx, gy, None, None


def _check_pair(x, y):
    if x.dim() != 4 or x.shape != y.shape:
        raise ValueError(f"expected two (B,C,H,W) tensors of equal shape, got {tuple(x.shape)} / {tuple(y.shape)}")


def ssim(x, y):
    """SSIM.forward -- losses.py:23-37."""
    _check_pair(x, y)
    return _Photometric.apply(x, y, True, False)[0]


def photometric(prediction, target):
    """photometric_loss -- losses.py:97-117 (0.85 mean_c SSIM + 0.15 mean_c L1)."""
    _check_pair(prediction, target)
    return _Photometric.apply(prediction, target, False, True)[1]


# ---------------------------------------------------------------------------------------------
class _WarpPhotometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth_tgt, depth_src, init_tgt, init_src, src, tgt, K, inv_K, T, pad, use_mask, reg_kind, want_pmap):
        B, _, H, W = depth_tgt.shape
        dt = L.dev(depth_tgt, "depth_tgt").contiguous()
        src, tgt = L.dev(src, "source_frame"), L.dev(tgt, "target_frame")
        K, inv_K, T = _mat(K, "K", B), _mat(inv_K, "inv_K", B), _mat(T, "T", B)
        ds = it = is_ = None
        if reg_kind:
            ds = L.dev(depth_src, "depth_src").contiguous()
            it, is_ = L.dev(init_tgt, "init_tgt").contiguous(), L.dev(init_src, "init_src").contiguous()
        dev = dt.device
        synth = torch.empty(B, 3, H, W, device=dev, dtype=torch.float32)
        valid = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32)
        pmap = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32) if want_pmap else None
        loss = torch.zeros(2, device=dev, dtype=torch.float32)
        ws = torch.empty(L.load().e2e_warp_photo_workspace_floats(B, H, W), device=dev, dtype=torch.float32)
        L.call("e2e_warp_photo_fwd", L.ptr(dt), L.ptr(src), L.strides4(src), L.ptr(tgt), L.strides4(tgt), L.ptr(K), L.ptr(inv_K),
               L.ptr(T), L.ptr(synth), L.ptr(valid), L.ptr(pmap), int(use_mask), pad, reg_kind, L.ptr(it), L.ptr(is_), L.ptr(ds),
               L.ptr(loss), L.ptr(ws), B, H, W, L.stream())
        ctx.save_for_backward(dt, ds, it, is_, src, tgt, K, inv_K, T, synth, valid)
        ctx.cfg = (pad, int(use_mask), reg_kind, B, H, W)
        ctx.mark_non_differentiable(synth, valid)
        if pmap is not None:
            ctx.mark_non_differentiable(pmap)
        return loss[0], loss[1], synth, valid, pmap

    @staticmethod
    @once_differentiable
    def backward(ctx, g0, g1, *_):
        dt, ds, it, is_, src, tgt, K, inv_K, T, synth, valid = ctx.saved_tensors
        pad, use_mask, reg_kind, B, H, W = ctx.cfg
        z = torch.zeros((), device=dt.device)
        gl = torch.stack([g0 if g0 is not None else z, g1 if g1 is not None else z]).contiguous()
        gdt = torch.empty(B, 1, H, W, device=dt.device, dtype=torch.float32)
        gds = torch.empty(B, 1, H, W, device=dt.device, dtype=torch.float32) if reg_kind else None
        L.call("e2e_warp_photo_bwd", L.ptr(dt), L.ptr(src), L.strides4(src), L.ptr(tgt), L.strides4(tgt), L.ptr(K), L.ptr(inv_K),
               L.ptr(T), L.ptr(synth), L.ptr(valid), use_mask, pad, reg_kind, L.ptr(it), L.ptr(is_), L.ptr(ds), L.ptr(gl),
               L.ptr(gdt), L.ptr(gds), B, H, W, L.stream())
        return (gdt, gds) + (None,) * 11


def warp_photometric(depth_tgt, src, tgt, K, inv_K, T, padding_mode="border", use_mask=True,
                     depth_src=None, init_tgt=None, init_src=None, reg_kind=None, want_pmap=False):
    """Fused image-space part of one refinement step (2 launches forward, 1 backward).

    depth_tgt (B,1,H,W); src/tgt (B,3,H,W) views (NHWC memory welcome); K, inv_K, T (B,4,4).
    reg_kind None | "l1" | "l2" adds mean-reg(init_tgt, depth_tgt) + mean-reg(init_src, depth_src).
    Returns dict(photometric, reg, synth, valid, pmap).
    reference: online_adaption.py:412-455, :544-564, :482-511, :612-623."""
    if depth_tgt.dim() != 4 or depth_tgt.shape[1] != 1:
        raise ValueError(f"depth_tgt: expected (B,1,H,W), got {tuple(depth_tgt.shape)}")
    B, _, H, W = depth_tgt.shape
    for n, t in (("source_frame", src), ("target_frame", tgt)):
        if tuple(t.shape) != (B, 3, H, W):
            raise ValueError(f"{n}: expected ({B},3,{H},{W}), got {tuple(t.shape)}")
    rk = {None: 0, "l1": 1, "l2": 2}.get(reg_kind, -1)
    if rk < 0:
        raise ValueError("please specify a correct norm")          # losses.py:146
    if rk:
        for n, t in (("depth_src", depth_src), ("init_tgt", init_tgt), ("init_src", init_src)):
            if t is None or tuple(t.shape) != (B, 1, H, W):
                raise ValueError(f"{n}: expected ({B},1,{H},{W})")
    p, r, synth, valid, pmap = _WarpPhotometric.apply(depth_tgt, depth_src, init_tgt, init_src, src, tgt, K, inv_K, T,
                                                      _padding(padding_mode), bool(use_mask), rk, bool(want_pmap))
    return {"photometric": p, "reg": r, "synth": synth, "valid": valid, "pmap": pmap}


# ---------------------------------------------------------------------------------------------
# RGB-D unprojection, rigid transforms, nearest neighbours
# ---------------------------------------------------------------------------------------------
def fusion_alpha_den(sigma, eps=1e-7):
    """2 sigma^2 + eps evaluated in double, rounded to fp32 when passed (gradslam get_alpha)."""
    return 2.0 * (float(sigma) ** 2) + eps


class _VertexMaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, K, pose, alpha_den):
        B, H, W = depth.shape
        d = L.dev(depth, "depth").contiguous()
        K, pose = _mat(K, "intrinsics", B), _mat(pose, "poses", B)
        f = dict(device=d.device, dtype=torch.float32)
        V, Nm, Vg, Ng = (torch.empty(B, H, W, 3, **f) for _ in range(4))
        alpha = torch.empty(B, H, W, **f)
        L.call("e2e_vertex_normal_maps", L.ptr(d), L.ptr(K), L.ptr(pose), float(alpha_den), L.ptr(V), L.ptr(Nm), L.ptr(Vg), L.ptr(Ng),
               L.ptr(alpha), B, H, W, L.stream())
        ctx.save_for_backward(d, K, pose)
        ctx.mark_non_differentiable(Nm, Ng, alpha)
        return V, Nm, Vg, Ng, alpha

    @staticmethod
    @once_differentiable
    def backward(ctx, gV, gN, gVg, gNg, ga):
        d, K, pose = ctx.saved_tensors
        B, H, W = d.shape
        if gV is None and gVg is None:
            return None, None, None, None
        gV = gV.contiguous() if gV is not None else None
        gVg = gVg.contiguous() if gVg is not None else None
        gd = torch.empty_like(d)
        L.call("e2e_vertex_maps_bwd", L.ptr(d), L.ptr(K), L.ptr(pose), L.ptr(gV), L.ptr(gVg), L.ptr(gd), B, H, W, L.stream())
        return gd, None, None, None


def vertex_normal_maps(depth, K, pose, sigma=0.6):
    """depth (B,H,W), K/pose (B,4,4) -> dict V, n, Vg, ng (B,H,W,3), alpha (B,H,W), valid (B,H,W) bool.
    gradslam RGBDImages maps; differentiable wrt depth through V and Vg."""
    if depth.dim() != 3:
        raise ValueError(f"depth: expected (B,H,W), got {tuple(depth.shape)}")
    V, Nm, Vg, Ng, alpha = _VertexMaps.apply(depth, K, pose, fusion_alpha_den(sigma))
    return {"V": V, "n": Nm, "Vg": Vg, "ng": Ng, "alpha": alpha, "valid": depth.detach() != 0}


class _TransformPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, T):
        p = L.dev(points, "points").contiguous()
        T = L.dev(T, "transform").contiguous()
        out = torch.empty_like(p)
        L.call("e2e_transform_points", L.ptr(p), L.ptr(T), L.ptr(out), p.shape[0], 0, L.stream())
        ctx.save_for_backward(T)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        T, = ctx.saved_tensors
        g = g.contiguous()
        gp = torch.empty_like(g)
        L.call("e2e_transform_points", L.ptr(g), L.ptr(T), L.ptr(gp), g.shape[0], 1, L.stream())
        return gp, None


def transform_points(points, T):
    """gradslam.geometry.geometryutils.transform_pointcloud: (N,3), (4,4) -> (N,3)."""
    if not torch.is_tensor(points) or not torch.is_tensor(T):
        raise TypeError("Expected torch.Tensor inputs")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError(f"pointcloud must have shape (N,3), got {tuple(points.shape)}")
    if tuple(T.shape) != (4, 4):
        raise ValueError(f"transform must have shape (4,4), got {tuple(T.shape)}")
    if points.shape[0] == 0:
        return points.clone()
    return _TransformPoints.apply(points, T)


KNN_ALGORITHMS = {"auto": 0, "brute": 1, "grid": 2}


class _SelectRows(torch.autograd.Function):
    """x[mask] for a (N, C) tensor and a (N,) bool mask.  Same values as boolean indexing; the backward writes the
    incoming rows back with index_copy_ (the indices are unique) instead of ATen's accumulate path, which sorts the
    index list on every call (2 rocPRIM merge passes + indexing_backward: ~180 us per refinement step at 480x640)."""

    @staticmethod
    def forward(ctx, x, mask):
        idx = torch.nonzero(mask, as_tuple=False).reshape(-1)
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return torch.zeros(ctx.shape, dtype=g.dtype, device=g.device).index_copy_(0, idx, g.contiguous()), None


def select_rows(x, mask):
    if x.dim() != 2 or mask.shape != x.shape[:1] or mask.dtype != torch.bool:
        raise ValueError("select_rows expects x (N, C) and a bool mask (N,)")
    return _SelectRows.apply(x, mask)


class _Knn1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p1, p2, algorithm):
        a, b = L.dev(p1, "p1").contiguous(), L.dev(p2, "p2").contiguous()
        n1, n2 = a.shape[0], b.shape[0]
        d = torch.empty(n1, device=a.device, dtype=torch.float32)
        idx = torch.empty(n1, device=a.device, dtype=torch.int64)
        ws = torch.empty(L.load().e2e_knn1_workspace_bytes(n1, n2), device=a.device, dtype=torch.uint8)
        L.call("e2e_knn1_fwd", L.ptr(a), n1, L.ptr(b), n2, L.ptr(d), L.ptr(idx), L.ptr(ws), algorithm, L.stream())
        ctx.save_for_backward(a, b, idx)
        ctx.mark_non_differentiable(idx)
        return d, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, gd, _gi):
        a, b, idx = ctx.saved_tensors
        gd = gd.contiguous()
        gp = gq = None
        if ctx.needs_input_grad[0]:
            gp = torch.empty_like(a)
            L.call("e2e_knn1_bwd", L.ptr(gd), L.ptr(a), L.ptr(b), L.ptr(idx), a.shape[0], L.ptr(gp), L.stream())
        if ctx.needs_input_grad[1]:          # the reference cloud is differentiable (ChamferDistance's reverse term): reproducible scatter
            gq = torch.empty_like(b)
            fx = torch.empty(3 * b.shape[0] + 1, device=b.device, dtype=torch.int64)
            L.call("e2e_knn1_bwd_ref", L.ptr(gd), L.ptr(a), L.ptr(b), L.ptr(idx), a.shape[0], b.shape[0], L.ptr(fx), L.ptr(gq), L.stream())
        return gp, gq, None


class KnnIndex:
    """Uniform-grid index over a fixed reference cloud: built once, queried many times with results identical to
    knn1(p1, ref).  The reference tensor must not change while the index is in use (FusionMap drops its index whenever
    the map is updated)."""

    def __init__(self, ref, max_queries):
        ref = L.dev(ref, "ref")
        if ref.dim() != 2 or ref.shape[1] != 3 or ref.shape[0] == 0 or ref.requires_grad:
            raise ValueError("KnnIndex: reference cloud must be a detached, non-empty (P,3) tensor")
        self.ref = ref.contiguous()
        self.n2, self.max_queries = self.ref.shape[0], int(max_queries)
        self.ws = torch.empty(L.load().e2e_knn1_workspace_bytes(self.max_queries, self.n2), device=ref.device, dtype=torch.uint8)
        L.call("e2e_knn1_index_build", L.ptr(self.ref), self.n2, self.max_queries, L.ptr(self.ws), L.stream())

    resident = False

    def query(self, p1, n1, dists, idx, stream, row_len=0, warm=None):
        L.call("e2e_knn1_index_query", L.ptr(p1), int(n1), self.n2, self.max_queries, L.ptr(self.ws), L.ptr(dists), L.ptr(idx), stream)


class _Knn1Indexed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p1, index):
        a = L.dev(p1, "p1").contiguous()
        n1 = a.shape[0]
        d = torch.empty(n1, device=a.device, dtype=torch.float32)
        idx = torch.empty(n1, device=a.device, dtype=torch.int64)
        index.query(a, n1, d, idx, L.stream())
        ctx.save_for_backward(a, index.ref, idx)
        ctx.mark_non_differentiable(idx)
        return d, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, gd, _gi):
        a, b, idx = ctx.saved_tensors
        gp = torch.empty_like(a)
        L.call("e2e_knn1_bwd", L.ptr(gd.contiguous()), L.ptr(a), L.ptr(b), L.ptr(idx), a.shape[0], L.ptr(gp), L.stream())
        return gp, None


def knn1(p1, p2, algorithm="auto"):
    """p2 may be a KnnIndex (prebuilt grid over the reference cloud).  algorithm: "auto" | "brute" | "grid" (identical results).  K=1 nearest neighbour of every row of p1 (P1,3) among p2 (P2,3): (squared dists (P1,), idx (P1,) int64).
    Differentiable wrt p1 (d/dp1 = 2 g (p1 - p2[idx])) and -- plain tensors only, not a prebuilt index -- wrt p2 (the scatter of the
    negative, as chamferdist's knn_points; the online path detaches p2, online_adaption.py:643)."""
    if isinstance(p2, KnnIndex) or getattr(p2, "resident", False):       # a prebuilt index (e2ehip.fusionmap.ResidentKnnIndex over the map)
        if p1.dim() != 2 or p1.shape[1] != 3 or p1.shape[0] == 0:
            raise ValueError(f"p1: expected non-empty (P,3), got {tuple(p1.shape)}")
        if p1.shape[0] > p2.max_queries:
            raise ValueError(f"knn1: {p1.shape[0]} queries exceed the {p2.max_queries} the index was built for")
        return _Knn1Indexed.apply(p1, p2)
    for n, t in (("p1", p1), ("p2", p2)):
        if t.dim() != 2 or t.shape[1] != 3:
            raise ValueError(f"{n}: expected (P,3), got {tuple(t.shape)}")
    if p1.shape[0] == 0 or p2.shape[0] == 0:
        raise ValueError("knn1: empty point cloud")
    return _Knn1.apply(p1, p2, KNN_ALGORITHMS[algorithm])


# ---------------------------------------------------------------------------------------------
# median scaling chain, regulariser, metrics
# ---------------------------------------------------------------------------------------------
def median_lower(x):
    """torch.median(x) over all elements (lower median) -> 0-dim device tensor."""
    x = L.dev(x, "x").contiguous()
    out = torch.empty((), device=x.device, dtype=torch.float32)
    ws = torch.empty(L.load().e2e_median_workspace_bytes(), device=x.device, dtype=torch.uint8)
    L.call("e2e_median_lower", L.ptr(x), x.numel(), L.ptr(out), L.ptr(ws), L.stream())
    return out


class _DepthScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, median_gt, elements):
        d = L.dev(disp, "disp").contiguous()
        mg = L.dev(median_gt, "median_gt").reshape(1).contiguous()
        ctx.elements = elements
        n = d.numel()
        delta, depth = torch.empty_like(d), torch.empty_like(d)
        md = torch.empty(1, device=d.device, dtype=torch.float32)
        ratio = torch.empty(1, device=d.device, dtype=torch.float32)
        ws = torch.empty(L.load().e2e_depth_scale_workspace_bytes(), device=d.device, dtype=torch.uint8)
        L.call("e2e_depth_scale_fwd", L.ptr(d), L.ptr(mg), L.ptr(delta), L.ptr(depth), L.ptr(md), L.ptr(ratio), L.ptr(ws), n, L.stream())
        ctx.save_for_backward(delta, mg, md, ws)
        ratio = ratio.reshape(())
        ctx.mark_non_differentiable(delta, ratio)
        return depth, delta, ratio

    @staticmethod
    @once_differentiable
    def backward(ctx, g_depth, _gd, _gr):
        delta, mg, md, ws = ctx.saved_tensors
        g = g_depth.contiguous()
        g_disp = torch.empty_like(delta)
        el = ctx.elements
        L.call("e2e_depth_scale_bwd_at", L.ptr(g), L.ptr(delta), L.ptr(mg), L.ptr(md), L.ptr(el), 0 if el is None else int(el.numel()), L.ptr(g_disp),
               L.ptr(ws), delta.numel(), L.stream())
        return g_disp, None, None


def depth_from_disp_median_scaled(disp, median_gt, median_elements=None):
    """disp (F,1,H,W) of the keyframe pair -> (depth = (median_gt / median(1/disp)) / disp, unscaled 1/disp, ratio).
    The ratio stays inside the autograd graph exactly as in online_adaption.py:292-298 (the in-place `*= ratio`): its gradient is shared
    by the elements that hold the median value, as torch.median(x) differentiates.  median_elements (device int32 indices): name those
    elements instead (parity tests: the choice among near-tied values belongs to the evaluation, e2e_depth_scale_bwd_at)."""
    if median_elements is not None and (median_elements.dtype != torch.int32 or not median_elements.is_cuda):
        raise TypeError("median_elements: a device int32 tensor of flat indices")
    return _DepthScale.apply(disp, median_gt, median_elements)


class _FixedScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, scale):
        d = L.dev(disp, "disp").contiguous()
        depth = torch.empty_like(d)
        L.call("e2e_depth_fixed_scale_fwd", L.ptr(d), float(scale), None, L.ptr(depth), d.numel(), L.stream())
        ctx.save_for_backward(d)
        ctx.scale = float(scale)
        return depth

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        gd = torch.empty_like(d)
        L.call("e2e_depth_fixed_scale_bwd", L.ptr(g.contiguous()), L.ptr(d), ctx.scale, L.ptr(gd), d.numel(), L.stream())
        return gd, None


def depth_from_disp_fixed_scale(disp, scale=1.0):
    """depth = (1 / disp) * scale (train_depth.py:331, :343-345: the development harness scales by the constant
    ABLATION.scaling_depth instead of the median ratio); scale 1.0 is the plain `1 / disp`."""
    return _FixedScale.apply(disp, scale)


class _MeanDiff(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, kind):
        a, b = L.dev(a, "initial_depth").contiguous(), L.dev(b, "refined_depth").contiguous()
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        ws = torch.empty(L.load().e2e_reduce_workspace_floats(), device=a.device, dtype=torch.float32)
        L.call("e2e_mean_diff_fwd", L.ptr(a), L.ptr(b), a.numel(), kind, L.ptr(out), L.ptr(ws), L.stream())
        ctx.save_for_backward(a, b)
        ctx.kind = kind
        return out.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        gb = torch.empty_like(b)
        L.call("e2e_mean_diff_bwd", L.ptr(a), L.ptr(b), L.ptr(g.reshape(1).contiguous()), a.numel(), ctx.kind, L.ptr(gb), L.stream())
        return None, gb, None


def mean_diff(initial, refined, kind):
    """mean |initial - refined| ("l1") or mean (initial - refined)^2 ("l2"); gradient flows to `refined` only
    (the initial depth is a detached clone: online_adaption.py:284-285)."""
    k = {"l1": 1, "l2": 2}.get(kind)
    if k is None:
        raise ValueError("please specify a correct norm")
    if initial.shape != refined.shape:
        raise ValueError("shape mismatch")
    if initial.requires_grad:
        raise NotImplementedError("gradient wrt the initial depth is not on the reference path")
    return _MeanDiff.apply(initial, refined, k)


def depth_metrics(gt, pred, mask_zero_gt):
    """-> (7,) device tensor: abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 (losses.py:162-201)."""
    gt, pred = L.dev(gt, "gt").contiguous(), L.dev(pred, "pred").contiguous()
    if gt.numel() != pred.numel():
        raise ValueError("gt and pred must have the same number of elements")
    out = torch.empty(7, device=gt.device, dtype=torch.float32)
    ws = torch.empty(L.load().e2e_reduce_workspace_floats(), device=gt.device, dtype=torch.float32)
    L.call("e2e_depth_metrics", L.ptr(gt), L.ptr(pred), gt.numel(), int(bool(mask_zero_gt)), L.ptr(out), L.ptr(ws), L.stream())
    return out


# ---------------------------------------------------------------------------------------------------------------------
# off-by-default losses (csrc/aux_losses.hip): each Function computes the loss AND the gradient for a unit upstream
# gradient in its forward launch; backward only scales the stored gradient by the incoming scalar.
# ---------------------------------------------------------------------------------------------------------------------
def _aux_ws(dev):
    return torch.empty(L.load().e2e_aux_workspace_floats(), device=dev, dtype=torch.float32)


class _Smoothness(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img):
        d = L.dev(disp, "disp").contiguous()
        im = L.dev(img, "img")
        B, C, H, W = im.shape
        if tuple(d.shape) != (B, 1, H, W):
            raise ValueError(f"disp {tuple(d.shape)} does not match img {tuple(im.shape)}")
        out = torch.empty(2, device=d.device, dtype=torch.float32)
        g = torch.empty_like(d) if ctx.needs_input_grad[0] else None
        L.call("e2e_smoothness_lossgrad", L.ptr(d), L.ptr(im), L.strides4(im), B, C, H, W, L.ptr(out), L.ptr(g), L.ptr(_aux_ws(d.device)), L.stream())
        ctx.save_for_backward(g)
        return out[0] + out[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return (g * go if g is not None else None), None


def smoothness(disp, img):
    """loss/losses.py:119-132: mean_x(|dx disp| exp(-mean_c |dx img|)) + mean_y(...); gradient w.r.t. disp only."""
    return _Smoothness.apply(disp, img)


class _GeomConsistency(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wd, idp, mask):
        a, b = L.dev(wd, "warped_depth").contiguous(), L.dev(idp, "interpolated_depth").contiguous()
        if a.shape != b.shape:
            raise ValueError("warped / interpolated depth shapes differ")
        m = L.dev(mask, "valid_mask").expand_as(a).to(torch.float32).contiguous()
        stats = torch.empty(3, device=a.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        ga = torch.empty_like(a) if need else None
        gb = torch.empty_like(a) if need else None
        L.call("e2e_geometric_consistency_lossgrad", L.ptr(a), L.ptr(b), L.ptr(m), a.numel(), L.ptr(stats), L.ptr(ga), L.ptr(gb),
               L.ptr(_aux_ws(a.device)), L.stream())
        ctx.save_for_backward(ga, gb)
        return stats[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        ga, gb = ctx.saved_tensors
        return (ga * go if ga is not None else None), (gb * go if gb is not None else None), None


def geometric_consistency(warped_depth, interpolated_depth, valid_mask):
    """loss/losses.py:84-95; the `mask.sum() > 10000` gate is evaluated on the device (no host sync)."""
    return _GeomConsistency.apply(warped_depth, interpolated_depth, valid_mask)


class _MaskedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mask):
        p = L.dev(pred, "prediction").squeeze().contiguous()
        g = L.dev(gt, "sparse_groundtruth").squeeze().to(torch.float32).contiguous()
        m = L.dev(mask, "sparse_mask").squeeze().to(torch.float32).contiguous()
        if not (p.shape == g.shape == m.shape):
            raise ValueError("prediction / ground truth / mask shapes differ after squeeze()")
        out = torch.empty(1, device=p.device, dtype=torch.float32)
        gp = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        L.call("e2e_masked_l1_lossgrad", L.ptr(p), L.ptr(g), L.ptr(m), p.numel(), L.ptr(out), L.ptr(gp), L.ptr(_aux_ws(p.device)), L.stream())
        ctx.save_for_backward(gp)
        ctx.shape = pred.shape
        return out.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        (gp,) = ctx.saved_tensors
        return ((gp * go).reshape(ctx.shape) if gp is not None else None), None, None


def masked_l1(prediction, sparse_groundtruth, sparse_mask):
    """loss/losses.py:151-160 depth_gt_loss."""
    return _MaskedL1.apply(prediction, sparse_groundtruth, sparse_mask)


class _MinReprojection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, errors):
        e = L.dev(errors, "errors").contiguous()
        B, C, H, W = e.shape
        out = torch.empty(1, device=e.device, dtype=torch.float32)
        ge = torch.empty_like(e) if ctx.needs_input_grad[0] else None
        L.call("e2e_min_reprojection_lossgrad", L.ptr(e), B, C, H, W, L.ptr(out), L.ptr(ge), L.ptr(_aux_ws(e.device)), L.stream())
        ctx.save_for_backward(ge)
        return out.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        (ge,) = ctx.saved_tensors
        return ge * go if ge is not None else None


def min_reprojection(errors):
    """train_depth.py:657-661: torch.min over the stacked per-source (and identity) error maps, then the mean."""
    if errors.dim() != 4:
        raise ValueError("errors must be (B, C, H, W)")
    return _MinReprojection.apply(errors)


class _DispBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pair):
        d = L.dev(pair, "disp pair").contiguous()
        if d.dim() != 4 or d.shape[0] != 2 or d.shape[1] != 1:
            raise ValueError("process_disparity expects the (2,1,H,W) disparities of [img, flip(img)]")
        H, W = d.shape[2], d.shape[3]
        out = torch.empty(1, 1, H, W, device=d.device, dtype=torch.float32)
        L.call("e2e_disp_blend_fwd", L.ptr(d), H, W, L.ptr(out), L.stream())
        ctx.hw = (H, W)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        H, W = ctx.hw
        g = go.contiguous()
        gd = torch.empty(2, 1, H, W, device=g.device, dtype=torch.float32)
        L.call("e2e_disp_blend_bwd", L.ptr(g), H, W, L.ptr(gd), L.stream())
        return gd


def process_disparity(disp_pair):
    """train_depth.py:224-237 (dual-disparity post-processing)."""
    return _DispBlend.apply(disp_pair)


# ---------------------------------------------------------------------------------------------------------------------
# helpers of train_depth.py's operator-by-operator loss assembly
# ---------------------------------------------------------------------------------------------------------------------
class _MaskMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        x = L.dev(x, "image")
        B, C, H, W = x.shape
        m = L.dev(mask, "valid_mask").reshape(B, 1, H, W).contiguous()
        out = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
        L.call("e2e_mask_mul", L.ptr(x), L.strides4(x), L.ptr(m), B, C, H, W, L.ptr(out), L.stream())
        ctx.save_for_backward(m)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        g = g.contiguous()
        B, C, H, W = g.shape
        gx = torch.empty_like(g)
        L.call("e2e_mask_mul", L.ptr(g), L.strides4(g), L.ptr(m), B, C, H, W, L.ptr(gx), L.stream())
        return gx, None


def mask_mul(x, mask):
    """x (B,C,H,W) * mask (B,1,H,W) -- `prediction * valid_mask` of train_depth.py:713-714; the mask carries no gradient."""
    return _MaskMul.apply(x, mask)


class _ChannelMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = L.dev(x, "maps").contiguous()
        B, C, H, W = x.shape
        out = torch.empty(B, 1, H, W, device=x.device, dtype=torch.float32)
        L.call("e2e_channel_mean", L.ptr(x), B, C, H, W, 0, L.ptr(out), L.stream())
        ctx.C = C
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = g.contiguous()
        B, _, H, W = g.shape
        gx = torch.empty(B, ctx.C, H, W, device=g.device, dtype=torch.float32)
        L.call("e2e_channel_mean", L.ptr(g), B, ctx.C, H, W, 1, L.ptr(gx), L.stream())
        return gx


def channel_mean(x):
    """x.mean(1, keepdim=True) of stacked photometric maps (train_depth.py:630)."""
    if x.dim() != 4:
        raise ValueError("channel_mean expects (B,C,H,W)")
    return x if x.shape[1] == 1 else _ChannelMean.apply(x)


class _MeanNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d):
        d = L.dev(d, "disp").contiguous()
        B, C, H, W = d.shape
        out = torch.empty_like(d)
        ws = torch.empty(B * C * 130, device=d.device, dtype=torch.float32)
        L.call("e2e_mean_normalize", L.ptr(d), None, B * C, H, W, L.ptr(out), L.ptr(ws), L.stream())
        ctx.save_for_backward(d)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        B, C, H, W = d.shape
        gd = torch.empty_like(d)
        ws = torch.empty(B * C * 130, device=d.device, dtype=torch.float32)
        L.call("e2e_mean_normalize", L.ptr(d), L.ptr(g.contiguous()), B * C, H, W, L.ptr(gd), L.ptr(ws), L.stream())
        return gd


def mean_normalize(disp):
    """disp / (disp.mean(2, True).mean(3, True) + 1e-7) (train_depth.py:768-770)."""
    if disp.dim() != 4:
        raise ValueError("mean_normalize expects (B,C,H,W)")
    return _MeanNormalize.apply(disp)
