"""HIP unprojection / PointFusion / KNN kernels against the CPU oracle: index tables and masks bit-exact,
float payloads bit-exact where the arithmetic is IEEE (+,-,*,/,sqrt) and 1e-6 where exp() is involved."""
import math

import pytest
import torch

from oracle import knn as oknn
from oracle import pointfusion as opf

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _K(H, W):
    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 481.2 * W / 640, -480.0 * H / 480, 319.5 * W / 640, 239.5 * H / 480
    return K


def _pose(rx=0.0, ry=0.0, rz=0.0, t=(0.0, 0.0, 0.0)):
    a, b, c = (math.radians(v) for v in (rx, ry, rz))
    Rx = torch.tensor([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]], dtype=torch.float32)
    Ry = torch.tensor([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]], dtype=torch.float32)
    Rz = torch.tensor([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]], dtype=torch.float32)
    T = torch.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = torch.tensor(t)
    return T


def _scene(H, W, seed, holes=0.02):
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    depth = 2.0 + 0.4 * torch.sin(xs / W * 5.0) * torch.cos(ys / H * 4.0) + 0.01 * torch.rand(H, W, generator=g)
    depth[torch.rand(H, W, generator=g) < holes] = 0.0
    rgb = torch.rand(H, W, 3, generator=g)
    return depth, rgb


def test_vertex_normal_maps_bitexact_and_grad():
    from e2ehip import ops
    H, W = 45, 70
    depth, _ = _scene(H, W, 1)
    K, pose = _K(H, W), _pose(3, -4, 5, (0.3, -0.2, 0.1))
    ref = opf.vertex_normal_maps(depth, K, pose)
    out = ops.vertex_normal_maps(depth.to(DEV)[None], K.to(DEV)[None], pose.to(DEV)[None], sigma=0.6)
    for a, b in (("V", "V"), ("n", "n"), ("Vg", "Vg"), ("ng", "ng")):
        assert torch.equal(out[a][0].cpu(), ref[b]), f"{a} not bit-exact"
    assert torch.equal(out["valid"][0].cpu(), ref["valid"])
    torch.testing.assert_close(out["alpha"][0].cpu(), opf.fusion_alpha(ref["V"], 0.6), rtol=1e-6, atol=0)
    # backward through V and Vg
    g = torch.Generator().manual_seed(2)
    w1, w2 = torch.rand(H, W, 3, generator=g), torch.rand(H, W, 3, generator=g)
    dc = depth.clone().requires_grad_(True)
    r2 = opf.vertex_normal_maps(dc, K, pose)
    ((r2["V"] * w1).sum() + (r2["Vg"] * w2).sum()).backward()
    dg = depth.to(DEV)[None].requires_grad_(True)
    o2 = ops.vertex_normal_maps(dg, K.to(DEV)[None], pose.to(DEV)[None])
    ((o2["V"][0] * w1.to(DEV)).sum() + (o2["Vg"][0] * w2.to(DEV)).sum()).backward()
    torch.testing.assert_close(dg.grad[0].cpu(), dc.grad, rtol=1e-5, atol=1e-6)


def test_transform_points_bitexact_and_grad():
    from e2ehip import ops
    p = torch.rand(1000, 3) * 4 - 2
    T = _pose(10, 20, -30, (0.5, 1.0, -2.0))
    assert torch.equal(ops.transform_points(p.to(DEV), T.to(DEV)).cpu(), opf.transform_pointcloud(p, T))
    pg = p.to(DEV).requires_grad_(True)
    w = torch.rand(1000, 3)
    (ops.transform_points(pg, T.to(DEV)) * w.to(DEV)).sum().backward()
    torch.testing.assert_close(pg.grad.cpu(), w @ T[:3, :3], rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        ops.transform_points(p.to(DEV)[:, :2], T.to(DEV))


def _gpu_map(state, H, W, cap):
    from e2ehip.fusionmap import FusionMap
    m = FusionMap(cap, H, W, DEV)
    m.load_state(state["points"].to(DEV), state["normals"].to(DEV), state["colors"].to(DEV), state["ccounts"].to(DEV))
    return m


def _assert_state(m, st, exact_geometry):
    P, Nn, C, cc = (t.cpu() for t in m.live())
    assert P.shape[0] == st["points"].shape[0], "map size differs"
    if exact_geometry:
        assert torch.equal(P, st["points"]) and torch.equal(Nn, st["normals"]) and torch.equal(C, st["colors"])
    else:
        torch.testing.assert_close(P, st["points"], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(Nn, st["normals"], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(C, st["colors"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(cc, st["ccounts"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("H,W", [(24, 32), (60, 80), (120, 160), (480, 640)])
def test_pointfusion_step_tables_bitexact(H, W):
    """frame 0 into an empty map, then frame 1 (moved camera) against it: all three index tables bit-exact."""
    K = _K(H, W)
    d0, c0 = _scene(H, W, 10)
    d1, c1 = _scene(H, W, 10)                       # same surface seen from a moved camera (depth map reused)
    p0, p1 = _pose(), _pose(0.5, 1.0, 0.3, (0.02, 0.01, -0.015))
    st0, _ = opf.pointfusion_step(opf.empty_state(), c0, d0, K, p0)
    m = _gpu_map(opf.empty_state(), H, W, 3 * H * W)
    m.step(c0.to(DEV), d0.to(DEV), K.to(DEV), p0.to(DEV))
    assert m.table("active").shape[0] == 0
    _assert_state(m, st0, exact_geometry=True)        # first frame: pure append, no exp() in positions
    # second frame from the SAME map state on both sides
    st1, tab = opf.pointfusion_step(st0, c1, d1, K, p1)
    m = _gpu_map(st0, H, W, 3 * H * W)
    maps = m.step(c1.to(DEV), d1.to(DEV), K.to(DEV), p1.to(DEV))
    for name in ("active", "similar", "unique"):
        got = m.table(name).cpu()
        assert got.shape == tab[name].shape, f"{name}: {got.shape[0]} rows vs {tab[name].shape[0]}"
        assert torch.equal(got, tab[name]), f"{name} table differs"
    assert tab["unique"].shape[0] > (0.3 if H < 480 else 0.05) * H * W     # the case is not trivial (the fixed 2 cm motion is many pixels at 480x640)
    assert torch.equal(maps["Vg"][0].cpu(), tab["maps"]["Vg"])
    _assert_state(m, st1, exact_geometry=False)


def test_pointfusion_three_frame_chain():
    """Each side evolves its own map over 3 frames (exp() ulps may enter the confidences): sizes and tables stay equal."""
    H, W = 48, 64
    K = _K(H, W)
    st = opf.empty_state()
    m = _gpu_map(st, H, W, 4 * H * W)
    for f in range(3):
        d, c = _scene(H, W, 20)
        pose = _pose(0.3 * f, 0.5 * f, 0.0, (0.01 * f, 0.0, -0.01 * f))
        st, tab = opf.pointfusion_step(st, c, d, K, pose)
        m.step(c.to(DEV), d.to(DEV), K.to(DEV), pose.to(DEV))
        assert torch.equal(m.table("unique").cpu(), tab["unique"])
        _assert_state(m, st, exact_geometry=False)


def test_pointfusion_edge_cases():
    from e2ehip.fusionmap import FusionMap
    H, W = 16, 24
    K = _K(H, W)
    d, c = _scene(H, W, 3)
    m = FusionMap(2 * H * W, H, W, DEV)
    m.step(c.to(DEV), torch.zeros(H, W, device=DEV), K.to(DEV), _pose().to(DEV))      # all-invalid depth: nothing appended
    assert m.M == 0
    m.step(c.to(DEV), d.to(DEV), K.to(DEV), _pose().to(DEV))
    n1 = m.M
    assert n1 == int((d != 0).sum())
    # camera looking away: no active points, the whole frame is appended, the old points are untouched (bitwise)
    before = m.points[:n1].clone()
    m.step(c.to(DEV), d.to(DEV), K.to(DEV), _pose(0, 180, 0).to(DEV))
    assert m.table("active").shape[0] == 0 and m.M == 2 * n1 and torch.equal(m.points[:n1], before)
    small = FusionMap(n1 // 2, H, W, DEV)
    with pytest.raises(RuntimeError):
        small.step(c.to(DEV), d.to(DEV), K.to(DEV), _pose().to(DEV))


@pytest.mark.parametrize("algo", ["brute", "grid"])
@pytest.mark.parametrize("n1,n2", [(1, 1), (777, 5000), (4096, 30000), (300, 3)])
def test_knn1_bitexact(n1, n2, algo):
    from e2ehip import ops
    g = torch.Generator().manual_seed(n1 + n2)
    a, b = torch.rand(n1, 3, generator=g), torch.rand(n2, 3, generator=g)
    if n2 > 20:
        b[17] = b[5]; b[n2 - 1] = b[5]           # duplicates: the smallest index must win
        a[0] = b[17]
    d_ref, i_ref = oknn.knn1(a, b)
    ag = a.to(DEV).requires_grad_(True)
    d, i = ops.knn1(ag, b.to(DEV), algo)
    assert torch.equal(d.detach().cpu(), d_ref) and torch.equal(i.cpu(), i_ref)
    w = torch.rand(n1, generator=g)
    (d * w.to(DEV)).sum().backward()
    ref_g = 2 * w[:, None] * (a - b[i_ref])
    torch.testing.assert_close(ag.grad.cpu(), ref_g, rtol=1e-6, atol=1e-7)


def test_knn_full_frame_properties():
    """BASELINE size (307 200 queries): checked through size-independent properties instead of the slow oracle."""
    from e2ehip import ops
    g = torch.Generator().manual_seed(0)
    q = torch.rand(307200, 3, generator=g).to(DEV)
    ref = torch.rand(50000, 3, generator=g).to(DEV)
    d, i = ops.knn1(q, ref)
    # (a) the reported distance is the distance to the reported point, bitwise
    dd = q - ref[i]
    assert torch.equal(d, (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2])
    # (b) no sampled reference point is closer (spot-check 64 random references for every query)
    pick = torch.randint(0, ref.shape[0], (64,), generator=g).to(DEV)
    dp = ((q[:, None, :] - ref[pick][None]) ** 2).sum(-1).min(1)[0]
    assert bool((d <= dp * (1 + 1e-6)).all())
    # (c) querying the reference set against itself finds every point at distance 0 with its own (first) index
    d0, i0 = ops.knn1(ref, ref)
    assert float(d0.max()) == 0.0 and torch.equal(i0, torch.arange(ref.shape[0], device=DEV))
    # (d) a subset agrees with the CPU oracle exactly
    dr, ir = oknn.knn1(q[:2000].cpu(), ref.cpu())
    assert torch.equal(d[:2000].cpu(), dr) and torch.equal(i[:2000].cpu(), ir)


@pytest.mark.parametrize("case", ["surface", "clustered", "far_queries", "degenerate", "big_map", "sparse_far", "large_coordinates", "dense_5m_map"])
def test_knn_grid_equals_brute(case):
    """The grid search must return exactly what the brute force returns (distances AND indices, ties included) on
    surface-like data, heavy clusters with duplicates, queries far outside the reference set, and a degenerate cloud."""
    from e2ehip import ops
    g = torch.Generator().manual_seed(hash(case) % 1000)
    if case == "surface":          # two overlapping depth-map-like sheets, queries slightly off the surface
        u = torch.rand(120000, 2, generator=g) * 4 - 2
        ref = torch.stack([u[:, 0], u[:, 1], 2.2 + 0.3 * torch.sin(u[:, 0] * 2) + 0.002 * torch.randn(120000, generator=g)], 1)
        v = torch.rand(60000, 2, generator=g) * 4.4 - 2.2
        q = torch.stack([v[:, 0] + 0.06, v[:, 1], 2.2 + 0.3 * torch.sin(v[:, 0] * 2)], 1)
    elif case == "clustered":
        c = torch.randn(50, 3, generator=g)
        ref = (c[torch.randint(0, 50, (60000,), generator=g)] + 0.01 * torch.randn(60000, 3, generator=g))
        ref[1000:1400] = ref[7]                                  # 400 exact duplicates
        q = torch.cat([ref[torch.randint(0, 60000, (5000,), generator=g)], torch.randn(20000, 3, generator=g)], 0)
    elif case == "far_queries":
        ref = torch.rand(20000, 3, generator=g)
        q = torch.cat([torch.rand(3000, 3, generator=g) * 40 - 20, torch.rand(3000, 3, generator=g)], 0)
    elif case == "big_map":        # > 1.5 M surface points: the 256^3 grid; a third of the queries look at unmapped space
        n = 1_800_000
        u = torch.rand(n, 2, generator=g) * 8 - 4
        ref = torch.stack([u[:, 0], 1.2 * torch.cos(u[:, 1]) + 0.001 * torch.randn(n, generator=g), u[:, 1]], 1)
        ref[5000:5300] = ref[11]                                 # duplicates
        v = torch.rand(20000, 2, generator=g) * 8 - 4
        near = torch.stack([v[:, 0], 1.2 * torch.cos(v[:, 1]) + 0.01, v[:, 1]], 1)
        far = torch.rand(10000, 3, generator=g) * torch.tensor([20.0, 6.0, 20.0]) - torch.tensor([10.0, 3.0, 10.0])
        q = torch.cat([near, far, ref[11:12], ref[5100:5101]], 0)
    elif case == "dense_5m_map":   # a late-sequence map: 5.2 M points = 13 overlapping passes over the same 5 x 3 m surface (hundreds of points per
        n, layers = 400_000, 13    # 2 cm cell), 60 % of the queries a few millimetres off the surface, the rest looking at unmapped space
        refs = []
        for _ in range(layers):
            u = torch.rand(n, 2, generator=g) * torch.tensor([5.0, 3.0]) - torch.tensor([2.5, 1.5])
            refs.append(torch.stack([u[:, 0], u[:, 1], 2.0 + 0.4 * torch.sin(u[:, 0] * 1.3) * torch.cos(u[:, 1]) + 0.002 * torch.randn(n, generator=g)], 1))
        ref = torch.cat(refs, 0)
        ref[70000:70200] = ref[5]
        v = torch.rand(12000, 2, generator=g) * torch.tensor([5.0, 3.0]) - torch.tensor([2.5, 1.5])
        near = torch.stack([v[:, 0], v[:, 1], 2.0 + 0.4 * torch.sin(v[:, 0] * 1.3) * torch.cos(v[:, 1]) + 0.004], 1)
        far = torch.rand(8000, 3, generator=g) * torch.tensor([9.0, 6.0, 5.0]) - torch.tensor([4.5, 3.0, 0.5])
        q = torch.cat([near, far, ref[5:6]], 0)
    elif case == "large_coordinates":   # a dense sheet 300 m from the origin: fp32 ulp there is 3e-5 m, several times the cell-face
        n = 200000                          # slack a fixed 1e-5 m margin would give; queries both near the sheet and 2 km away
        u = torch.rand(n, 2, generator=g) * 6
        ref = torch.stack([300.0 + u[:, 0], -200.0 + 0.3 * torch.sin(u[:, 0]) + 0.001 * torch.randn(n, generator=g), 150.0 + u[:, 1]], 1)
        v = torch.rand(30000, 2, generator=g) * 6
        near = torch.stack([300.0 + v[:, 0], -200.0 + 0.3 * torch.sin(v[:, 0]) + 0.004, 150.0 + v[:, 1]], 1)
        far = torch.rand(2000, 3, generator=g) * 4000 - 2000
        q = torch.cat([near, far], 0)
    elif case == "sparse_far":     # tiny clusters far apart: the finishing pass has to double its radius through empty space
        ref = torch.cat([0.01 * torch.randn(6000, 3, generator=g), 0.01 * torch.randn(6000, 3, generator=g) + 30.0], 0)
        q = torch.cat([torch.rand(4000, 3, generator=g) * 30, 0.01 * torch.randn(500, 3, generator=g) + 30.0], 0)
    else:
        ref = torch.zeros(9000, 3) + 1.5                          # zero extent
        q = torch.rand(2000, 3, generator=g)
    db, ib = ops.knn1(q.to(DEV), ref.to(DEV), "brute")
    dg, ig = ops.knn1(q.to(DEV), ref.to(DEV), "grid")
    assert torch.equal(db, dg) and torch.equal(ib, ig)


def test_knn_index_equals_one_shot_and_is_reusable():
    """A prebuilt KnnIndex answers several query sets exactly like knn1(p1, ref) (brute force), gradients included."""
    from e2ehip import ops
    g = torch.Generator().manual_seed(77)
    ref = torch.rand(70000, 3, generator=g)
    ref[100:140] = ref[3]
    index = ops.KnnIndex(ref.to(DEV), 9000)
    for k, n1 in enumerate((9000, 1, 4096)):
        q = (torch.rand(n1, 3, generator=g) * (1.0 + k)).to(DEV).requires_grad_(True)       # later sets reach outside the cloud
        d, i = ops.knn1(q, index)
        qb = q.detach().clone().requires_grad_(True)
        db, ib = ops.knn1(qb, ref.to(DEV), "brute")
        assert torch.equal(d, db) and torch.equal(i, ib)
        d.sum().backward()
        db.sum().backward()
        assert torch.equal(q.grad, qb.grad)
    with pytest.raises(ValueError):
        ops.knn1(torch.rand(9001, 3, device=DEV), index)


def test_resident_map_step_and_index_equal_the_host_visible_forms():
    """FusionMap.step_resident (live size on the device: e2e_pf_associate_dev / e2e_pf_fuse_append_dev, preallocated frame maps, no host
    read) against FusionMap.step (host-visible size, itself checked against the oracle above): identical maps bit for bit over a 4-frame
    chain; the resident nearest-neighbour index (one capacity-sized buffer rebuilt in place from the device-resident count,
    e2e_knn1_index_build_dev) against the brute force after every frame; the sticky overflow flag."""
    from e2ehip import _lib as L, ops
    from e2ehip.fusionmap import FusionMap
    H, W = 48, 64
    K = _K(H, W).to(DEV)
    a, b = FusionMap(6 * H * W, H, W, DEV), FusionMap(6 * H * W, H, W, DEV)
    g = torch.Generator().manual_seed(9)
    for f in range(4):
        d, c = _scene(H, W, 30 + (f % 2))
        pose = _pose(0.4 * f, 0.7 * f, 0.1 * f, (0.02 * f, 0.0, -0.01 * f)).to(DEV)
        a.step(c.to(DEV), d.to(DEV), K, pose)
        b.step_resident(c.to(DEV).contiguous(), d.to(DEV).contiguous(), K, pose)
        assert b._M is None                              # nothing was read back
        q = (torch.rand(3000, 3, generator=g) * 4 - 2).to(DEV)
        index = b.knn_index(4096)                        # built from count[0] on the device
        assert b._M is None
        di, ii = ops.knn1(q, index)
        assert b.M == a.M and b.M > 0
        for x, y in zip(a.live(), b.live()):
            assert torch.equal(x, y)
        db, ib = ops.knn1(q, b.points[: b.M].contiguous(), "brute")
        assert torch.equal(di, db) and torch.equal(ii, ib)
    assert b.knn_index(4096) is index                    # one buffer for the run
    # image-ordered queries (e2e_knn1_index_query_dev_image: the lanes of a wave take 8 x 8 pixel tiles): same results as the plain order
    # -- a back-projected depth image of the map's own size, and a size that is not a multiple of 8 (falls back to the plain order)
    big = b.knn_index(H * W)
    for hh, ww in ((H, W), (H - 3, W)):
        qi = (torch.rand(hh * ww, 3, generator=g) * 4 - 2).to(DEV)
        d0, i0 = torch.empty(hh * ww, device=DEV), torch.empty(hh * ww, dtype=torch.int64, device=DEV)
        d1, i1 = torch.empty_like(d0), torch.empty_like(i0)
        big.query(qi, hh * ww, d0, i0, L.stream())
        big.query(qi, hh * ww, d1, i1, L.stream(), row_len=ww)
        assert torch.equal(d0, d1) and torch.equal(i0, i1)
        db, ib = ops.knn1(qi, b.points[: b.M].contiguous(), "brute")
        assert torch.equal(d1, db.reshape(-1)) and torch.equal(i1, ib.reshape(-1))
        # warm starts (e2e_knn1_index_query_dev_image_warm): whatever the candidates -- the true neighbours of slightly moved queries (the
        # use case: consecutive refinement steps), random valid rows, or rows outside the live map -- the result is the exact one
        moved = qi + 0.003 * torch.randn(qi.shape, generator=g).to(DEV)
        dm, im = ops.knn1(moved, b.points[: b.M].contiguous(), "brute")
        for cand in (i1.clone(), torch.randint(0, int(b.M), (hh * ww,), generator=g).to(DEV), torch.full((hh * ww,), int(b.M) + 5, device=DEV),
                     torch.full((hh * ww,), -1, dtype=torch.int64, device=DEV)):
            dw, iw = torch.empty_like(d0), cand.clone()
            big.query(moved, hh * ww, dw, iw, L.stream(), row_len=ww, warm=iw)          # in place, as the step plan does
            assert torch.equal(dw, dm.reshape(-1)) and torch.equal(iw, im.reshape(-1))
    small = FusionMap(int(a.M) // 3, H, W, DEV)
    d, c = _scene(H, W, 30)
    small.step_resident(c.to(DEV), d.to(DEV), K, _pose().to(DEV))
    with pytest.raises(RuntimeError):
        small.M


def test_chamfer_bidirectional_and_colour_loss_values_vs_oracle():
    """N3 leftovers: ChamferDistance(bidirectional=True) as train_depth.py:690-692 calls it, and
    knn_points_loss + color_points_loss as gradient_experiments.py:143-150 chains them -- VALUES and gradients against the oracle.
    The reverse term searches the differentiable cloud, so its gradient is the scatter half of knn_points' backward
    (e2e_knn1_bwd_ref: fixed-point accumulation, bitwise reproducible)."""
    from chamferdist import ChamferDistance
    from loss.losses import color_points_loss, knn_points_loss
    from oracle import knn as oknn
    g = torch.Generator().manual_seed(31)
    src = torch.randn(1, 3000, 3, generator=g) * 0.5
    tgt = torch.cat([src[:, :2200] + 0.02 * torch.randn(1, 2200, 3, generator=g), torch.randn(1, 300, 3, generator=g)], 1)   # shared neighbours
    cs, ct = torch.rand(1, 3000, 3, generator=g), torch.rand(1, 2500, 3, generator=g)
    cd = ChamferDistance()
    for kw in (dict(bidirectional=True), dict(reverse=True), dict(), dict(bidirectional=True, reduction="sum")):
        a, b = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
        ref = oknn.chamfer_distance(a, b, **kw)
        ga, gb = torch.autograd.grad(ref, [a, b], allow_unused=True)
        x, y = src.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True)
        outs = []
        for rep in range(2):
            val = cd(x, y, **kw)
            gx, gy = torch.autograd.grad(val, [x, y], allow_unused=True)
            outs.append((val.detach().clone(), gx, gy))
        torch.testing.assert_close(outs[0][0].cpu(), ref.detach(), rtol=1e-5, atol=1e-9)
        for got, want in ((outs[0][1], ga), (outs[0][2], gb)):
            if want is None:
                assert got is None or float(got.abs().max()) == 0.0
            else:
                torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-7 * float(want.abs().max()) + 1e-12)
        for p, q in zip(outs[0], outs[1]):                                  # run-to-run: bit for bit (integer accumulation)
            assert (p is None and q is None) or torch.equal(p, q)
    # knn_points_loss -> indices -> colour loss (values, indices, gradient wrt the noisy colours)
    l_ref, i_ref = oknn.knn_points_loss(tgt, src)
    l_gpu, i_gpu = knn_points_loss(gt_pointcloud=tgt.to(DEV), noisy_pointcloud=src.to(DEV))
    assert torch.equal(i_gpu.cpu(), i_ref)
    torch.testing.assert_close(l_gpu.cpu(), l_ref, rtol=1e-5, atol=0)
    c = cs.clone().requires_grad_(True)
    c_ref = oknn.color_points_loss(ct, c, i_ref)
    cg = cs.to(DEV).requires_grad_(True)
    c_gpu = color_points_loss(ct.to(DEV), cg, i_gpu)
    torch.testing.assert_close(c_gpu.detach().cpu(), c_ref.detach(), rtol=1e-6, atol=0)
    torch.testing.assert_close(torch.autograd.grad(c_gpu, cg)[0].cpu(), torch.autograd.grad(c_ref, c)[0], rtol=1e-6, atol=1e-12)
    with pytest.raises(ValueError):
        color_points_loss(ct.to(DEV)[..., :2], cg, i_gpu)


@pytest.mark.parametrize("cells", [16, 128])
def test_resolution_parameterised_index_is_exact(cells):
    """e2e_knn1_index_*_res (the odometry's index: caller-chosen cells per axis, a wave per query for small query sets, cold and
    warm-started): distances and indices equal the brute force bit for bit -- near queries, queries far outside the cloud, warm
    candidates that are right, random, out of range and negative."""
    from e2ehip import _lib as L
    from e2ehip import ops
    lib = L.load()
    g = torch.Generator().manual_seed(77 + cells)
    n2, cap, n1 = 40000, 65536, 6000
    ref = torch.zeros(cap, 3)
    ref[:n2] = torch.rand(n2, 3, generator=g) * torch.tensor([4.0, 3.0, 0.2])            # a slab-like cloud
    q = torch.cat([ref[torch.randint(0, n2, (n1 // 2,), generator=g)] + 0.01 * torch.randn(n1 // 2, 3, generator=g),
                   torch.rand(n1 // 2, 3, generator=g) * 6.0 - 1.0])                       # half near the cloud, half anywhere (some far outside)
    refd, qd = ref.to(DEV), q.to(DEV).contiguous()
    count = torch.tensor([n2, 0, 0], device=DEV, dtype=torch.int64)
    index = torch.empty(lib.e2e_knn1_index_capacity_bytes_res(n1, cap, cells), device=DEV, dtype=torch.uint8)
    L.call("e2e_knn1_index_build_dev_res", L.ptr(refd), L.ptr(count), cap, n1, L.ptr(index), cells, L.stream())
    d_ref, i_ref = ops.knn1(qd, refd[:n2].contiguous(), algorithm="brute")
    d, i = torch.empty(n1, device=DEV), torch.empty(n1, device=DEV, dtype=torch.int64)
    L.call("e2e_knn1_index_query_dev_res", L.ptr(qd), n1, None, None, cap, n1, L.ptr(index), cells, L.ptr(d), L.ptr(i), L.stream())
    assert torch.equal(i, i_ref) and torch.equal(d, d_ref)
    for name, warm in (("true", i_ref.clone()), ("random", torch.randint(0, n2, (n1,), generator=g).to(DEV)),
                       ("mixed", torch.where(torch.arange(n1, device=DEV) % 3 == 0, torch.full((n1,), -1, device=DEV), torch.full((n1,), n2 + 5, device=DEV)))):
        warm = warm.to(torch.int64).contiguous()
        d2, i2 = torch.full((n1,), -1.0, device=DEV), warm.clone()                                  # idx aliases the warm buffer, as the odometry does
        L.call("e2e_knn1_index_query_dev_res", L.ptr(qd), n1, L.ptr(refd), L.ptr(i2), cap, n1, L.ptr(index), cells, L.ptr(d2), L.ptr(i2), L.stream())
        assert torch.equal(i2, i_ref) and torch.equal(d2, d_ref), name
