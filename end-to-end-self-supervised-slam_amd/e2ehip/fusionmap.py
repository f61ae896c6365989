"""Resident PointFusion map for ONE sequence (one process / GPU owns one): capacity-sized arrays that
stay in HBM for the whole run and are appended in place.  60 frames x 307 200 px x 40 B = 737 MB, a
rounding error on 288 GB, so the default capacity simply covers the whole sequence.

The live size M is DEVICE data (`count[0]`): the driver's map update (`step_resident`) and the nearest-neighbour index over the
map (`knn_index`) take it from there, so a keyframe costs no host synchronisation (gradslam reads the length of its point
lists on the host after every fusion step; round 2 of this build did the same with one `.item()` per keyframe, which idled the
GPU for ~1.5 ms per keyframe).  `M` on the host is a lazily synchronised mirror for tests, odometry and reports."""
import math

import torch

from . import _lib as L
from .ops import vertex_normal_maps


class FusionMap:
    def __init__(self, capacity, height, width, device, dist_th=0.05, angle_th=20.0, sigma=0.6):
        self.cap, self.H, self.W = int(capacity), int(height), int(width)
        self.device = torch.device(device)
        f = dict(device=self.device, dtype=torch.float32)
        self.points = torch.zeros(self.cap, 3, **f)
        self.normals = torch.zeros(self.cap, 3, **f)
        self.colors = torch.zeros(self.cap, 3, **f)
        self.ccounts = torch.zeros(self.cap, **f)
        self.count = torch.zeros(3, device=self.device, dtype=torch.int64)     # {M, scratch, sticky overflow flag} (e2e_pf_fuse_append_dev)
        self._M = 0                                         # host mirror of count[0]; None: the device is ahead (read on demand)
        self._knn = None
        self._knn_dirty = True
        self._frame = None                                  # preallocated frame maps of the resident step
        # thresholds: python doubles rounded to fp32 at the call, like a tensor-vs-float comparison in torch
        self.dist_th = float(dist_th)
        self.dot_th = math.cos(float(angle_th) * math.pi / 180.0)
        self.sigma = float(sigma)
        self.ws = torch.empty(L.load().e2e_pf_workspace_bytes(self.cap, self.H, self.W), device=self.device, dtype=torch.uint8)
        self._count = torch.zeros(1, device=self.device, dtype=torch.int64)
        self._assoc_M = None

    # -- live size ------------------------------------------------------------------------------------------------------
    @property
    def M(self):
        """Live rows.  Synchronises with the device when map steps have run since the last read."""
        if self._M is None:
            c = self.count.cpu()
            if int(c[2]) != 0:
                raise RuntimeError(f"PointFusion map capacity exceeded ({int(c[2])} > {self.cap}); size it for the sequence")
            self._M = int(c[0])
        return self._M

    @M.setter
    def M(self, v):
        v = int(v)
        if v < 0 or v > self.cap:
            raise ValueError("map size out of range")
        self._M = v
        self.count[0] = v
        self._knn_dirty = True

    def check_capacity(self):
        """Raises if any resident map step overflowed the capacity (one host read; call where the host synchronises anyway)."""
        return self.M

    # -- nearest-neighbour index over the live points (rebuilt lazily after every map change) ------------------------
    def knn_index(self, max_queries=None):
        """ONE index buffer for the whole run, sized for the capacity and rebuilt in place after every map change from the
        device-resident point count: no allocation, no host read, constant launch arguments (e2e_knn1_index_build_dev)."""
        mq = int(max_queries or self.H * self.W)
        if self._knn is None or self._knn.max_queries < mq:
            self._knn = ResidentKnnIndex(self, mq)
            self._knn_dirty = True
        if self._knn_dirty:
            self._knn.build()
            self._knn_dirty = False
        return self._knn

    # -- views of the live part -------------------------------------------------------------------
    def live(self):
        M = self.M
        return self.points[:M], self.normals[:M], self.colors[:M], self.ccounts[:M]

    def load_state(self, points, normals, colors, ccounts):
        """Replace the live rows (used by the parity tests to start both sides from the same map)."""
        M = points.shape[0]
        if M > self.cap:
            raise ValueError("map state exceeds capacity")
        self.points[:M], self.normals[:M], self.colors[:M], self.ccounts[:M] = points, normals, colors, ccounts
        self.M = M

    # -- one map step --------------------------------------------------------------------------------
    def frame_maps(self, depth, K, pose):
        """depth (H,W), K (4,4), pose (4,4) -> maps dict with batch dim 1."""
        return vertex_normal_maps(depth.reshape(1, self.H, self.W), K.reshape(1, 4, 4), pose.reshape(1, 4, 4), self.sigma)

    def associate(self, maps, K, pose):
        L.call("e2e_pf_associate", L.ptr(self.points), L.ptr(self.normals), L.ptr(self.ccounts), self.M, L.ptr(K.contiguous()),
               L.ptr(pose.contiguous()), L.ptr(maps["Vg"]), L.ptr(maps["ng"]), self.dist_th, self.dot_th, L.ptr(self.ws), self.cap,
               self.H, self.W, L.stream())
        self._assoc_M = self.M

    def table(self, which):
        """Index table as int64 rows [n, h, w]: 'active' | 'similar' | 'unique' (host sync: reads the row count)."""
        w = {"active": 0, "similar": 1, "unique": 2}[which]
        M = self._assoc_M
        rows = torch.empty(max(M if w < 2 else min(M, self.H * self.W), 1), 3, device=self.device, dtype=torch.int64)
        L.call("e2e_pf_table", w, M, L.ptr(self.ws), self.cap, self.H, self.W, L.ptr(rows), L.ptr(self._count), L.stream())
        return rows[: int(self._count.item())]

    def fuse_append(self, maps, rgb, depth):
        rgb = L.dev(rgb, "rgb").contiguous()
        depth = L.dev(depth, "depth").contiguous()
        L.call("e2e_pf_fuse_append", L.ptr(self.points), L.ptr(self.normals), L.ptr(self.colors), L.ptr(self.ccounts), self.M,
               self.cap, L.ptr(depth), L.ptr(maps["Vg"]), L.ptr(maps["ng"]), L.ptr(rgb), L.ptr(maps["alpha"]), L.ptr(self.ws),
               self.H, self.W, L.ptr(self._count), L.stream())
        new_m = int(self._count.item())                     # one host sync per map step (module path; the driver uses step_resident)
        if new_m > self.cap:
            raise RuntimeError(f"PointFusion map capacity exceeded ({new_m} > {self.cap}); size it for the sequence")
        self.M = new_m                                      # (also marks the index dirty: fused points moved, new ones were appended)

    def step(self, rgb, depth, K, pose):
        """PointFusion.step with a known pose (update_map_fusion).  rgb (H,W,3), depth (H,W).  Host-visible form: returns the
        frame maps and leaves `M` known on the host (one synchronisation)."""
        with torch.no_grad():
            maps = self.frame_maps(depth, K, pose)
            self.associate(maps, K, pose)
            self.fuse_append(maps, rgb, depth)
        return maps

    def step_resident(self, rgb, depth, K, pose):
        """The same map step without any host read or allocation: frame maps into preallocated buffers, association / fusion /
        append with the live size taken from (and written back to) `count` on the device.  rgb (H,W,3), depth (H,W), K / pose (4,4):
        contiguous device tensors.  Results are bit-identical to step()."""
        if self._frame is None:
            f = dict(device=self.device, dtype=torch.float32)
            self._frame = {"Vg": torch.empty(1, self.H, self.W, 3, **f), "ng": torch.empty(1, self.H, self.W, 3, **f),
                           "alpha": torch.empty(1, self.H, self.W, **f)}
            from .ops import fusion_alpha_den
            self._alpha_den = float(fusion_alpha_den(self.sigma))
        m, st = self._frame, L.stream()
        for n, t in (("rgb", rgb), ("depth", depth), ("K", K), ("pose", pose)):
            if not L.dev(t, n).is_contiguous():
                raise ValueError(f"step_resident: {n} must be contiguous")
        L.call("e2e_vertex_normal_maps", L.ptr(depth), L.ptr(K), L.ptr(pose), self._alpha_den, None, None, L.ptr(m["Vg"]), L.ptr(m["ng"]),
               L.ptr(m["alpha"]), 1, self.H, self.W, st)
        L.call("e2e_pf_associate_dev", L.ptr(self.points), L.ptr(self.normals), L.ptr(self.ccounts), L.ptr(self.count), L.ptr(K), L.ptr(pose),
               L.ptr(m["Vg"]), L.ptr(m["ng"]), self.dist_th, self.dot_th, L.ptr(self.ws), self.cap, self.H, self.W, st)
        L.call("e2e_pf_fuse_append_dev", L.ptr(self.points), L.ptr(self.normals), L.ptr(self.colors), L.ptr(self.ccounts), L.ptr(self.count),
               self.cap, L.ptr(depth), L.ptr(m["Vg"]), L.ptr(m["ng"]), L.ptr(rgb), L.ptr(m["alpha"]), L.ptr(self.ws), self.H, self.W, st)
        self._M = None                                      # the device knows; the host asks when it needs to
        self._assoc_M = None
        self._knn_dirty = True


def _mark_updated_on_device(self, index_current):
    """Bookkeeping after a captured map update was REPLAYED (the Python side of step_resident / knn_index did not run)."""
    self._M = None
    self._assoc_M = None
    self._knn_dirty = not index_current


FusionMap.mark_updated_on_device = _mark_updated_on_device


class ResidentKnnIndex:
    """Exact 1-NN grid over the live rows of a FusionMap, in ONE buffer sized for the map's capacity."""
    resident = True

    def __init__(self, fmap, max_queries):
        self.map, self.max_queries = fmap, int(max_queries)
        self.ref = fmap.points                              # (cap,3): rows beyond the live count are never referenced by a result
        self.ws = torch.empty(L.load().e2e_knn1_index_capacity_bytes(self.max_queries, fmap.cap), device=fmap.device, dtype=torch.uint8)

    def build(self):
        L.call("e2e_knn1_index_build_dev", L.ptr(self.ref), L.ptr(self.map.count), self.map.cap, self.max_queries, L.ptr(self.ws), L.stream())

    def query(self, p1, n1, dists, idx, stream, row_len=0, warm=None):
        """row_len > 0: the queries are an image's pixels in row-major order (lanes take 8 x 8 tiles: e2e_knn1_index_query_dev_image).
        warm: int64 indices of an earlier query of nearby points against the same map state (may be `idx` itself): an upper bound the
        search starts from; the result is exact either way (e2e_knn1_index_query_dev_image_warm)."""
        if warm is not None:
            L.call("e2e_knn1_index_query_dev_image_warm", L.ptr(p1), int(n1), int(row_len), L.ptr(self.ref), L.ptr(warm), self.map.cap, self.max_queries, L.ptr(self.ws),
                   L.ptr(dists), L.ptr(idx), stream)
        elif row_len:
            L.call("e2e_knn1_index_query_dev_image", L.ptr(p1), int(n1), int(row_len), self.map.cap, self.max_queries, L.ptr(self.ws), L.ptr(dists), L.ptr(idx), stream)
        else:
            L.call("e2e_knn1_index_query_dev", L.ptr(p1), int(n1), self.map.cap, self.max_queries, L.ptr(self.ws), L.ptr(dists), L.ptr(idx), stream)
