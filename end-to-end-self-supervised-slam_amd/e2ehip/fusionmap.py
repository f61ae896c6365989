"""Resident PointFusion map for ONE sequence (one process / GPU owns one): capacity-sized arrays that
stay in HBM for the whole run and are appended in place.  60 frames x 307 200 px x 40 B = 737 MB, a
rounding error on 288 GB, so the default capacity simply covers the whole sequence."""
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
        self.M = 0
        # thresholds: python doubles rounded to fp32 at the call, like a tensor-vs-float comparison in torch
        self.dist_th = float(dist_th)
        self.dot_th = math.cos(float(angle_th) * math.pi / 180.0)
        self.sigma = float(sigma)
        self.ws = torch.empty(L.load().e2e_pf_workspace_bytes(self.cap, self.H, self.W), device=self.device, dtype=torch.uint8)
        self._count = torch.zeros(1, device=self.device, dtype=torch.int64)
        self._assoc_M = None

    # -- nearest-neighbour index over the live points (rebuilt lazily after every map change) ------------------------
    def knn_index(self, max_queries=None):
        from .ops import KnnIndex
        mq = int(max_queries or self.H * self.W)
        idx = getattr(self, "_knn", None)
        if idx is None or idx[0] != self.M or idx[1].max_queries < mq:
            idx = (self.M, KnnIndex(self.points[: self.M], mq))
            self._knn = idx
        return idx[1]

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
        self._knn = None

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
        new_m = int(self._count.item())                     # one host sync per map step
        if new_m > self.cap:
            raise RuntimeError(f"PointFusion map capacity exceeded ({new_m} > {self.cap}); size it for the sequence")
        self.M = new_m
        self._knn = None                                    # fused points moved, new ones were appended

    def step(self, rgb, depth, K, pose):
        """PointFusion.step with a known pose (update_map_fusion).  rgb (H,W,3), depth (H,W)."""
        with torch.no_grad():
            maps = self.frame_maps(depth, K, pose)
            self.associate(maps, K, pose)
            self.fuse_append(maps, rgb, depth)
        return maps
