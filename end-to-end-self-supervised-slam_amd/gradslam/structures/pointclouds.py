import torch


class Pointclouds:
    """Batch of variable-length point clouds held as per-batch lists (points / normals / colors (N,3),
    features (N,1) = fusion confidence counts).  gradslam Pointclouds, list view only: every reference call
    site reads `.points_list[0]` / `.has_points` (online_adaption.py:224,642-643)."""

    def __init__(self, points=None, normals=None, colors=None, features=None, device=None):
        self.points_list = list(points) if points is not None else []
        self.normals_list = list(normals) if normals is not None else []
        self.colors_list = list(colors) if colors is not None else []
        self.features_list = list(features) if features is not None else []
        self._device = torch.device(device) if device is not None else (self.points_list[0].device if self.points_list else torch.device("cpu"))
        self._fusion_maps = None          # resident e2ehip FusionMap per batch element (owned by PointFusion)

    @property
    def device(self):
        return self._device

    def __len__(self):
        return len(self.points_list)

    @property
    def has_points(self):
        return any(p.shape[0] > 0 for p in self.points_list)

    @property
    def has_normals(self):
        return len(self.normals_list) > 0

    @property
    def has_colors(self):
        return len(self.colors_list) > 0

    @property
    def has_features(self):
        return len(self.features_list) > 0

    @property
    def num_points_per_pointcloud(self):
        return torch.tensor([p.shape[0] for p in self.points_list], device=self._device)

    def _map(self, fn):
        out = Pointclouds([fn(p) for p in self.points_list], [fn(p) for p in self.normals_list], [fn(p) for p in self.colors_list],
                          [fn(p) for p in self.features_list], device=self._device)
        out._fusion_maps = self._fusion_maps
        return out

    def detach(self):
        return self._map(lambda t: t.detach())

    def clone(self):
        out = self._map(lambda t: t.clone())
        out._fusion_maps = None
        return out

    def to(self, device):
        out = self._map(lambda t: t.to(device))
        out._device, out._fusion_maps = torch.device(device), None
        return out

    def __getitem__(self, i):
        sl = slice(i, i + 1) if isinstance(i, int) else i
        return Pointclouds(self.points_list[sl], self.normals_list[sl], self.colors_list[sl], self.features_list[sl], device=self._device)

    def append_points(self, other):
        """Per-batch concatenation (update_map_aggregate)."""
        if not isinstance(other, Pointclouds):
            raise TypeError(f"Append object must be of type Pointclouds, got {type(other)}")
        if len(self) == 0:
            self.points_list, self.normals_list = [p.clone() for p in other.points_list], [p.clone() for p in other.normals_list]
            self.colors_list, self.features_list = [p.clone() for p in other.colors_list], [p.clone() for p in other.features_list]
            return self
        if len(self) != len(other):
            raise ValueError("Batch sizes must match")
        for name in ("points_list", "normals_list", "colors_list", "features_list"):
            a, b = getattr(self, name), getattr(other, name)
            if a and b:
                setattr(self, name, [torch.cat([x, y], 0) for x, y in zip(a, b)])
        return self

    def plotly(self, *a, **k):
        raise NotImplementedError("visualisation is out of the hot-path scope (SURVEY.md 2.1 P10)")
