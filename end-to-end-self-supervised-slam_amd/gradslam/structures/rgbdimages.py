import torch

from e2ehip import ops


class RGBDImages:
    """Channels-last RGB-D frame stack: rgb (B,L,H,W,3), depth (B,L,H,W,1), intrinsics (B,1,4,4),
    poses (B,L,4,4) or None.  Vertex / normal maps are computed lazily by one HIP kernel per frame set and
    cached; setting rgb / depth / poses invalidates the cache (gradslam RGBDImages)."""

    def __init__(self, rgb_image, depth_image, intrinsics, poses=None, channels_first=False, device=None):
        if channels_first:
            raise NotImplementedError("the reference only builds channels-last RGBDImages")
        for n, t in (("rgb_image", rgb_image), ("depth_image", depth_image), ("intrinsics", intrinsics)):
            if not torch.is_tensor(t):
                raise TypeError(f"Expected {n} to be of type tensor. Got {type(t)}.")
        if rgb_image.ndim != 5 or depth_image.ndim != 5:
            raise ValueError(f"rgb_image / depth_image should have ndim=5, got {rgb_image.ndim} / {depth_image.ndim}")
        if intrinsics.ndim != 4 or tuple(intrinsics.shape[1:]) != (1, 4, 4):
            raise ValueError(f"intrinsics should have shape (B,1,4,4), got {tuple(intrinsics.shape)}")
        if tuple(depth_image.shape) != tuple(rgb_image.shape[:4]) + (1,):
            raise ValueError(f"depth_image shape {tuple(depth_image.shape)} does not match rgb_image {tuple(rgb_image.shape)}")
        if rgb_image.shape[-1] != 3 or intrinsics.shape[0] != rgb_image.shape[0]:
            raise ValueError("rgb_image must be (B,L,H,W,3) and intrinsics must share its batch size")
        if poses is not None:
            if not torch.is_tensor(poses):
                raise TypeError(f"Expected poses to be of type tensor. Got {type(poses)}.")
            if tuple(poses.shape) != tuple(rgb_image.shape[:2]) + (4, 4):
                raise ValueError(f"poses should have shape (B,L,4,4), got {tuple(poses.shape)}")
        devs = {t.device for t in (rgb_image, depth_image, intrinsics) + ((poses,) if poses is not None else ())}
        if len(devs) != 1:
            raise ValueError(f"All inputs must be on one device, got {devs}")
        self._rgb, self._depth, self._K, self._poses = rgb_image, depth_image, intrinsics, poses
        self._cache = None
        if device is not None:
            self.to(device)

    # ---- basic properties ------------------------------------------------------------------------
    @property
    def shape(self):
        return tuple(self._rgb.shape[:4])

    @property
    def device(self):
        return self._rgb.device

    @property
    def channels_first(self):
        return False

    @property
    def rgb_image(self):
        return self._rgb

    @rgb_image.setter
    def rgb_image(self, v):
        if tuple(v.shape) != tuple(self._rgb.shape):
            raise ValueError("rgb_image shape must not change")
        self._rgb, self._cache = v, None

    @property
    def depth_image(self):
        return self._depth

    @depth_image.setter
    def depth_image(self, v):
        if tuple(v.shape) != tuple(self._depth.shape):
            raise ValueError("depth_image shape must not change")
        self._depth, self._cache = v, None

    @property
    def intrinsics(self):
        return self._K

    @property
    def poses(self):
        return self._poses

    @poses.setter
    def poses(self, v):
        if v is not None and tuple(v.shape) != self.shape[:2] + (4, 4):
            raise ValueError(f"poses should have shape {self.shape[:2] + (4, 4)}, got {tuple(v.shape)}")
        self._poses, self._cache = v, None

    @property
    def has_poses(self):
        return self._poses is not None

    # ---- maps --------------------------------------------------------------------------------------
    def _maps(self):
        if self._cache is None:
            B, L, H, W = self.shape
            K = self._K.expand(B, L, 4, 4).reshape(B * L, 4, 4)
            poses = self._poses if self._poses is not None else torch.eye(4, device=self.device).expand(B, L, 4, 4)
            m = ops.vertex_normal_maps(self._depth.reshape(B * L, H, W), K, poses.reshape(B * L, 4, 4))
            self._cache = {k: (v.reshape((B, L) + tuple(v.shape[1:])) if v.dim() == 4 else v.reshape(B, L, H, W, 1)) for k, v in m.items()}
        return self._cache

    @property
    def valid_depth_mask(self):
        return self._depth != 0

    @property
    def vertex_map(self):
        return self._maps()["V"]

    @property
    def normal_map(self):
        return self._maps()["n"]

    @property
    def global_vertex_map(self):
        return self._maps()["Vg"]

    @property
    def global_normal_map(self):
        return self._maps()["ng"]

    # ---- container protocol ------------------------------------------------------------------------
    def __getitem__(self, index):
        """rgbd[:, s] keeps the sequence dimension (length 1), as gradslam does."""
        if isinstance(index, tuple) and len(index) == 2 and index[0] == slice(None):
            s = index[1]
            s = slice(s, s + 1) if isinstance(s, int) else s
            return RGBDImages(self._rgb[:, s], self._depth[:, s], self._K, None if self._poses is None else self._poses[:, s])
        if isinstance(index, (int, slice)):
            b = slice(index, index + 1) if isinstance(index, int) else index
            return RGBDImages(self._rgb[b], self._depth[b], self._K[b], None if self._poses is None else self._poses[b])
        raise IndexError("RGBDImages supports [b] and [:, s] indexing")

    def __len__(self):
        return self.shape[0]

    def to(self, device):
        self._rgb, self._depth, self._K = self._rgb.to(device), self._depth.to(device), self._K.to(device)
        self._poses = None if self._poses is None else self._poses.to(device)
        self._cache = None
        return self

    def detach(self):
        return RGBDImages(self._rgb.detach(), self._depth.detach(), self._K.detach(), None if self._poses is None else self._poses.detach())

    def clone(self):
        return RGBDImages(self._rgb.clone(), self._depth.clone(), self._K.clone(), None if self._poses is None else self._poses.clone())
