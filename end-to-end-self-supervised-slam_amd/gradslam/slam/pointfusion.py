from ..structures import Pointclouds
from .icpslam import ICPSLAM


def frame_as_pointcloud(frame):
    """All valid-depth pixels of a 1-frame RGBDImages as a Pointclouds (row-major order), differentiable wrt
    depth through the global vertex map.  This is what PointFusion.step returns for an EMPTY map
    (online_adaption.py:461-469) and what ICPSLAM's aggregation appends."""
    B, L, H, W = frame.shape
    if L != 1:
        raise ValueError(f"Expected a frame with sequence length 1. Got {L}.")
    m = frame._maps()
    pts, nrm, col, feat = [], [], [], []
    for b in range(B):
        keep = m["valid"][b, 0, ..., 0] if m["valid"].dim() == 5 else m["valid"][b, 0]
        pts.append(m["Vg"][b, 0][keep])
        nrm.append(m["ng"][b, 0][keep])
        col.append(frame.rgb_image[b, 0][keep])
        feat.append(m["alpha"][b, 0][keep].reshape(-1, 1))
    return Pointclouds(pts, nrm, col, feat, device=frame.device)


class PointFusion(ICPSLAM):
    """PointFusion map update (gradslam PointFusion = ICPSLAM + update_map_fusion).

    The global map lives in a resident e2ehip.FusionMap (capacity-sized HBM arrays, appended in place); the
    returned Pointclouds exposes zero-copy views of its live rows.  `map_capacity` points are reserved on
    first use (default: 64 frames' worth)."""

    def __init__(self, *, odom="gradicp", dist_th=0.05, angle_th=20, sigma=0.6, dsratio=4, numiters=20, damp=1e-8, dist_thresh=None,
                 lambda_max=2.0, B=1.0, B2=1.0, nu=200.0, device=None, map_capacity=None):
        super().__init__(odom=odom, dsratio=dsratio, numiters=numiters, damp=damp, dist_thresh=dist_thresh, lambda_max=lambda_max,
                         B=B, B2=B2, nu=nu, device=device)
        if not isinstance(dist_th, (float, int)):
            raise TypeError(f"Distance threshold must be of type float or int; but was of type {type(dist_th)}.")
        if not isinstance(angle_th, (float, int)):
            raise TypeError(f"Angle threshold must be of type float or int; but was of type {type(angle_th)}.")
        if dist_th < 0:
            raise ValueError(f"Distance threshold must be non-negative: {dist_th}")
        if not 0 <= angle_th <= 90:
            raise ValueError(f"Angle threshold must be in [0, 90]: {angle_th}")
        self.dist_th, self.angle_th, self.sigma = dist_th, angle_th, sigma
        self.map_capacity = map_capacity

    def _map(self, pointclouds, live_frame, inplace=False):
        if len(pointclouds) > 1 or live_frame.shape[0] != 1:
            raise NotImplementedError("batch size 1 only (OPTIMIZATION.batch_size, configs/config.yaml:60)")
        _, _, H, W = live_frame.shape
        if not pointclouds.has_points:
            # empty map: nothing to associate with -> the frame's valid pixels, still attached to the autograd
            # graph of the depth (the 3-D loss differentiates through this: online_adaption.py:461-469,638-645)
            out = frame_as_pointcloud(live_frame)
            return out
        fm = self._resident_map(pointclouds, live_frame)      # adopts an externally built cloud once
        fm.step(live_frame.rgb_image[0, 0].detach(), live_frame.depth_image[0, 0, ..., 0].detach(),
                live_frame.intrinsics[0, 0], live_frame.poses[0, 0])
        P, Nn, C, cc = fm.live()
        out = Pointclouds([P], [Nn], [C], [cc.reshape(-1, 1)], device=live_frame.device)
        out._fusion_maps = fm
        return out
