import warnings

import torch
import torch.nn as nn

from ..structures import Pointclouds, RGBDImages


class ICPSLAM(nn.Module):
    """Point-based SLAM with plain aggregation as the map update (gradslam ICPSLAM).
    odom: "gt" uses the frame's own poses; "icp" / "gradicp" run frame-to-model point-to-plane ICP (e2ehip.icp)."""

    def __init__(self, *, odom="gradicp", dsratio=4, numiters=20, damp=1e-8, dist_thresh=None, lambda_max=2.0, B=1.0, B2=1.0,
                 nu=200.0, device=None):
        super().__init__()
        if odom not in ("gt", "icp", "gradicp"):
            raise ValueError(f"odometry method ({odom}) not supported for PointFusion. Currently supported odometry modules for PointFusion are: 'gt', 'icp', 'gradicp'")
        self.odom, self.dsratio, self.numiters = odom, dsratio, numiters
        self.damp, self.dist_thresh, self.lambda_max, self.B, self.B2, self.nu = damp, dist_thresh, lambda_max, B, B2, nu
        self.device = torch.device(device) if device is not None else torch.device("cpu")

    # -- odometry ------------------------------------------------------------------------------------
    def _localize(self, pointclouds, live_frame, prev_frame):
        if not isinstance(pointclouds, Pointclouds):
            raise TypeError(f"Expected pointclouds to be of type gradslam.Pointclouds. Got {type(pointclouds)}.")
        if not isinstance(live_frame, RGBDImages):
            raise TypeError(f"Expected live_frame to be of type gradslam.RGBDImages. Got {type(live_frame)}.")
        if not isinstance(prev_frame, (RGBDImages, type(None))):
            raise TypeError(f"Expected prev_frame to be of type gradslam.RGBDImages or None. Got {type(prev_frame)}.")
        if prev_frame is not None and self.odom == "gt":
            warnings.warn("`prev_frame` is not used when using `odom='gt'` (should be None)")
        if prev_frame is None or self.odom == "gt":
            if pointclouds.has_points and prev_frame is None and self.odom != "gt":
                raise ValueError(f"`odom={self.odom}` with a non-empty map needs `prev_frame`")
            if not live_frame.has_poses:
                raise ValueError("`live_frame` must have poses when `prev_frame` is None or `odom='gt'`")
            return live_frame.poses
        # frame-to-model ICP against the active map points, initialised with the previous frame's pose
        from e2ehip import icp
        if len(pointclouds) != 1 or live_frame.shape[0] != 1:
            raise NotImplementedError("batch size 1 only (OPTIMIZATION.batch_size, configs/config.yaml:60)")
        if not pointclouds.has_points:
            raise ValueError("frame-to-model odometry needs a non-empty map")
        fm = self._resident_map(pointclouds, live_frame)
        pose, self.last_trace = icp.frame_to_model(fm, live_frame.depth_image[0, 0, ..., 0].detach(), live_frame.intrinsics[0, 0],
                                                   prev_frame.poses[0, 0], dsratio=self.dsratio, numiters=self.numiters, damp=self.damp,
                                                   dist_thresh=self.dist_thresh, mode=self.odom, lambda_max=self.lambda_max, B=self.B,
                                                   B2=self.B2, nu=self.nu)
        return pose.view(1, 1, 4, 4)

    def _resident_map(self, pointclouds, live_frame):
        from e2ehip.fusionmap import FusionMap
        fm = pointclouds._fusion_maps
        if fm is None:
            _, _, H, W = live_frame.shape
            M = pointclouds.points_list[0].shape[0]
            fm = FusionMap(max(getattr(self, "map_capacity", None) or 0, M + 64 * H * W), H, W, live_frame.device,
                           getattr(self, "dist_th", 0.05), getattr(self, "angle_th", 20), getattr(self, "sigma", 0.6))
            fm.load_state(pointclouds.points_list[0].detach(), pointclouds.normals_list[0].detach(), pointclouds.colors_list[0].detach(),
                          pointclouds.features_list[0].detach().reshape(-1))
            pointclouds._fusion_maps = fm
        return fm

    # -- map update ------------------------------------------------------------------------------------
    def _map(self, pointclouds, live_frame, inplace=False):
        from .pointfusion import frame_as_pointcloud
        new = frame_as_pointcloud(live_frame)
        target = pointclouds if inplace else pointclouds.clone()
        return target.append_points(new)

    def step(self, pointclouds, live_frame, prev_frame=None, inplace=False):
        live_frame.poses = self._localize(pointclouds, live_frame, prev_frame)
        pointclouds = self._map(pointclouds, live_frame, inplace)
        return pointclouds, live_frame.poses

    def forward(self, frames):
        if not isinstance(frames, RGBDImages):
            raise TypeError(f"Expected frames to be of type gradslam.RGBDImages. Got {type(frames)}.")
        B, L = frames.shape[:2]
        pc = Pointclouds(device=frames.device)
        poses, prev = [], None
        for s in range(L):
            live = frames[:, s]
            if s == 0 and live.poses is None:
                live.poses = torch.eye(4, device=frames.device).view(1, 1, 4, 4).repeat(B, 1, 1, 1)
            pc, live.poses = self.step(pc, live, prev, inplace=True)
            prev = live if self.odom != "gt" else None
            poses.append(live.poses[:, 0])
        return pc, torch.stack(poses, 1)
