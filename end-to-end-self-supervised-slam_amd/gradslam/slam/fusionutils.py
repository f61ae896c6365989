"""gradslam.slam.fusionutils entry points used by the reference (online_adaption.py:35 imports
find_active_map_points; PointFusion.step drives the rest)."""
import torch

from e2ehip.fusionmap import FusionMap


def _single(pointclouds, rgbdimages):
    if len(pointclouds) > 1 or rgbdimages.shape[0] != 1:
        raise NotImplementedError("batch size 1 only (OPTIMIZATION.batch_size, configs/config.yaml:60)")
    if rgbdimages.shape[1] != 1:
        raise ValueError(f"Expected rgbdimages to have sequence length of 1. Got {rgbdimages.shape[1]}.")


def _scratch_map(pointclouds, rgbdimages, **kw):
    _, _, H, W = rgbdimages.shape
    M = pointclouds.points_list[0].shape[0] if len(pointclouds) else 0
    fm = FusionMap(M + H * W, H, W, rgbdimages.device, **kw)
    if M:
        fm.load_state(pointclouds.points_list[0].detach(), pointclouds.normals_list[0].detach(), pointclouds.colors_list[0].detach(),
                      pointclouds.features_list[0].detach().reshape(-1))
    return fm


def find_active_map_points(pointclouds, rgbdimages):
    """-> (P,4) int64 rows [b, n, h, w] of the map points that project into the live frame (ascending n)."""
    _single(pointclouds, rgbdimages)
    if not pointclouds.has_points:
        return torch.zeros(0, 4, dtype=torch.int64, device=rgbdimages.device)
    if not rgbdimages.has_poses:
        raise ValueError("Pointclouds should be in global frame: rgbdimages must have poses")
    fm = _scratch_map(pointclouds, rgbdimages)
    K, pose = rgbdimages.intrinsics[0, 0], rgbdimages.poses[0, 0]
    maps = fm.frame_maps(rgbdimages.depth_image[0, 0, ..., 0].detach(), K, pose)
    fm.associate(maps, K, pose)
    rows = fm.table("active")
    return torch.cat([torch.zeros(rows.shape[0], 1, dtype=torch.int64, device=rows.device), rows], 1)
