"""Drop-in for slam/custom_slam.py (reference: slam/custom_slam.py:6-35)."""
import torch

from gradslam.structures import Pointclouds


def image_recover_slam(noisy_rgbd, slam, device):
    """Feed an RGBDImages sequence frame by frame through `slam.step`; every frame but the last is detached so
    that gradients only reach the final (corrupted) frame.  No `prev_frame` is passed, exactly like the
    reference, so the SLAM object must run with odom='gt' (or the map must be empty)."""
    cloud = Pointclouds(device=device)
    batch, length = noisy_rgbd.shape[:2]
    identity = torch.eye(4, device=device).view(1, 1, 4, 4).repeat(batch, 1, 1, 1)
    for s in range(length):
        frame = noisy_rgbd[:, s].to(device)
        if s < length - 1:
            frame = frame.detach()
        if s == 0 and frame.poses is None:
            frame.poses = identity
        cloud, frame.poses = slam.step(cloud, frame)
        frame.poses = frame.poses.detach()
    return cloud
