from e2ehip import ops


def transform_pointcloud(pointcloud, transform):
    """(N,3) points, (4,4) rigid transform -> (R p^T + t)^T, differentiable wrt the points
    (called at online_adaption.py:642)."""
    return ops.transform_points(pointcloud, transform)
