"""The subset of `chamferdist` the reference uses (README.md:17-19; loss/losses.py:3; online_adaption.py:33),
backed by the HIP nearest-neighbour kernel."""
from .chamfer import ChamferDistance, knn_points  # noqa: F401
