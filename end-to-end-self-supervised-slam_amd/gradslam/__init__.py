"""The subset of `gradslam` the reference's hot path uses (README.md:9-21; online_adaption.py:29-36),
re-implemented for MI355X: vertex/normal maps, PointFusion association / fusion and rigid transforms are
HIP kernels (include/e2eslam.h); these classes only carry tensors and call them.

Semantics follow SURVEY.md Appendix A (gradslam is not vendored in the reference: parity unpinned).
Supported: batch size 1 (OPTIMIZATION.batch_size, configs/config.yaml:60), odom "gt" (or prev_frame=None).
ICP / GradICP odometry is the next scope row and raises NotImplementedError.
"""
from .structures import Pointclouds, RGBDImages  # noqa: F401
from . import datasets, geometry, slam  # noqa: F401
