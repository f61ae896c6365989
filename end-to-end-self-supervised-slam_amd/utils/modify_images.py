"""Import-surface stand-in for utils/modify_images.py (online_adaption.py:22 imports corrupt_rgbd; only
gradient_experiments.py:111 calls it -- OUT OF SCOPE)."""

_MSG = "out of scope: SURVEY.md section 2 row P9 -- RGB-D corruption for the gradient-flow experiments"


def corrupt_rgbd(*args, **kwargs):
    raise NotImplementedError(_MSG)
