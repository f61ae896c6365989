"""Convolution-level operators of the depth network.

Backend selection (E2E_CONV_BACKEND):
  "hip"     hand-written fp32-MFMA implicit-GEMM kernels (csrc/conv.hip) -- the product path.
  "miopen"  torch.nn.functional (MIOpen) -- BRING-UP SCAFFOLD ONLY, kept so that the rest of the pipeline
            could be validated end to end before the native kernels existed and as
            an A/B reference when tuning them.  Never the default once csrc/conv.hip covers a layer type.
Activations are NCHW-shaped tensors in channels_last memory (NHWC), the layout the implicit-GEMM kernels want
(K = Cin contiguous) and the layout the reference's frames already have (online_adaption.py:215-220).
"""
import os

import torch
import torch.nn.functional as F

BACKEND = os.environ.get("E2E_CONV_BACKEND", "auto")


def _use_hip(x):
    if not x.is_cuda:
        from ._lib import E2EError
        raise E2EError(f"the depth network runs on the HIP device only (got a {x.device} tensor); there is no CPU fallback")
    if BACKEND == "miopen":
        return False
    from . import conv
    return conv.available()


def conv2d(x, weight, bias=None, stride=1, padding=0, pad_mode="zeros", act=None, bn=None):
    """conv (+ eval-mode BN) (+ activation).  pad_mode: "zeros" | "reflect".  act: None | "relu" | "elu".
    bn: None or (weight, bias, running_mean, running_var, eps)."""
    if _use_hip(x):
        from . import conv
        return conv.conv2d(x, weight, bias, stride, padding, pad_mode, act, bn)
    if pad_mode == "reflect" and padding:
        x = F.pad(x, (padding,) * 4, mode="reflect")
        padding = 0
    y = F.conv2d(x, weight, bias, stride, padding)
    if bn is not None:
        w, b, rm, rv, eps = bn
        y = F.batch_norm(y, rm, rv, w, b, False, 0.0, eps)
    if act == "relu":
        y = F.relu(y)
    elif act == "elu":
        y = F.elu(y)
    return y


def max_pool_3x3_s2(x):
    return F.max_pool2d(x, 3, 2, 1)


def upsample2_concat(x, skip=None):
    """nearest x2 upsample of x, concatenated with `skip` along channels (networks.py:218-221,283-286)."""
    y = F.interpolate(x, scale_factor=2, mode="nearest")
    return y if skip is None else torch.cat([y, skip], 1)
