"""Convolution-level operators of the depth network.

Backend selection (E2E_CONV_BACKEND):
  "hip" (default)  hand-written fp32-MFMA implicit-GEMM kernels (csrc/conv.hip) with padding / upsample / concat /
                   eval-BN / residual / activation fused -- the product path.
  "miopen"         torch.nn.functional (MIOpen) -- BRING-UP SCAFFOLD, kept only as the A/B reference for the native
                   kernels.
Two small pieces still run through torch on either backend and are listed in DESIGN.md: the 1-channel disparity head
(Cout = 1 does not fill an MFMA tile) and the stem's 3x3/2 max-pool.
Activations are NCHW-shaped tensors in channels_last memory (NHWC), the layout the implicit-GEMM kernels want
(K = Cin contiguous) and the layout the reference's frames already have (online_adaption.py:215-220).
"""
import os

import torch
import torch.nn.functional as F

BACKEND = os.environ.get("E2E_CONV_BACKEND", "hip")
_BN_CACHE = {}


def _use_hip(x):
    if not x.is_cuda:
        from ._lib import E2EError
        raise E2EError(f"the depth network runs on the HIP device only (got a {x.device} tensor); there is no CPU fallback")
    if BACKEND == "miopen":
        return False
    from . import conv
    if not conv.available():
        from ._lib import E2EError
        raise E2EError("libe2eslam_hip.so lacks the convolution kernels; rebuild it (python __graft_entry__.py build)")
    return True


def _fold_bn(bn):
    """eval-mode BatchNorm as per-channel (scale, shift); cached until any of its tensors changes."""
    w, b, rm, rv, eps = bn
    key = (w.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), w._version, b._version, rm._version, rv._version, float(eps))
    hit = _BN_CACHE.get(id(w))
    if hit is None or hit[0] != key:
        with torch.no_grad():
            scale = (w / torch.sqrt(rv + eps)).contiguous()
            shift = (b - rm * scale).contiguous()
        hit = (key, scale, shift)
        _BN_CACHE[id(w)] = hit
    return hit[1], hit[2]


def conv2d(x, weight, bias=None, stride=1, padding=0, pad_mode="zeros", act=None, bn=None, residual=None, skip=None, upsample=1,
           in_norm=None):
    """act( BN_eval( conv( cat(nearest_up(x, upsample), skip) ) + bias ) + residual ).
    bn: None or (weight, bias, running_mean, running_var, eps) of a FROZEN eval-mode BatchNorm.
    in_norm: (sub, mul) applied to x before the convolution (the stem's (x - 0.45) / 0.225)."""
    from . import conv
    if _use_hip(x) and conv.supports(weight):
        return conv.conv2d(x, weight, bias, stride, padding, pad_mode, act, _fold_bn(bn) if bn is not None else None, residual, skip,
                           upsample, in_norm)
    # ---- scaffold / tiny-layer path (torch) --------------------------------------------------------------------------
    if in_norm is not None:
        x = (x - in_norm[0]) * in_norm[1]
    if upsample != 1:
        x = F.interpolate(x, scale_factor=upsample, mode="nearest")
    if skip is not None:
        x = torch.cat([x, skip], 1)
    if pad_mode == "reflect" and padding:
        x = F.pad(x, (padding,) * 4, mode="reflect")
        padding = 0
    y = F.conv2d(x, weight, bias, stride, padding)
    if bn is not None:
        w, b, rm, rv, eps = bn
        y = F.batch_norm(y, rm, rv, w, b, False, 0.0, eps)
    if residual is not None:
        y = y + residual
    if act == "relu":
        y = F.relu(y)
    elif act == "elu":
        y = F.elu(y)
    elif act == "disp":
        y = 10 * torch.sigmoid(y) + 0.01
    return y


def max_pool_3x3_s2(x):
    return F.max_pool2d(x, 3, 2, 1)


def upsample2_concat(x, skip=None):
    """nearest x2 upsample of x, concatenated with `skip` along channels (networks.py:218-221,283-286); only the
    stand-alone `upsample()` helper uses this -- inside the decoder the operation is fused into the next convolution."""
    y = F.interpolate(x, scale_factor=2, mode="nearest")
    return y if skip is None else torch.cat([y, skip], 1)
