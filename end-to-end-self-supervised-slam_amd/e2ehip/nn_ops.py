"""Convolution-level operators of the depth network: thin argument plumbing over the native kernels (csrc/conv.hip,
csrc/nn_misc.hip).  There is ONE backend -- the hand-written HIP kernels; a layer shape they do not cover raises
(no torch / MIOpen path: the A/B scaffold that round 1 kept here now lives in tools/conv_bench.py, outside the product).

Activations are NCHW-shaped tensors in channels_last memory (NHWC), the layout the implicit-GEMM kernels want
(K = Cin contiguous) and the layout the reference's frames already have (online_adaption.py:215-220).
"""
import torch

from . import conv
from ._lib import E2EError


def _require_device(x):
    if not x.is_cuda:
        raise E2EError(f"the depth network runs on the HIP device only (got a {x.device} tensor); there is no CPU fallback")


def _fold_bn(bn):
    """eval-mode BatchNorm as per-channel (scale, shift); cached ON the BatchNorm's weight tensor until any of its tensors changes
    (round 3 kept a module-level dict keyed by id(weight): an id -- and, through the caching allocator, the four data pointers -- of
    a dead model's BatchNorm can be reissued to a new one, whose fold would then have been the dead model's)."""
    w, b, rm, rv, eps = bn
    key = (w.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), w._version, b._version, rm._version, rv._version, float(eps),
           conv.epoch_of(w), conv.epoch_of(b))
    hit = getattr(w, "_e2e_bnfold", None)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            scale = (w / torch.sqrt(rv + eps)).contiguous()
            shift = (b - rm * scale).contiguous()
        hit = (key, scale, shift)
        w._e2e_bnfold = hit
    return hit[1], hit[2]


def conv2d(x, weight, bias=None, stride=1, padding=0, pad_mode="zeros", act=None, bn=None, residual=None, skip=None, upsample=1,
           in_norm=None):
    """act( BN_eval( conv( cat(nearest_up(x, upsample), skip) ) + bias ) + residual ).
    bn: None or (weight, bias, running_mean, running_var, eps) of an eval-mode BatchNorm; a FROZEN one is folded into the
    epilogue as constants, one whose affine parameters still train (the reference's `downsample.1`, online_adaption.py:182-184)
    goes through conv.conv2d_bn_affine so that gamma / beta receive their gradients.
    in_norm: (sub, mul) applied to x before the convolution (the stem's (x - 0.45) / 0.225)."""
    _require_device(x)
    if not conv.supports(weight):
        raise NotImplementedError(f"convolution weight {tuple(weight.shape)}: the native kernels cover Cout % 16 == 0, the 16 -> 1 3x3 "
                                  "reflect disparity head and the 1 -> 1 1x1 scale layer; there is no library fallback")
    if bn is not None and (bn[0].requires_grad or bn[1].requires_grad):
        if bias is not None or act not in (None, "relu") or skip is not None or upsample != 1:
            raise NotImplementedError("a trainable eval-mode BatchNorm follows a plain convolution (ResNet body) only")
        return conv.conv2d_bn_affine(x, weight, bn, stride, padding, pad_mode, residual, act == "relu", in_norm)
    return conv.conv2d(x, weight, bias, stride, padding, pad_mode, act, _fold_bn(bn) if bn is not None else None, residual, skip,
                       upsample, in_norm)


def max_pool_3x3_s2(x):
    """nn.MaxPool2d(3, 2, 1) of the ResNet stem (networks.py:53), native forward / backward (argmax recomputed, no index tensor)."""
    _require_device(x)
    return conv.max_pool_3x3_s2(x)


def upsample2_concat(x, skip=None):
    """nearest x2 upsample of x, concatenated with `skip` along channels (networks.py:218-221,283-286); only the
    stand-alone `upsample()` helper uses this -- inside the decoder the operation is fused into the next convolution."""
    _require_device(x)
    return conv.upsample2_concat(x, skip)


def scale_layer(x, scale):
    """ScaleLayer.forward (networks.py:206-215): x * scale for a one-element parameter, on the single-channel depth map."""
    _require_device(x)
    if x.dim() != 4 or x.shape[1] != 1 or scale.numel() != 1:
        raise NotImplementedError("ScaleLayer is applied to a (B,1,H,W) depth / disparity map with a single scale")
    return conv._Affine.apply(x, scale, None, None, None, 0.0, None, False)
