"""Per-entry-point launch timing with HIP events on the stream each call is launched on (bench.py's in-run roofline
figures; `torch.cuda.Event` alone only sees torch's current stream, the backward-weight chains may run on a side stream).

    with KernelTimer() as kt:
        ...                       # any code that goes through e2ehip._lib.call
    kt.summary()  ->  {entry point: {"calls", "ms", "flops", "bytes"}}

Every C-ABI call is bracketed by one event pair, so an entry point that launches several kernels (a split-K GEMM and its
epilogue, a backward-weight GEMM and its slab reduction) is timed as a whole; one pair costs about 2 us of stream time,
which the figures include (they read slightly LOW as rates, never high)."""
import ctypes

import torch

from . import _lib as L


def _conv_flops(name, a):
    """Algorithmic FLOPs of one convolution GEMM call from its argument list (2 x pixels x Cout x Cin x KH x KW)."""
    if name == "e2e_conv2d_fwd":
        B, Hs, Ws, Cin, Cout, KH, KW, stride, pad = a[10:19]
        Ho, Wo = (Hs + 2 * pad - KH) // stride + 1, (Ws + 2 * pad - KW) // stride + 1
        return 2.0 * B * Ho * Wo * Cout * Cin * KH * KW
    if name in ("e2e_conv2d_bwd_data", "e2e_conv2d_bwd_data_acc", "e2e_conv2d_bwd_data_fused"):
        B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW = a[4:13]
        return 2.0 * B * Ho * Wo * Cout * Cin * KH * KW
    if name == "e2e_conv2d_bwd_weight":
        B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW = a[8:17]
        return 2.0 * B * Ho * Wo * Cout * Cin * KH * KW
    if name in ("e2e_conv2d_bwd_weight_scaled", "e2e_conv2d_bwd_weight_scaled_deferred"):
        B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW = a[9:18]
        return 2.0 * B * Ho * Wo * Cout * Cin * KH * KW
    return 0.0


def _conv_bytes(name, a):
    """Algorithmic (compulsory) HBM bytes of one convolution call: every operand once -- the gathered input domain, the weights, the
    result -- in fp32.  (An upsampled / concatenated input counts with its gather domain B x Hs x Ws x Cin: an upper bound of what has to
    be read; the backward-weight form reads dZ and the input and writes dW.)"""
    if name == "e2e_conv2d_fwd":
        B, Hs, Ws, Cin, Cout, KH, KW, stride, pad = a[10:19]
        Ho, Wo = (Hs + 2 * pad - KH) // stride + 1, (Ws + 2 * pad - KW) // stride + 1
    elif name in ("e2e_conv2d_bwd_data", "e2e_conv2d_bwd_data_acc", "e2e_conv2d_bwd_data_fused"):
        B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW = a[4:13]
    elif name == "e2e_conv2d_bwd_weight":
        B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW = a[8:17]
    elif name in ("e2e_conv2d_bwd_weight_scaled", "e2e_conv2d_bwd_weight_scaled_deferred"):
        B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW = a[9:18]
    else:
        return 0
    return 4 * (B * Hs * Ws * Cin + Cout * Cin * KH * KW + B * Ho * Wo * Cout)


def _warp_bytes(name, a):
    """Algorithmic HBM bytes of one fused warp + photometric (+ regulariser) loss-and-gradient launch:
    reads depth 4N + src 12N + tgt 12N (+ init_t, init_s, depth_s 12N), writes g_tgt 4N (+ g_src 4N)."""
    if name == "e2e_warp_photo_lossgrad_hostgeo":
        reg, H, W = a[8], a[18], a[19]
        return (32 + (16 if reg else 0)) * H * W
    if name == "e2e_warp_photo_lossgrad":
        reg, B, H, W = a[10], a[20], a[21], a[22]
        return (32 + (16 if reg else 0)) * B * H * W
    if name == "e2e_warp_photo_lossgrad_chain":
        reg, B, H, W = a[11], a[23], a[24], a[25]
        return (32 + (16 if reg else 0)) * B * H * W
    return 0


def _stream_of(arg):
    v = arg.value if isinstance(arg, ctypes.c_void_p) else arg
    return int(v or 0)


class KernelTimer:
    def __init__(self):
        self.rows = []
        self._streams = {}

    def __enter__(self):
        self._prev = L.PROFILE_HOOK[0]
        L.PROFILE_HOOK[0] = self
        return self

    def __exit__(self, *exc):
        L.PROFILE_HOOK[0] = self._prev

    def _stream(self, ptr):
        s = self._streams.get(ptr)
        if s is None:
            s = self._streams[ptr] = torch.cuda.ExternalStream(ptr) if ptr else torch.cuda.default_stream()
        return s

    def around(self, name, args, fn):
        s = self._stream(_stream_of(args[-1]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        rc = fn()
        e1.record(s)
        plain = [a.value if isinstance(a, ctypes.c_void_p) else a for a in args]
        self.rows.append((name, e0, e1, _conv_flops(name, plain), _warp_bytes(name, plain) + _conv_bytes(name, plain)))
        return rc

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, fl, by in self.rows:
            r = out.setdefault(name, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0})
            r["calls"] += 1
            r["ms"] += e0.elapsed_time(e1)
            r["flops"] += fl
            r["bytes"] += by
        return out
