"""Pre-planned (allocation-free, hipGraph-capturable) launches of the fused refinement kernels.

`WarpPhotoPlan` owns every intermediate buffer of the image-space part of one refinement step, so a
step is exactly: e2e_warp_photo_fwd (+ its 1-block reduction) and e2e_warp_photo_bwd -- no
allocator traffic, no host synchronisation.  The build's driver and bench.py use it; the autograd
form for ad-hoc use is ops.warp_photometric."""
import ctypes

import torch

from . import _lib as L
from .ops import _padding


class WarpPhotoPlan:
    def __init__(self, B, H, W, device, padding_mode="border", use_mask=True, reg_kind="l2"):
        self.B, self.H, self.W = B, H, W
        self.pad = _padding(padding_mode)
        self.use_mask = int(bool(use_mask))
        self.reg = {None: 0, "l1": 1, "l2": 2}[reg_kind]
        f = dict(device=device, dtype=torch.float32)
        self.synth = torch.empty(B, 3, H, W, **f)
        self.valid = torch.empty(B, 1, H, W, **f)
        self.loss = torch.zeros(2, **f)            # [photometric mean, regulariser sum of means]
        self.g_loss = torch.ones(2, **f)           # upstream gradients of the two scalars
        self.g_depth_tgt = torch.empty(B, 1, H, W, **f)
        self.g_depth_src = torch.empty(B, 1, H, W, **f)
        self.ws = torch.empty(L.load().e2e_warp_photo_workspace_floats(B, H, W), **f)

    def bind(self, depth_tgt, depth_src, init_tgt, init_src, src, tgt, K, inv_K, T):
        """Fix the input tensors (they may be rewritten in place between steps)."""
        B, H, W = self.B, self.H, self.W
        for n, t, shp in (("depth_tgt", depth_tgt, (B, 1, H, W)), ("src", src, (B, 3, H, W)), ("tgt", tgt, (B, 3, H, W)),
                          ("K", K, (B, 4, 4)), ("inv_K", inv_K, (B, 4, 4)), ("T", T, (B, 4, 4))):
            L.dev(t, n)
            if tuple(t.shape) != shp:
                raise ValueError(f"{n}: expected {shp}, got {tuple(t.shape)}")
        if self.reg:
            for n, t in (("depth_src", depth_src), ("init_tgt", init_tgt), ("init_src", init_src)):
                L.dev(t, n)
                if tuple(t.shape) != (B, 1, H, W) or not t.is_contiguous():
                    raise ValueError(f"{n}: expected contiguous {(B, 1, H, W)}")
        if not (depth_tgt.is_contiguous() and K.is_contiguous() and inv_K.is_contiguous() and T.is_contiguous()):
            raise ValueError("depth_tgt / K / inv_K / T must be contiguous")
        self.t = (depth_tgt, depth_src, init_tgt, init_src, src, tgt, K, inv_K, T)
        return self

    def forward(self, pmap=None):
        dt, ds, it, is_, src, tgt, K, iK, T = self.t
        L.call("e2e_warp_photo_fwd", L.ptr(dt), L.ptr(src), L.strides4(src), L.ptr(tgt), L.strides4(tgt), L.ptr(K), L.ptr(iK),
               L.ptr(T), L.ptr(self.synth), L.ptr(self.valid), L.ptr(pmap), self.use_mask, self.pad, self.reg,
               L.ptr(it) if self.reg else None, L.ptr(is_) if self.reg else None, L.ptr(ds) if self.reg else None,
               L.ptr(self.loss), L.ptr(self.ws), self.B, self.H, self.W, L.stream())
        return self.loss

    def backward(self):
        dt, ds, it, is_, src, tgt, K, iK, T = self.t
        L.call("e2e_warp_photo_bwd", L.ptr(dt), L.ptr(src), L.strides4(src), L.ptr(tgt), L.strides4(tgt), L.ptr(K), L.ptr(iK),
               L.ptr(T), L.ptr(self.synth), L.ptr(self.valid), self.use_mask, self.pad, self.reg,
               L.ptr(it) if self.reg else None, L.ptr(is_) if self.reg else None, L.ptr(ds) if self.reg else None,
               L.ptr(self.g_loss), L.ptr(self.g_depth_tgt), L.ptr(self.g_depth_src) if self.reg else None,
               self.B, self.H, self.W, L.stream())
        return self.g_depth_tgt, self.g_depth_src


class LossGradPlan(WarpPhotoPlan):
    """One launch per step: e2e_warp_photo_lossgrad (loss and d/d depth together)."""

    def __init__(self, B, H, W, device, padding_mode="border", use_mask=True, reg_kind="l2", w_photo=1.0, w_reg=1e-2):
        super().__init__(B, H, W, device, padding_mode, use_mask, reg_kind)
        self.w_photo, self.w_reg = float(w_photo), float(w_reg)
        # zero-initialised ONCE: the tail of the workspace is the arrival ticket, which the kernel re-arms itself
        self.ws = torch.zeros(L.load().e2e_warp_photo_lossgrad_workspace_floats(B, H, W), device=device, dtype=torch.float32)

    def set_host_geometry(self, K, inv_K, T):
        """Give the pair's geometry as HOST (CPU) 4x4 tensors: the 12 numbers of c = d * M [x,y,1] + p4 then travel as
        kernel arguments (B = 1).  Call again whenever the pair changes; None switches back to the device matrices."""
        if K is None:
            self._geo = None
            return self
        if self.B != 1:
            raise ValueError("host geometry is per keyframe pair (B = 1)")
        K, inv_K, T = (t.detach().double().cpu().reshape(4, 4) for t in (K, inv_K, T))
        P = (K @ T)[:3]
        M = P[:, :3] @ inv_K[:3, :3]
        self._geo = (ctypes.c_float * 12)(*[float(v) for v in M.reshape(-1)], *[float(v) for v in P[:, 3]])
        return self

    def step_chain(self, set_cur, set_prev=-1, loss_prev=None):
        """One kernel for the whole step: the loss sums go to slot set `set_cur` (0..7) and the previous chained launch's
        set `set_prev` is finalised into `loss_prev` (a 2-float tensor).  Finish the last step with flush_chain().
        -> (g_depth_tgt, g_depth_src)."""
        dt, ds, it, is_, src, tgt, K, iK, T = self.t
        geo = getattr(self, "_geo", None)
        L.call("e2e_warp_photo_lossgrad_chain", L.ptr(dt), L.ptr(src), L.strides4(src), L.ptr(tgt), L.strides4(tgt),
               None if geo is not None else L.ptr(K), None if geo is not None else L.ptr(iK), None if geo is not None else L.ptr(T),
               ctypes.cast(geo, ctypes.c_void_p) if geo is not None else None, self.use_mask, self.pad, self.reg,
               L.ptr(it) if self.reg else None, L.ptr(is_) if self.reg else None, L.ptr(ds) if self.reg else None, self.w_photo, self.w_reg,
               int(set_cur), int(set_prev), L.ptr(loss_prev) if set_prev >= 0 else None, L.ptr(self.g_depth_tgt),
               L.ptr(self.g_depth_src) if self.reg else None, L.ptr(self.ws), self.B, self.H, self.W, L.stream())
        return self.g_depth_tgt, self.g_depth_src

    def flush_chain(self, set_last, loss_out):
        L.call("e2e_warp_photo_lossgrad_chain_flush", L.ptr(self.ws), int(set_last), self.reg, L.ptr(loss_out), self.B, self.H, self.W, L.stream())
        return loss_out

    def step(self, want_loss=True):
        """-> (loss[2], g_depth_tgt, g_depth_src); gradients are of w_photo*loss[0] + w_reg*loss[1].
        want_loss=False launches the main kernel only (no second-stage reduction of the loss)."""
        dt, ds, it, is_, src, tgt, K, iK, T = self.t
        loss_ptr = L.ptr(self.loss) if want_loss else None
        if getattr(self, "_geo", None) is not None:
            L.call("e2e_warp_photo_lossgrad_hostgeo", L.ptr(dt), L.ptr(src), L.strides4(src), L.ptr(tgt), L.strides4(tgt),
                   ctypes.cast(self._geo, ctypes.c_void_p), self.use_mask, self.pad, self.reg, L.ptr(it) if self.reg else None,
                   L.ptr(is_) if self.reg else None, L.ptr(ds) if self.reg else None, self.w_photo, self.w_reg, loss_ptr,
                   L.ptr(self.g_depth_tgt), L.ptr(self.g_depth_src) if self.reg else None, L.ptr(self.ws), self.H, self.W, L.stream())
            return self.loss, self.g_depth_tgt, self.g_depth_src
        L.call("e2e_warp_photo_lossgrad", L.ptr(dt), L.ptr(src), L.strides4(src), L.ptr(tgt), L.strides4(tgt), L.ptr(K), L.ptr(iK),
               L.ptr(T), self.use_mask, self.pad, self.reg, L.ptr(it) if self.reg else None, L.ptr(is_) if self.reg else None,
               L.ptr(ds) if self.reg else None, self.w_photo, self.w_reg, loss_ptr, L.ptr(self.g_depth_tgt),
               L.ptr(self.g_depth_src) if self.reg else None, L.ptr(self.ws), self.B, self.H, self.W, L.stream())
        return self.loss, self.g_depth_tgt, self.g_depth_src
