"""Drivers D4/D6 of the reference reduced to their optimisation cores, on the fused kernels:

* `refine_depth_tensor` -- "OFT", train_depth_OFT.py:279-282: the depth map itself is the parameter
  (`define_optim(args, [inputs[("depth", 1, 0)]])`), refined with the same warp + photometric (+ regulariser) loss.
* `learn_depth_scale`   -- absolute_scale.py:207-240: a `ScaleLayer` (one scalar) or `Conv1x1(1, 1, bias=True)` (scale +
  offset) on top of a fixed depth prediction, trained through the same loss.

Both run one `e2e_warp_photo_lossgrad` launch + one fused Adam launch per step; nothing else touches the GPU."""
import torch

from . import _lib as L
from .fused import LossGradPlan
from .optim import FusedAdam


def _plan(depth, depth_src, init_tgt, init_src, src, tgt, K, T, padding_mode, use_mask, reg_kind, w_reg):
    B, _, H, W = depth.shape
    plan = LossGradPlan(B, H, W, depth.device, padding_mode, use_mask, reg_kind, 1.0, w_reg)
    plan.bind(depth, depth_src, init_tgt, init_src, src, tgt, K.contiguous(), torch.linalg.pinv(K).contiguous(), T.contiguous())
    return plan


def refine_depth_tensor(depth_tgt, src, tgt, K, T, steps=3, lr=1e-3, padding_mode="border", use_mask=True,
                        depth_src=None, reg_kind=None, w_reg=1e-2):
    """depth_tgt (B,1,H,W) is refined IN A COPY by Adam on the photometric loss of warping `src` (B,3,H,W view) into the
    target view with depth_tgt, K (B,4,4), T (B,4,4).  With reg_kind "l1"/"l2" and depth_src, the regulariser of
    online_adaption.py:612-623 pulls both maps to their initial values.  Returns (refined depth, [loss per step])."""
    d = L.dev(depth_tgt, "depth_tgt").detach().clone().contiguous()
    p = torch.nn.Parameter(d)
    if reg_kind is not None and depth_src is None:
        raise ValueError("the depth regulariser needs depth_src")
    ds = L.dev(depth_src, "depth_src").detach().clone().contiguous() if reg_kind else None
    it, is_ = (d.clone(), ds.clone()) if reg_kind else (None, None)
    opt = FusedAdam([p], lr=lr)
    trace = []
    plan = None
    for _ in range(steps):
        if plan is None or plan.t[0].data_ptr() != p.data.data_ptr():   # FusedAdam re-homes the parameter at its first step
            plan = _plan(p.data, ds, it, is_, src, tgt, K, T, padding_mode, use_mask, reg_kind, w_reg)
        loss, g_tgt, _ = plan.step()
        trace.append(loss.clone())
        opt.zero_grad()
        if p.grad is None:
            p.grad = g_tgt.clone()
        else:
            p.grad.copy_(g_tgt)
        opt.step()
    return p.data.clone(), [float(v[0]) for v in torch.stack(trace).cpu()]


class ScaleLayer(torch.nn.Module):
    """depth_estimation/networks.py:207-215."""

    def __init__(self, init_value=0.5):
        super().__init__()
        self.scale = torch.nn.Parameter(torch.tensor([init_value]))

    def forward(self, x):
        return x * self.scale


def learn_depth_scale(depth_pred, src, tgt, K, T, steps=50, lr=1e-2, init_value=0.5, affine=False, padding_mode="border", use_mask=True,
                      init_bias=0.0):
    """Learn depth = w * depth_pred (+ b) through the photometric loss (absolute_scale.py:207-240; `affine` = the
    Conv1x1(1, 1, bias=True) variant, whose bias torch initialises at random: pass it as init_bias).  The depth network is frozen
    (its parameters are not in train_params, :223-224), so its prediction is an input.  Returns (w, b, [loss per step]).
    d loss / d w = <g, depth_pred>, d loss / d b = sum(g) with g = the fused kernel's d loss / d depth."""
    base = L.dev(depth_pred, "depth_pred").detach().contiguous()
    w = torch.nn.Parameter(torch.full((1,), float(init_value), device=base.device))
    b = torch.nn.Parameter(torch.full((1,), float(init_bias), device=base.device))
    params = [w, b] if affine else [w]
    opt = FusedAdam(params, lr=lr)
    cur = torch.empty_like(base)
    plan = _plan(cur, None, None, None, src, tgt, K, T, padding_mode, use_mask, None, 0.0)
    trace = []
    for _ in range(steps):
        torch.add(b.data if affine else torch.zeros_like(b.data), base * w.data, out=cur)
        loss, g, _ = plan.step()
        trace.append(loss.clone())
        opt.zero_grad()
        gw, gb = (g * base).sum().reshape(1), g.sum().reshape(1)
        for prm, gr in ((w, gw), (b, gb)) if affine else ((w, gw),):
            if prm.grad is None:
                prm.grad = gr.clone()
            else:
                prm.grad.copy_(gr)
        opt.step()
    return float(w.data), float(b.data), [float(v[0]) for v in torch.stack(trace).cpu()]


def scale_grid_search(depth_pred, src, tgt, K, T, grid, steps=3, lr=1e-2, affine=True, padding_mode="border", use_mask=True, init_bias=0.0):
    """The outer loop of absolute_scale.py:268-405: for every initial value of SCALE_GRID_SEARCH.grid a FRESH scale layer and a fresh
    optimiser (`scale_layer_init(init_depth_scale)`, :207-240, :273) are trained for OPTIMIZATION.refinement_steps steps on the same
    batch.  Returns one record per grid value: {"init", "w", "b", "losses"} -- what the reference prints as "Trainable param after
    training" for each experiment."""
    out = []
    for init in grid:
        w, b, trace = learn_depth_scale(depth_pred, src, tgt, K, T, steps=steps, lr=lr, init_value=float(init), affine=affine,
                                        padding_mode=padding_mode, use_mask=use_mask, init_bias=init_bias)
        out.append({"init": float(init), "w": w, "b": b, "losses": trace})
    return out
