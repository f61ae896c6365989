"""Native convolutions of the depth network (csrc/conv.hip): fp32-MFMA implicit GEMM with the gather
(stride, zero / reflection padding, nearest x2 upsample + channel concat) and the epilogue (folded eval-BN /
bias, residual add, ReLU / ELU / disparity head) fused.  Tensors are NCHW-shaped, channels_last in memory."""
import ctypes

import torch
from torch.autograd.function import once_differentiable

from . import _lib as L

ACT = {None: 0, "relu": 1, "elu": 2, "disp": 3}
CL = torch.channels_last

# ---- state -------------------------------------------------------------------------------------------------------------------
# This module keeps NO process-wide state.  Everything that outlives a call hangs on the objects it describes:
#   * a weight's k-major GEMM copies: `w._e2e_layouts` on the Parameter itself (dies with it);
#   * the set of weights refreshed together: a LayoutGroup owned by ONE model (group_layouts), holding its members strongly;
#   * "an optimiser rewrote these parameters behind torch's version counters": a counter cell owned by that optimiser's
#     FlatParams and shared by its parameters (`p._e2e_epoch`);
#   * "backward-weight kernels may write straight into the optimiser's flat bucket", and the side stream those chains use:
#     fields of that FlatParams (`p._e2e_grad_owner`), switched on for the extent of ONE backward by direct_weight_grads();
#   * a forced GEMM decomposition (tests, tools): the `tuning` argument of conv2d().
# Round 3 kept a module-level registry of every weight of every model ever built and refreshed all of them in one launch from a
# table of raw device pointers; a dead model collected by the garbage collector between the pointer walk and the launch left
# the kernel writing into blocks the allocator had already handed out again ("the order-dependent test failures").


def epoch_of(p):
    """How often an optimiser has rewritten parameter p through a raw pointer (0: never / torch optimisers, whose in-place
    updates move p._version instead)."""
    cell = getattr(p, "_e2e_epoch", None)
    return cell[0] if cell is not None else 0


def epoch_sum(params):
    """Monotone stamp over the (distinct) epoch cells of several parameters."""
    cells = {}
    for p in params:
        c = getattr(p, "_e2e_epoch", None)
        if c is not None:
            cells[id(c)] = c
    return sum(c[0] for c in cells.values())


class direct_weight_grads:
    """with direct_weight_grads(optimizer, overlap): inside, the backward-weight kernels ACCUMULATE straight into the parameters'
    slices of the optimiser's flat gradient bucket and autograd gets no gradient for them (one AccumulateGrad add kernel less per
    parameter and step: 48 at 5 us).  The switch is a field of THAT optimiser's FlatParams -- another model's backward running
    meanwhile is not affected -- and it is off outside the block: torch.autograd.grad() callers get their gradients returned.
    An optimiser without a flat bucket (torch's own, or FusedAdam before its first step): a no-op.
    overlap=True additionally moves every backward-weight chain (GEMM + slab folds + reduce) to the FlatParams' side stream: it
    depends only on dZ, like the backward-data GEMM next to it.  The exit joins the streams."""

    def __init__(self, optimizer, overlap=False):
        self.flat = getattr(optimizer, "flat", None)
        self.overlap = bool(overlap)

    def __enter__(self):
        f = self.flat
        if f is not None:
            self.prev = (f.direct, f.overlap)
            f.direct, f.overlap = True, self.overlap
        return self

    def __exit__(self, *exc):
        f = self.flat
        if f is not None:
            f.direct, f.overlap = self.prev
            if f.side is not None:
                torch.cuda.current_stream(f.side.device).wait_stream(f.side)


def _sink(param, shape):
    """The parameter's slice of its optimiser's flat gradient bucket while direct_weight_grads() is active for that optimiser."""
    if param is None:
        return None
    owner = getattr(param, "_e2e_grad_owner", None)
    if owner is None or not owner.direct:
        return None
    t = getattr(param, "_e2e_grad_sink", None)
    if t is None or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.dtype != torch.float32:
        return None
    return t


def _overlap_stream(param):
    """Side stream for the backward-weight chain of `param` (None: same stream)."""
    owner = getattr(param, "_e2e_grad_owner", None)
    if owner is None or not (owner.direct and owner.overlap):
        return None
    return owner.side_stream()


class LayoutGroup:
    """The convolution weights of ONE model whose GEMM layouts are refreshed together: after an optimiser step every one is stale,
    and one launch over a descriptor table replaces ~40.  The group holds its members strongly and is owned by the model
    (weight -> group -> weights is an ordinary reference cycle inside the model): every pointer that goes into a table belongs to
    a tensor that is alive for as long as the caller -- a forward of that model -- runs."""

    def __init__(self):
        self.members = []
        self._sig = self._desc = None

    def add(self, w):
        old = getattr(w, "_e2e_group", None)
        if old is self:
            return
        if old is not None:
            old.members = [m for m in old.members if m is not w]
            old._sig = old._desc = None
        w._e2e_group = self
        self.members.append(w)


def group_layouts(module):
    """Put every MFMA-path convolution weight of `module` into one LayoutGroup (the outermost model that calls this wins)."""
    g = LayoutGroup()
    for m in module.modules():
        if isinstance(m, torch.nn.Conv2d) and m.weight.shape[0] % 16 == 0:
            g.add(m.weight)
    return g


def _layout_key(w):
    return (w.data_ptr(), w._version, epoch_of(w))


def _weight_layouts(w, want_bwd):
    """k-major GEMM copies of a weight.  The entry lives ON the weight tensor object (a recycled allocation of another
    tensor can never alias it) and is valid while (data_ptr, torch version counter, optimiser epoch) are unchanged.
    The buffers are allocated once and rewritten in place; when an entry is stale, the stale weights of the SAME LayoutGroup
    (= the same model) are refreshed in the same launch."""
    ent = getattr(w, "_e2e_layouts", None)
    if ent is not None and ent["key"] == _layout_key(w) and ent["wf"].device == w.device and (ent["wb"] is not None or not want_bwd):
        return ent["wf"], ent["wb"]
    Cout, Cin, KH, KW = w.shape
    ldf, ldb = _ld(Cout), _ld(Cin)
    if ent is None or ent["wf"].device != w.device:
        ent = {"key": None, "wf": torch.zeros(KH * KW * Cin, ldf, device=w.device, dtype=torch.float32), "wb": None}
        try:
            w._e2e_layouts = ent
        except AttributeError:                      # cannot attach (exotic tensor subclass): uncached single refresh
            wb = torch.zeros(KH * KW * Cout, ldb, device=w.device, dtype=torch.float32) if want_bwd else None
            L.call("e2e_conv_weight_layouts", L.ptr(w), Cout, Cin, KH, KW, L.ptr(ent["wf"]), ldf, L.ptr(wb), ldb, L.stream())
            return ent["wf"], wb
    if want_bwd and ent["wb"] is None:
        ent["wb"] = torch.zeros(KH * KW * Cout, ldb, device=w.device, dtype=torch.float32)
        ent["key"] = None
    grp = getattr(w, "_e2e_group", None)
    _refresh_stale(w, grp)
    return ent["wf"], ent["wb"]


def _refresh_stale(w, grp):
    """Rewrite the layouts of `w` and of the stale members of its group.  `todo` keeps every tensor whose pointer goes into the
    table strongly referenced until after the launch (stream order covers everything later)."""
    cand = [w] if grp is None else ([w] + [m for m in grp.members if m is not w])
    todo = []
    for t in cand:
        ent = getattr(t, "_e2e_layouts", None)
        if ent is None or t.device != w.device or ent["wf"].device != t.device or ent["key"] == _layout_key(t):
            continue
        todo.append((t, ent, ent["wf"], ent["wb"]))
    if not todo:
        return
    if len(todo) == 1:
        t, ent, wf, wb = todo[0]
        Cout, Cin, KH, KW = t.shape
        L.call("e2e_conv_weight_layouts", L.ptr(t), Cout, Cin, KH, KW, L.ptr(wf), _ld(Cout), L.ptr(wb), _ld(Cin), L.stream())
        ent["key"] = _layout_key(t)
        return
    rows = []
    for t, ent, wf, wb in todo:
        Cout, Cin, KH, KW = t.shape
        rows.append([t.data_ptr(), wf.data_ptr(), wb.data_ptr() if wb is not None else 0, Cout, Cin, KH, KW, _ld(Cout), _ld(Cin), 0])
    sig = tuple(tuple(r) for r in rows)
    if grp._sig != sig:
        grp._desc = torch.tensor(rows, dtype=torch.int64).to(w.device)
        grp._sig = sig
    L.call("e2e_conv_weight_layouts_batched", L.ptr(grp._desc), len(rows), L.stream())
    for t, ent, _, _ in todo:
        ent["key"] = _layout_key(t)
    del todo


def available():
    try:
        return hasattr(L.load(), "e2e_conv2d_fwd")
    except Exception:
        return False


def supports(weight):
    """Layer shapes the native kernels cover: Cout a multiple of 16 (MFMA path), the 16 -> 1 3x3 disparity head and the
    1 -> 1 1x1 scale layer (Conv1x1(1, 1) of the scale-learning experiments)."""
    return weight.shape[0] % 16 == 0 or tuple(weight.shape) in ((1, 16, 3, 3), (1, 1, 1, 1))


def _grad_out(param, shape):
    """(buffer, accumulate): the parameter's slice of FusedAdam's flat gradient buffer inside direct_weight_grads(),
    else a fresh tensor that is returned to autograd."""
    s = _sink(param, shape)
    return (s, 1) if s is not None else (torch.empty(shape, device=param.device, dtype=torch.float32), 0)


class _Affine(torch.autograd.Function):
    """y = relu?( z * scale[c] + shift[c] + residual ) on NHWC memory, with the gradients of the parameters the scale / shift
    come from.  BN form (mean given): scale = gamma / sqrt(var + eps), shift = beta - mean * scale (an eval-mode BatchNorm
    whose affine still trains: online_adaption.py:182-184 leaves `downsample.1` trainable; a network that was not put in
    refinement mode trains every BatchNorm's affine).  Plain form: C = 1, scale = the (1,1,1,1) weight and shift = the bias
    of Conv1x1(1, 1) / ScaleLayer (networks.py:191-215)."""

    @staticmethod
    def forward(ctx, z, p_scale, p_shift, mean, var, eps, residual, relu):
        z = L.dev(z, "input")
        C = z.shape[1]
        z = _cl(z)
        dev, st = z.device, L.stream()
        if mean is not None:
            scale, shift, rstd = (torch.empty(C, device=dev, dtype=torch.float32) for _ in range(3))
            L.call("e2e_bn_fold", L.ptr(p_scale), L.ptr(p_shift), L.ptr(mean), L.ptr(var), float(eps), L.ptr(scale), L.ptr(shift), L.ptr(rstd), C, st)
        else:
            if C != 1:
                raise NotImplementedError("the plain affine form is the single-channel scale layer")
            scale, shift, rstd = p_scale.reshape(1), (p_shift.reshape(1) if p_shift is not None else None), None
        if residual is not None:
            residual = _cl(L.dev(residual, "residual"))
        y = torch.empty_like(z)
        L.call("e2e_affine_fwd", L.ptr(z), L.ptr(scale), L.ptr(shift), L.ptr(residual), int(bool(relu)), L.ptr(y), z.numel(), C, st)
        ctx.save_for_backward(z, scale, mean, rstd, y if relu else None)
        ctx.params = (p_scale, p_shift)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        z, scale, mean, rstd, y = ctx.saved_tensors
        p_scale, p_shift = ctx.params
        C = z.shape[1]
        g = _cl(g)
        st = L.stream()
        if y is not None:                            # dA = g * relu'(y)
            dA = torch.empty_like(g)
            L.call("e2e_conv2d_act_bwd", L.ptr(g), L.ptr(y), None, L.ptr(dA), g.numel(), C, ACT["relu"], st)
            g = dA
        gs = gb = None
        want_s, want_b = ctx.needs_input_grad[1], (p_shift is not None and ctx.needs_input_grad[2])
        if want_s or want_b:
            ss = _sink(p_scale, p_scale.shape) if want_s else None
            sb = _sink(p_shift, p_shift.shape) if want_b else None
            direct = (ss is not None or not want_s) and (sb is not None or not want_b)
            gs = (ss if direct else torch.empty(p_scale.shape, device=g.device, dtype=torch.float32)) if want_s else None
            gb = (sb if direct else torch.empty(p_shift.shape, device=g.device, dtype=torch.float32)) if want_b else None
            ws = torch.empty(L.load().e2e_affine_bwd_workspace_floats(C), device=g.device, dtype=torch.float32)
            L.call("e2e_affine_bwd", L.ptr(g), L.ptr(z), L.ptr(mean), L.ptr(rstd), z.numel() // C, C, L.ptr(gs), L.ptr(gb), 1 if direct else 0,
                   L.ptr(ws), st)
            if direct:
                gs = gb = None
        dz = None
        if ctx.needs_input_grad[0]:
            dz = torch.empty_like(g)
            L.call("e2e_conv2d_act_bwd", L.ptr(g), L.ptr(g), L.ptr(scale), L.ptr(dz), g.numel(), C, 0, st)
        return dz, gs, gb, None, None, None, (g if ctx.has_res else None), None


def conv2d_bn_affine(x, weight, bn, stride, padding, pad_mode, residual=None, relu=False, in_norm=None):
    """relu?( BN_eval_with_trainable_affine(conv(x)) + residual ): the convolution writes its raw output, the affine is its own
    small kernel so that gamma / beta get exact gradients from the stored convolution output.  On the refinement path these
    are the three tiny 1x1 stride-2 `downsample` branches of ResNet-18 (every other BatchNorm is frozen and folded)."""
    w, b, rm, rv, eps = bn
    z = conv2d(x, weight, None, stride, padding, pad_mode, in_norm=in_norm)
    return _Affine.apply(z, w, b, rm, rv, eps, residual, bool(relu))


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _cl(L.dev(x, "input"))
        B, C, H, W = x.shape
        y = torch.empty(B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device=x.device, dtype=torch.float32, memory_format=CL)
        L.call("e2e_maxpool3x3s2_fwd", L.ptr(x), L.ptr(y), B, H, W, C, L.stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        g = _cl(g)
        dx = torch.empty_like(x)
        L.call("e2e_maxpool3x3s2_bwd", L.ptr(x), L.ptr(g), L.ptr(dx), B, H, W, C, 0, 0, L.stream())
        return dx


def max_pool_3x3_s2(x):
    return _MaxPool.apply(x)


class _Upsample2Concat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, skip):
        x = _cl(L.dev(x, "input"))
        B, C1, h, w = x.shape
        C2 = 0
        if skip is not None:
            skip = _cl(L.dev(skip, "skip"))
            C2 = skip.shape[1]
            if tuple(skip.shape) != (B, C2, 2 * h, 2 * w):
                raise ValueError(f"skip tensor: expected {(B, C2, 2 * h, 2 * w)}, got {tuple(skip.shape)}")
        y = torch.empty(B, C1 + C2, 2 * h, 2 * w, device=x.device, dtype=torch.float32, memory_format=CL)
        L.call("e2e_upsample2_concat", L.ptr(x), L.ptr(skip), L.ptr(y), B, h, w, C1, C2, L.stream())
        ctx.dims = (B, h, w, C1, C2)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        B, h, w, C1, C2 = ctx.dims
        g = _cl(g)
        g0 = torch.empty(B, C1, h, w, device=g.device, dtype=torch.float32, memory_format=CL)
        g1 = torch.empty(B, C2, 2 * h, 2 * w, device=g.device, dtype=torch.float32, memory_format=CL) if C2 else None
        L.call("e2e_conv2d_gather_adjoint", L.ptr(g), B, 2 * h, 2 * w, C1 + C2, C1, 2, 0, L.ptr(g0), L.ptr(g1), 0, 0, L.stream())
        return g0, g1


def upsample2_concat(x, skip=None):
    return _Upsample2Concat.apply(x, skip)


class _Head(torch.autograd.Function):
    """Conv3x3(reflect) 16 -> 1 + activation: 144-tap dot product per pixel (HBM-bound VALU kernels)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = _cl(L.dev(x, "input"))
        B, C, H, W = x.shape
        w = L.dev(weight, "weight").contiguous()
        y = torch.empty(B, 1, H, W, device=x.device, dtype=torch.float32)
        L.call("e2e_head_fwd", L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), B, H, W, C, act, L.stream())
        ctx.save_for_backward(x, w, y)
        ctx.cfg = (act, bias is not None)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        act, has_bias = ctx.cfg
        B, C, H, W = x.shape
        st = L.stream()
        g = g.contiguous()
        dz = g
        if act:
            dz = torch.empty_like(g)
            L.call("e2e_conv2d_act_bwd", L.ptr(g), L.ptr(y), None, L.ptr(dz), g.numel(), 1, act, st)
        dx = torch.empty(B, C, H, W, device=g.device, dtype=torch.float32, memory_format=CL) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        db = torch.empty(1, device=g.device, dtype=torch.float32) if (has_bias and ctx.needs_input_grad[2]) else None
        ws = torch.empty(L.load().e2e_head_workspace_floats(), device=g.device, dtype=torch.float32)
        L.call("e2e_head_bwd", L.ptr(dz), L.ptr(x), L.ptr(w), L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(ws), B, H, W, C, st)
        sw = _sink(ctx.params[0], w.shape) if dw is not None else None
        sb = _sink(ctx.params[1], (1,)) if db is not None else None
        if sw is not None:                          # tiny tensors (145 numbers): add in place, return nothing
            sw.add_(dw)
            dw = None
        if sb is not None:
            sb.add_(db)
            db = None
        return dx, dw, db, None


def _cl(t):
    return t if t.is_contiguous(memory_format=CL) else t.contiguous(memory_format=CL)


def _ld(n):
    return (n + 3) // 4 * 4


def _gemm_workspace(n, dev):
    """A convolution GEMM workspace: its head (e2e_conv_workspace_flag_floats) holds the stream-K flags, zero outside a launch."""
    if not n:
        return None
    ws = torch.empty(n, device=dev, dtype=torch.float32)
    ws[: L.load().e2e_conv_workspace_flag_floats()].zero_()
    return ws


def check_streamk(ws):
    """Raises if a stream-K launch that used this workspace timed out waiting for a partial tile (host sync)."""
    if ws is not None and int(ws.view(torch.int32)[L.load().e2e_conv_streamk_error_index()]) != 0:
        raise L.E2EError("stream-K convolution: a workgroup timed out waiting for a partial tile")


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src0, src1, weight, bias, scale, shift, residual, up, stride, pad, pad_mode, act, in_norm, wf, wb, tune):
        ctx.params = (weight, bias)
        Cout, Cin, KH, KW = weight.shape
        src0 = _cl(L.dev(src0, "input"))
        B, C1 = src0.shape[0], src0.shape[1]
        Hs, Ws = src0.shape[2] * up, src0.shape[3] * up
        if src1 is not None:
            src1 = _cl(L.dev(src1, "skip"))
            if tuple(src1.shape) != (B, Cin - C1, Hs, Ws):
                raise ValueError(f"skip tensor: expected {(B, Cin - C1, Hs, Ws)}, got {tuple(src1.shape)}")
        elif C1 != Cin:
            raise ValueError(f"input has {C1} channels, the convolution expects {Cin}")
        Ho, Wo = (Hs + 2 * pad - KH) // stride + 1, (Ws + 2 * pad - KW) // stride + 1
        dev = src0.device
        w = L.dev(weight, "weight")
        if not w.is_contiguous():
            raise ValueError("convolution weights must be contiguous (Cout,Cin,KH,KW)")
        ldf, ldb = _ld(Cout), _ld(Cin)
        if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and wb is None:
            wb = _weight_layouts(w, True)[1]
        # epilogue vectors: y = scale * conv + shift ; a plain bias is shift with unit scale
        sh = shift
        if bias is not None:
            sh = bias if shift is None else shift       # BN-folded callers pass shift only
        if residual is not None:
            residual = _cl(L.dev(residual, "residual"))
        out = torch.empty(B, Cout, Ho, Wo, device=dev, dtype=torch.float32, memory_format=CL)
        isub, imul = in_norm if in_norm is not None else (0.0, 1.0)
        if tune is not None:
            ws = _gemm_workspace(L.load().e2e_conv_tuned_workspace_floats(B * Ho * Wo, Cout), dev)
            L.call("e2e_conv2d_fwd_tuned", L.ptr(src0), L.ptr(src1), C1, up, L.ptr(wf), ldf, L.ptr(scale), L.ptr(sh), L.ptr(residual), L.ptr(out),
                   B, Hs, Ws, Cin, Cout, KH, KW, stride, pad, pad_mode, act, float(isub), float(imul), L.ptr(ws), tune[0], tune[1], tune[2], L.stream())
            if tune[2] < 0:
                check_streamk(ws)
        else:
            ws = _gemm_workspace(L.load().e2e_conv2d_splitk_workspace_floats(B * Ho * Wo, Cout, KH * KW * Cin), dev)
            L.call("e2e_conv2d_fwd", L.ptr(src0), L.ptr(src1), C1, up, L.ptr(wf), ldf, L.ptr(scale), L.ptr(sh), L.ptr(residual), L.ptr(out),
                   B, Hs, Ws, Cin, Cout, KH, KW, stride, pad, pad_mode, act, float(isub), float(imul), L.ptr(ws), L.stream())
        ctx.tune = tune
        ctx.save_for_backward(src0, src1, wb, scale, out)
        ctx.cfg = (B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode, act, up, C1, ldb, float(isub), float(imul),
                   bias is not None, residual is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        src0, src1, wb, scale, out = ctx.saved_tensors
        (B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode, act, up, C1, ldb, isub, imul, has_bias, has_res) = ctx.cfg
        dev = g.device
        g = _cl(g)
        st = L.stream()
        n = g.numel()
        # dA = dY * act'(Y) (gradient of the residual branch), dZ = dA * scale (gradient of the convolution output)
        d_res = None
        if act != 0 or has_res:
            dA = torch.empty_like(g) if act != 0 else g
            if act != 0:
                L.call("e2e_conv2d_act_bwd", L.ptr(g), L.ptr(out), None, L.ptr(dA), n, Cout, act, st)
            d_res = dA if has_res else None
        else:
            dA = g
        if scale is not None:
            dZ = torch.empty_like(g)
            L.call("e2e_conv2d_act_bwd", L.ptr(dA), L.ptr(out), L.ptr(scale), L.ptr(dZ), n, Cout, 0, st)
        else:
            dZ = dA
        needs = ctx.needs_input_grad
        g0 = g1 = gw = gb = None
        if needs[0] or needs[1]:
            pp = pad if pad_mode == 1 else 0
            direct = pp == 0 and up == 1 and src1 is None
            dxp = torch.empty(B, Cin, Hs + 2 * pp, Ws + 2 * pp, device=dev, dtype=torch.float32, memory_format=CL)
            tune = ctx.tune
            if tune is not None:
                ws2 = _gemm_workspace(L.load().e2e_conv_tuned_workspace_floats(B * (Hs + 2 * pp) * (Ws + 2 * pp), Cin), dev)
                L.call("e2e_conv2d_bwd_data_fused_tuned", L.ptr(dZ), L.ptr(wb), ldb, L.ptr(dxp), B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode,
                       0, None, 0, None, L.ptr(ws2), tune[0], tune[1], tune[2], st)
                if tune[2] < 0:
                    check_streamk(ws2)
            else:
                ws2 = _gemm_workspace(L.load().e2e_conv2d_bwd_data_workspace_floats(B, Hs + 2 * pp, Ws + 2 * pp, Cin, KH * KW * Cout, stride), dev)
                L.call("e2e_conv2d_bwd_data", L.ptr(dZ), L.ptr(wb), ldb, L.ptr(dxp), B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode,
                       L.ptr(ws2), st)
            if direct:
                g0 = dxp
            else:
                g0 = torch.empty(B, C1, Hs // up, Ws // up, device=dev, dtype=torch.float32, memory_format=CL)
                if src1 is not None:
                    g1 = torch.empty(B, Cin - C1, Hs, Ws, device=dev, dtype=torch.float32, memory_format=CL)
                L.call("e2e_conv2d_gather_adjoint", L.ptr(dxp), B, Hs, Ws, Cin, C1, up, 1 if pp else 0, L.ptr(g0), L.ptr(g1), 0, 0, st)
        if needs[2]:
            want_b = has_bias and needs[3]
            sw = _sink(ctx.params[0], (Cout, Cin, KH, KW))
            sb = _sink(ctx.params[1], (Cout,)) if want_b else None
            direct = sw is not None and (sb is not None or not want_b)     # both into the flat gradient buffer, or neither
            gw = sw if direct else torch.empty(Cout, Cin, KH, KW, device=dev, dtype=torch.float32)
            gb = (sb if direct else torch.empty(Cout, device=dev, dtype=torch.float32)) if want_b else None
            ws = torch.empty(L.load().e2e_conv2d_wgrad_workspace_floats(B, Ho, Wo, Cin, Cout, KH, KW, 1 if gb is not None else 0),
                             device=dev, dtype=torch.float32)
            st_w = st
            side = _overlap_stream(ctx.params[0]) if direct else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(dev))          # dZ (and everything before it) is ready
                for t in (dZ, src0, src1, ws):
                    if t is not None:
                        t.record_stream(side)
                st_w = ctypes.c_void_p(side.cuda_stream)
            L.call("e2e_conv2d_bwd_weight", L.ptr(dZ), L.ptr(src0), L.ptr(src1), C1, up, L.ptr(gw), L.ptr(gb), L.ptr(ws), B, Hs, Ws, Cin,
                   Cout, Ho, Wo, KH, KW, stride, pad, pad_mode, 1 if direct else 0, isub, imul, st_w)
            if direct:
                gw = gb = None
        return g0, g1, gw, gb, None, None, d_res, None, None, None, None, None, None, None, None, None


def conv2d(x, weight, bias=None, stride=1, padding=0, pad_mode="zeros", act=None, bn_scale_shift=None, residual=None,
           skip=None, upsample=1, in_norm=None, tuning=None):
    """act( scale * conv(cat(upsample(x, upsample), skip)) + shift (+ residual) ).
    bn_scale_shift: (scale, shift) per output channel (folded eval-mode BatchNorm; constants, no gradient).
    tuning: None (the library's measured rule) or (tile_m, tile_n, ksplit) -- THIS call's forward and backward-data GEMMs run that
    decomposition through the *_tuned entry points (ksplit >= 1: K slices, < 0: stream-K on -ksplit persistent workgroups of
    64 x 64 tiles); tests and tools/gemm_tune.py use it, the library and this module keep no tuning state."""
    if pad_mode not in ("zeros", "reflect"):
        raise ValueError(f"pad_mode {pad_mode}")
    if tuple(weight.shape) == (1, 1, 1, 1):
        if stride != 1 or padding != 0 or skip is not None or upsample != 1 or residual is not None or bn_scale_shift is not None \
                or act is not None or in_norm is not None:
            raise NotImplementedError("the 1 -> 1 convolution is the plain scale layer (Conv1x1(1, 1))")
        return _Affine.apply(x, weight, bias, None, None, 0.0, None, False)
    if weight.shape[0] == 1:
        if tuple(weight.shape) != (1, 16, 3, 3) or pad_mode != "reflect" or padding != 1 or stride != 1 or skip is not None \
                or upsample != 1 or residual is not None or bn_scale_shift is not None:
            raise NotImplementedError("single-output-channel convolution other than the 16->1 reflect 3x3 disparity head")
        return _Head.apply(x, weight, bias, ACT[act])
    scale, shift = bn_scale_shift if bn_scale_shift is not None else (None, None)
    if bias is not None and bn_scale_shift is not None:
        raise ValueError("a bias together with a folded BatchNorm is not used by the network")
    if not weight.is_contiguous():
        raise ValueError("convolution weights must be contiguous (Cout,Cin,KH,KW)")
    want_bwd = torch.is_grad_enabled() and (x.requires_grad or (skip is not None and skip.requires_grad))
    wf, wb = _weight_layouts(weight, want_bwd)          # cached on the Parameter object across forwards
    return _Conv2d.apply(x, skip, weight, bias, scale, shift, residual, int(upsample), int(stride), int(padding),
                         1 if pad_mode == "reflect" else 0, ACT[act], in_norm, wf, wb,
                         None if tuning is None else tuple(int(v) for v in tuning))
