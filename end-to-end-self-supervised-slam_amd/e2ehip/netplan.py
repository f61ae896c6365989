"""Static launch plan of the depth network (DispResNet_Indoor: ResNet BasicBlock encoder + indoor decoder) for one fixed
input shape: every activation, gradient, workspace and weight-layout buffer is allocated ONCE, forward and backward are
fixed sequences of C-ABI launches (no autograd graph, no allocator traffic, no host synchronisation), so a whole
refinement step can be captured into a hipGraph and replayed (`online_adaption.SLAM`).  This is the hot-path form of
depth_estimation/networks.py:44-57 (encoder), :277-292 (decoder) and of `loss.backward()` through them
(online_adaption.py:539); the nn.Module path (per-layer autograd Functions over the same kernels) stays for ad-hoc use
and is the cross-check in tests/test_gpu_netplan.py.

Backward conventions
  * every gradient buffer holds d loss / d PRE-activation of the layer that produced its tensor ("dA"), not d loss / d output:
    whoever contributes to the gradient of a tensor x = act(u) multiplies its contribution by act'(u) -- recovered from x,
    which the contributor reads anyway -- inside its own epilogue (e2e_conv2d_bwd_data_fused, e2e_conv2d_gather_adjoint_act,
    e2e_head_bwd_act, e2e_maxpool3x3s2_bwd's mul_relu).  A layer's backward GEMMs therefore read its gradient buffer as is;
    a folded BatchNorm scale rides in the backward weight layout (dX = dA (scale W)^T) and in the backward-weight slab
    reduction (dW = scale * dA^T X).  The separate dY * act'(Y) [* scale] pass of every layer (round 1 / the autograd
    path: 0.4 ms of a 6.9 ms step) does not exist;
  * every tensor with several consumers (a BasicBlock's input: first convolution + residual add [+ downsample branch]; the
    encoder features that also feed decoder skips; the stem output: max-pool + last skip) owns ONE gradient buffer; the
    first contribution of a backward pass stores, later ones accumulate inside the producing kernel (accumulate flags of
    e2e_conv2d_bwd_data_acc / e2e_conv2d_act_bwd_acc / e2e_conv2d_gather_adjoint / e2e_maxpool3x3s2_bwd) -- no add kernels;
  * weight / bias / BatchNorm-affine gradients are written straight into the optimiser's flat gradient bucket (each
    parameter has exactly one writer per step, so the bucket needs no zero-fill);
  * the backward-weight GEMM of a layer depends only on its dZ, like the backward-data GEMM next to it: with
    `overlap=True` it goes to a second stream (fork / join inside the captured graph).
"""
import ctypes

import torch

from . import _lib as L
from .conv import ACT, _ld, epoch_sum

_f32 = torch.float32


class Buf:
    """An NHWC activation (B,h,w,C) with its gradient buffer."""
    __slots__ = ("t", "g", "written", "B", "h", "w", "C", "act")

    def __init__(self, B, h, w, C, dev, need_grad=True, act=0):
        self.B, self.h, self.w, self.C = B, h, w, C
        self.act = act                                  # activation of the producing layer (ACT code): its derivative is taken from t
        self.t = torch.empty(B, h, w, C, device=dev, dtype=_f32)
        self.g = torch.empty(B, h, w, C, device=dev, dtype=_f32) if need_grad else None
        self.written = False

    def nchw(self):
        return self.t.permute(0, 3, 1, 2)


def _at(t, slot):
    """Pointer to a batch buffer, or to image `slot` of it."""
    return L.ptr(t if slot is None else t[slot])


def _sink_of(p):
    s = getattr(p, "_e2e_grad_sink", None)
    if s is None:                                    # no flat bucket (plain torch optimiser): a private gradient tensor
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        s = p.grad
    return s


class _Conv:
    def __init__(self, plan, src0, src1, weight, bias, bn, residual, act, stride, pad, pad_mode, up, in_norm):
        dev = plan.dev
        self.src0, self.src1, self.res = src0, src1, residual
        self.weight, self.bias = weight, bias
        self.Cout, self.Cin, self.KH, self.KW = weight.shape
        self.C1 = src0.C
        if self.C1 + (src1.C if src1 is not None else 0) != self.Cin:
            raise ValueError("channel mismatch in the launch plan")
        self.up, self.stride, self.pad, self.pm, self.act = up, stride, pad, 1 if pad_mode == "reflect" else 0, ACT[act]
        self.isub, self.imul = in_norm if in_norm is not None else (0.0, 1.0)
        B = src0.B
        self.Hs, self.Ws = src0.h * up, src0.w * up
        self.Ho, self.Wo = (self.Hs + 2 * pad - self.KH) // stride + 1, (self.Ws + 2 * pad - self.KW) // stride + 1
        if self.act not in (0, ACT["relu"], ACT["elu"]):
            raise NotImplementedError("launch plan: convolution epilogues are none / ReLU / ELU (the sigmoid head has its own op)")
        self.out = Buf(B, self.Ho, self.Wo, self.Cout, dev, act=self.act)
        self.ldf, self.ldb = _ld(self.Cout), _ld(self.Cin)
        self.wf = torch.zeros(self.KH * self.KW * self.Cin, self.ldf, device=dev, dtype=_f32)
        self.need_dx = src0.g is not None
        self.wb = torch.zeros(self.KH * self.KW * self.Cout, self.ldb, device=dev, dtype=_f32) if self.need_dx else None
        # epilogue vectors: folded frozen BatchNorm (constants) or the live bias parameter
        self.scale = self.shift = None
        if bn is not None:
            w, b, rm, rv, eps = bn
            with torch.no_grad():
                self.scale = (w / torch.sqrt(rv + eps)).contiguous()
                self.shift = (b - rm * self.scale).contiguous()
        lib = L.load()
        # (a forward of ONE batch slot -- NetPlan.forward_one -- picks its own K split: sized for the larger of the two)
        n = max(lib.e2e_conv2d_splitk_workspace_floats(b * self.Ho * self.Wo, self.Cout, self.KH * self.KW * self.Cin) for b in {B, 1})
        self.ws_f = torch.zeros(n, device=dev, dtype=_f32) if n else None           # zeroed once: its head holds the stream-K hand-off flags
        self.pp = pad if self.pm == 1 else 0
        self.direct = self.pp == 0 and up == 1 and src1 is None
        if self.need_dx:
            n = lib.e2e_conv2d_bwd_data_workspace_floats(B, self.Hs + 2 * self.pp, self.Ws + 2 * self.pp, self.Cin, self.KH * self.KW * self.Cout, stride)
            self.ws_b = torch.zeros(n, device=dev, dtype=_f32) if n else None
            self.dxp = None if self.direct else torch.empty(B, self.Hs + 2 * self.pp, self.Ws + 2 * self.pp, self.Cin, device=dev, dtype=_f32)
        self.ws_w = torch.empty(lib.e2e_conv2d_wgrad_workspace_floats(B, self.Ho, self.Wo, self.Cin, self.Cout, self.KH, self.KW,
                                                                      1 if bias is not None else 0), device=dev, dtype=_f32)
        # residual wiring of a BasicBlock (NetPlan.__init__): `res_via` = the block's first convolution, whose backward-data launch
        # adds this layer's gradient to the block input (pre_add) -- no separate accumulate pass; `res_aliased`: the residual tensor
        # has no activation and no other consumer (downsample branch), its gradient buffer IS this layer's
        self.res_via, self.res_aliased, self.pre_from = None, False, None
        self.reduce_desc = L.WgradReduceDesc()               # filled by every backward-weight launch (identical every time: the plan is static)

    def layout_row(self):
        return [self.weight.data_ptr(), self.wf.data_ptr(), self.wb.data_ptr() if self.wb is not None else 0, self.Cout, self.Cin, self.KH, self.KW,
                self.ldf, self.ldb, self.scale.data_ptr() if (self.scale is not None and self.wb is not None) else 0]

    def fwd(self, plan, st, slot=None):
        """slot None: the whole batch; an int: that image of the batch alone (B = 1 launch on the slot's part of every buffer)."""
        s = self
        L.call("e2e_conv2d_fwd", _at(s.src0.t, slot), _at(s.src1.t, slot) if s.src1 is not None else None, s.C1, s.up, L.ptr(s.wf), s.ldf, L.ptr(s.scale),
               L.ptr(s.shift if s.bias is None else s.bias), _at(s.res.t, slot) if s.res is not None else None, _at(s.out.t, slot),
               s.src0.B if slot is None else 1, s.Hs, s.Ws, s.Cin,
               s.Cout, s.KH, s.KW, s.stride, s.pad, s.pm, s.act, float(s.isub), float(s.imul), L.ptr(s.ws_f), st)

    def bwd(self, plan, st):
        s = self
        B, n = s.src0.B, s.out.t.numel()
        g = s.out.g                                 # d loss / d (pre-activation of this layer)
        if s.res is not None and s.res_via is None and not s.res_aliased:
            # residual branch: its tensor's pre-activation gradient (+)= g * act_res'(.)
            L.call("e2e_conv2d_act_bwd_acc", L.ptr(g), L.ptr(s.res.t), None, L.ptr(s.res.g), n, s.Cout, s.res.act, 1 if s.res.written else 0, st)
            s.res.written = True
        if s.need_dx:
            if s.direct:
                pre = s.pre_from.out.g if s.pre_from is not None else None       # the block's residual gradient: same tensor, same act'
                L.call("e2e_conv2d_bwd_data_fused", L.ptr(g), L.ptr(s.wb), s.ldb, L.ptr(s.src0.g), B, s.Hs, s.Ws, s.Cin, s.Cout, s.Ho, s.Wo, s.KH, s.KW,
                       s.stride, s.pad, s.pm, 1 if s.src0.written else 0, L.ptr(s.src0.t), s.src0.act, L.ptr(pre), L.ptr(s.ws_b), st)
                s.src0.written = True
            else:
                L.call("e2e_conv2d_bwd_data", L.ptr(g), L.ptr(s.wb), s.ldb, L.ptr(s.dxp), B, s.Hs, s.Ws, s.Cin, s.Cout, s.Ho, s.Wo, s.KH, s.KW, s.stride,
                       s.pad, s.pm, L.ptr(s.ws_b), st)
                s1 = s.src1
                L.call("e2e_conv2d_gather_adjoint_act", L.ptr(s.dxp), B, s.Hs, s.Ws, s.Cin, s.C1, s.up, 1 if s.pp else 0, L.ptr(s.src0.g),
                       L.ptr(s1.g) if s1 is not None else None, 1 if s.src0.written else 0, 1 if (s1 is not None and s1.written) else 0,
                       L.ptr(s.src0.t), s.src0.act, L.ptr(s1.t) if s1 is not None else None, s1.act if s1 is not None else 0, st)
                s.src0.written = True
                if s1 is not None:
                    s1.written = True
        st_w = plan.fork(st)
        # the GEMM leaves its partial slabs in this layer's own workspace; the ~30 slab reductions of a pass are ONE launch at its end
        # (NetPlan._reduce_weight_gradients): each was 5 - 15 us of launch latency on an almost empty GPU
        L.call("e2e_conv2d_bwd_weight_scaled_deferred", L.ptr(g), L.ptr(s.scale), L.ptr(s.src0.t), L.ptr(s.src1.t) if s.src1 is not None else None, s.C1, s.up,
               L.ptr(plan.sink(s.weight)), L.ptr(plan.sink(s.bias)) if s.bias is not None else None, L.ptr(s.ws_w), B, s.Hs, s.Ws, s.Cin, s.Cout, s.Ho,
               s.Wo, s.KH, s.KW, s.stride, s.pad, s.pm, 0, float(s.isub), float(s.imul), ctypes.byref(s.reduce_desc), st_w)


class _Head:
    """Conv3x3(reflect) 16 -> 1 + 10 sigmoid + 0.01 (networks.py:271-272,289-290): the VALU head kernels."""

    def __init__(self, plan, src, weight, bias, act):
        self.src, self.weight, self.bias, self.act = src, weight, bias, ACT[act]
        self.out = Buf(src.B, src.h, src.w, 1, plan.dev)
        self.dz = torch.empty_like(self.out.t)
        self.ws = torch.empty(L.load().e2e_head_workspace_floats(), device=plan.dev, dtype=_f32)

    def fwd(self, plan, st, slot=None):
        s = self.src
        L.call("e2e_head_fwd", _at(s.t, slot), L.ptr(self.weight), L.ptr(self.bias), _at(self.out.t, slot), s.B if slot is None else 1, s.h, s.w, s.C, self.act, st)

    def bwd(self, plan, st):
        s, n = self.src, self.out.t.numel()
        L.call("e2e_conv2d_act_bwd", L.ptr(self.out.g), L.ptr(self.out.t), None, L.ptr(self.dz), n, 1, self.act, st)
        if s.written:
            raise RuntimeError("the disparity head's input has a single consumer")
        L.call("e2e_head_bwd_act", L.ptr(self.dz), L.ptr(s.t), L.ptr(self.weight), L.ptr(s.g), L.ptr(plan.sink(self.weight)),
               L.ptr(plan.sink(self.bias)) if self.bias is not None else None, L.ptr(self.ws), s.B, s.h, s.w, s.C, s.act, st)
        s.written = True


class _MaxPool:
    def __init__(self, plan, src):
        self.src = src
        self.out = Buf(src.B, (src.h - 1) // 2 + 1, (src.w - 1) // 2 + 1, src.C, plan.dev)
        self.argmax = torch.empty(self.out.t.shape, device=plan.dev, dtype=torch.uint8)      # window position of every maximum, for the backward

    def fwd(self, plan, st, slot=None):
        s = self.src
        L.call("e2e_maxpool3x3s2_fwd_idx", _at(s.t, slot), _at(self.out.t, slot), _at(self.argmax, slot), s.B if slot is None else 1, s.h, s.w, s.C, st)

    def bwd(self, plan, st):
        s = self.src
        if s.act not in (0, ACT["relu"]):
            raise NotImplementedError("launch plan: the max-pool follows a ReLU (ResNet stem)")
        L.call("e2e_maxpool3x3s2_bwd_idx", L.ptr(s.t), L.ptr(self.argmax), L.ptr(self.out.g), L.ptr(s.g), s.B, s.h, s.w, s.C, 1 if s.written else 0,
               1 if s.act else 0, st)
        s.written = True


class _BNAffine:
    """eval-mode BatchNorm whose gamma / beta still train (`downsample.1`, online_adaption.py:182-184)."""

    def __init__(self, plan, src, bn_mod):
        self.src, self.bn = src, bn_mod
        C = src.C
        self.out = Buf(src.B, src.h, src.w, C, plan.dev)
        self.scale, self.shift, self.rstd = (torch.empty(C, device=plan.dev, dtype=_f32) for _ in range(3))
        self.ws = torch.empty(L.load().e2e_affine_bwd_workspace_floats(C), device=plan.dev, dtype=_f32)

    def fwd(self, plan, st, slot=None):
        b, s = self.bn, self.src
        L.call("e2e_bn_fold", L.ptr(b.weight), L.ptr(b.bias), L.ptr(b.running_mean), L.ptr(b.running_var), float(b.eps), L.ptr(self.scale),
               L.ptr(self.shift), L.ptr(self.rstd), s.C, st)
        L.call("e2e_affine_fwd", _at(s.t, slot), L.ptr(self.scale), L.ptr(self.shift), None, 0, _at(self.out.t, slot),
               s.t.numel() if slot is None else s.t[0].numel(), s.C, st)

    def bwd(self, plan, st):
        b, s = self.bn, self.src
        n = s.t.numel()
        L.call("e2e_affine_bwd", L.ptr(self.out.g), L.ptr(s.t), L.ptr(b.running_mean), L.ptr(self.rstd), n // s.C, s.C, L.ptr(plan.sink(b.weight)),
               L.ptr(plan.sink(b.bias)), 0, L.ptr(self.ws), st)
        if s.act != 0:
            raise NotImplementedError("launch plan: a trainable BatchNorm follows a plain convolution")
        L.call("e2e_conv2d_act_bwd_acc", L.ptr(self.out.g), L.ptr(self.out.g), L.ptr(self.scale), L.ptr(s.g), n, s.C, 0, 1 if s.written else 0, st)
        s.written = True


class NetPlan:
    def __init__(self, model, B, H, W, device, overlap=True):
        """model: depth_estimation.networks.DispResNet_Indoor in refinement mode (every module eval(); BatchNorms either
        frozen or -- the downsample ones -- with a trainable affine).  Input: (B,H,W,3) NHWC frames in [0,1]."""
        from depth_estimation.networks import BasicBlock
        self.model, self.dev, self.overlap = model, torch.device(device), overlap
        self.B, self.H, self.W = B, H, W
        if H % 32 or W % 32:
            raise ValueError(f"launch plan: the encoder halves the image five times and the decoder doubles it back before every skip "
                             f"concatenation (networks.py:277-292): height and width must be multiples of 32, got {H} x {W}")
        self.ops, self._sinks, self._side = [], {}, None
        enc, dec = model.encoder.encoder, model.decoder
        if any(m.training for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)):
            raise NotImplementedError("the launch plan runs BatchNorm in eval mode (refinement mode, online_adaption.py:175-184)")
        self.x = Buf(B, H, W, 3, self.dev, need_grad=False)

        def bn_args(bn):
            return (bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)

        def conv_bn(src, conv, bn, relu, residual=None, in_norm=None):
            if bn.weight.requires_grad or bn.bias.requires_grad:
                if relu or residual is not None or in_norm is not None:
                    raise NotImplementedError("launch plan: a trainable BatchNorm is expected on the downsample branches only (set_refinement_mode)")
                z = self._add(_Conv(self, src, None, conv.weight, None, None, None, None, conv.stride[0], conv.padding[0], "zeros", 1, None))
                return self._add(_BNAffine(self, z, bn))
            return self._add(_Conv(self, src, None, conv.weight, None, bn_args(bn), residual, "relu" if relu else None, conv.stride[0],
                                   conv.padding[0], "zeros", 1, in_norm))

        f = conv_bn(self.x, enc.conv1, enc.bn1, True, in_norm=(0.45, 1.0 / 0.225))
        feats = [f]
        x = self._add(_MaxPool(self, f))
        for stage in (enc.layer1, enc.layer2, enc.layer3, enc.layer4):
            if stage is enc.layer4:
                self.split_index = len(self.ops)   # backward of ops[split_index:] (head, decoder, layer4) = 80 % of the gradient bucket, done first
            for blk in stage:
                if not isinstance(blk, BasicBlock):
                    raise NotImplementedError("launch plan: BasicBlock encoders (ResNet-18 / 34)")
                idt = x if blk.downsample is None else conv_bn(x, blk.downsample[0], blk.downsample[1], False)
                idt_op = self.ops[-1]
                h = conv_bn(x, blk.conv1, blk.bn1, True)
                c1 = self.ops[-1]
                x = conv_bn(h, blk.conv2, blk.bn2, True, residual=idt)
                c2 = self.ops[-1]
                if blk.downsample is None:
                    if c1.direct and c1.need_dx:          # out += identity: d/d(block input) rides in conv1's backward-data epilogue
                        c2.res_via, c1.pre_from = c1, c2
                elif isinstance(idt_op, _BNAffine) and idt.act == 0:
                    idt.g = c2.out.g                       # identity branch without activation and with ONE consumer: share the gradient buffer
                    c2.res_aliased = True
            feats.append(x)
        self.features = feats
        x = feats[-1]
        for i in range(4, -1, -1):
            c0, c1 = dec.convs[("upconv", i, 0)].conv, dec.convs[("upconv", i, 1)].conv
            x = self._add(_Conv(self, x, None, c0.conv.weight, c0.conv.bias, None, None, "elu", 1, 1, c0.pad_mode, 1, None))
            skip = feats[i - 1] if (dec.use_skips and i > 0) else None
            x = self._add(_Conv(self, x, skip, c1.conv.weight, c1.conv.bias, None, None, "elu", 1, 1, c1.pad_mode, 2, None))
        hc = dec.convs[("dispconv", 0)]
        if tuple(hc.conv.weight.shape) != (1, 16, 3, 3) or hc.pad_mode != "reflect":
            raise NotImplementedError("launch plan: the indoor decoder's 16 -> 1 reflect head")
        self.head = _Head(self, x, hc.conv.weight, hc.conv.bias, "disp")
        self.ops.append(self.head)
        self.disp = self.head.out                     # (B,H,W,1) NHWC == (B,1,H,W) contiguous
        self._desc = None
        self._desc_key = None
        self._retired_tables = []
        self._move_tables = {}                               # (src slot, dst slot) -> (copy descriptor table on the device, copies, work items)
        self._reduce_tables = {}                             # "late" / "early" -> (descriptor bytes, device copy, work items)

    def _add(self, op):
        self.ops.append(op)
        return op.out

    # -- parameters / gradient sinks ---------------------------------------------------------------------------------
    def parameters(self):
        out = []
        for op in self.ops:
            if isinstance(op, _Conv) or isinstance(op, _Head):
                out.append(op.weight)
                if op.bias is not None:
                    out.append(op.bias)
            elif isinstance(op, _BNAffine):
                out += [op.bn.weight, op.bn.bias]
        return out

    def sink(self, p):
        """Where the gradient of parameter p is written (its slice of the optimiser's flat bucket when there is one)."""
        s = self._sinks.get(id(p))
        if s is None or s.data_ptr() != _sink_of(p).data_ptr():
            s = self._sinks[id(p)] = _sink_of(p)
            if not s.is_contiguous() or tuple(s.shape) != tuple(p.shape):
                raise RuntimeError("gradient sink of a parameter must be a contiguous tensor of the parameter's shape")
        return s

    # -- weight layouts ------------------------------------------------------------------------------------------------
    def refresh_layouts(self, st=None):
        """k-major GEMM copies of every convolution weight in ONE launch (after an optimiser step every one is stale)."""
        convs = [op for op in self.ops if isinstance(op, _Conv)]
        key = tuple(op.weight.data_ptr() for op in convs)
        if self._desc is None or self._desc_key != key:     # the optimiser re-homed the parameters: new descriptor table
            self._desc = torch.tensor([op.layout_row() for op in convs], dtype=torch.int64).to(self.dev)
            self._desc_key = key
        L.call("e2e_conv_weight_layouts_batched", L.ptr(self._desc), len(convs), st if st is not None else L.stream())
        self.mark_layouts_current()

    def _stamp(self):
        ws = [op.weight for op in self.ops if isinstance(op, _Conv)]
        return (tuple(w.data_ptr() for w in ws), tuple(w._version for w in ws), epoch_sum(ws))

    def mark_layouts_current(self):
        """The plan's GEMM layouts match the weights as they are now (a replayed graph refreshed them after its Adam launch)."""
        self._epoch = self._stamp()

    def layouts_current(self):
        return self._desc is not None and getattr(self, "_epoch", None) == self._stamp()

    # -- streams -------------------------------------------------------------------------------------------------------
    def fork(self, st):
        """Stream for a backward-weight chain: the side stream (after it has seen everything launched so far) or `st`."""
        if not self.overlap:
            return st
        if self._side is None:
            self._side = torch.cuda.Stream(self.dev)
            self._record_side_stream_use()
        self._side.wait_stream(torch.cuda.current_stream(self.dev))
        self._forked = True
        return ctypes.c_void_p(self._side.cuda_stream)

    def join(self):
        if self.overlap and self._side is not None and getattr(self, "_forked", False):
            torch.cuda.current_stream(self.dev).wait_stream(self._side)
            self._forked = False

    def _record_side_stream_use(self):
        """Every buffer a backward-weight chain touches on the side stream is marked as used there (once: the mark stays with the
        block), so that the caching allocator does not hand the block to another stream's allocation, when the plan dies, before the
        side stream's last launch on it has finished.  join() already orders each backward pass before whatever the current stream does
        next; this covers the destruction of a plan whose last join was recorded on a different stream than the one that frees it."""
        seen = set()
        for op in self.ops:
            ts = [op.out.t, op.out.g]
            if isinstance(op, _Conv):
                ts += [op.scale, op.ws_w, op.src0.t, op.src1.t if op.src1 is not None else None]
                for p in (op.weight, op.bias):
                    if p is not None:
                        ts.append(_sink_of(p))
            for t in ts:
                if t is not None and t.is_cuda and t.untyped_storage().data_ptr() not in seen:
                    seen.add(t.untyped_storage().data_ptr())
                    t.record_stream(self._side)

    def close(self):
        """Deterministic end of the plan: both streams drained, nothing of it in flight; the buffers go back to the allocator when the
        last reference dies, with no pending work on any stream."""
        if self._side is not None:
            self._side.synchronize()
        torch.cuda.current_stream(self.dev).synchronize()
        self.ops, self._sinks, self._desc, self._reduce_tables = [], {}, None, {}

    # -- the two passes ------------------------------------------------------------------------------------------------
    def forward(self, frames=None):
        """frames (B,H,W,3) NHWC in [0,1] (copied into the plan's input buffer; None: the buffer was filled by the caller).
        Returns the disparity buffer as a (B,1,H,W) view.  Weight layouts must be current (refresh_layouts)."""
        if frames is not None:
            self.x.t.copy_(frames)
        st = L.stream()
        for op in self.ops:
            op.fwd(self, st)
        return self.disp.t.view(self.B, 1, self.H, self.W)

    def forward_one(self, slot):
        """Forward of ONE image of the batch (B = 1 launches on slot `slot` of every buffer); the other slots keep what they hold."""
        if not 0 <= slot < self.B:
            raise ValueError("slot out of range")
        st = L.stream()
        for op in self.ops:
            op.fwd(self, st, slot)

    def move_slot(self, src, dst):
        """Everything a backward pass reads of image `src` -- every layer's activations, the max-pool's argmax -- copied to image `dst`
        (ONE launch for the ~70 copies: e2e_copy_batched).  With the input frame of `src` loaded into `dst` as well, slot `dst` then holds a complete forward pass."""
        held = self._move_tables.get((src, dst))
        if held is None:
            if not 0 <= src < self.B or not 0 <= dst < self.B or src == dst:
                raise ValueError("slots out of range")
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("launch plan: move_slot builds its copy table on first use; run it eagerly once before capturing it")
            pairs = []
            for op in self.ops:
                pairs.append((op.out.t[src], op.out.t[dst]))
                if isinstance(op, _MaxPool):
                    pairs.append((op.argmax[src], op.argmax[dst]))
            if any(a.numel() * a.element_size() % 16 or a.data_ptr() % 16 or b.data_ptr() % 16 for a, b in pairs):
                for a, b in pairs:                               # (an activation that is not a whole number of 16-byte quads: plain copies)
                    b.copy_(a)
                return
            arr = (L.CopyDesc * len(pairs))(*[L.CopyDesc(a.data_ptr(), b.data_ptr(), a.numel() * a.element_size(), 0) for a, b in pairs])
            total = L.load().e2e_copy_batch_prepare(arr, len(pairs))
            if total <= 0:
                raise RuntimeError("launch plan: malformed copy descriptor")
            held = self._move_tables[(src, dst)] = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev), len(pairs), total)
        L.call("e2e_copy_batched", L.ptr(held[0]), held[1], held[2], L.stream())

    def backward(self, g_disp=None):
        """g_disp (B,1,H,W): gradient of the loss wrt the disparity (None: already in self.disp.g).  Parameter gradients
        go to their sinks; joins the side stream before returning."""
        if g_disp is not None:
            self.disp.g.copy_(g_disp.reshape(self.disp.g.shape))
        self.backward_late_layers()
        self.backward_early_layers()

    def backward_late_layers(self):
        """First part of the backward pass: head, decoder and layer4 -- the layers whose parameters form the tail of the optimiser's
        bucket (split_offset): data-parallel runs start that segment's all-reduce while the rest of the backward runs."""
        st = L.stream()
        for op in self.ops:
            op.out.written = False
        for op in reversed(self.ops[self.split_index:]):
            op.bwd(self, st)
        self.join()
        self._reduce_weight_gradients("late", self.ops[self.split_index:])
        # which gradient buffers hold a contribution at the hand-over point: the second half must start from exactly this state however
        # often either half is executed (the data-parallel step runs each half once eagerly and once more under graph capture)
        self._written_after_late = [op.out.written for op in self.ops]

    def backward_early_layers(self):
        """Second part: layer3 ... stem.  The accumulate flags its launches take are those left by backward_late_layers(), restored here:
        they are launch ARGUMENTS (frozen into a captured graph), so they must not depend on how many times this half ran before."""
        st = L.stream()
        state = getattr(self, "_written_after_late", None)
        if state is None:
            raise RuntimeError("launch plan: backward_early_layers() follows backward_late_layers()")
        for op, w in zip(self.ops, state):
            op.out.written = w
        for op in reversed(self.ops[:self.split_index]):
            op.bwd(self, st)
        self.join()
        self._reduce_weight_gradients("early", self.ops[:self.split_index])

    def _reduce_weight_gradients(self, which, ops):
        """ONE launch for the slab reductions the backward-weight GEMMs of `ops` deferred (e2e_wgrad_reduce_batched).  The descriptor table
        lives on the device; it is rebuilt when a descriptor changed (the optimiser re-homed a gradient sink) -- never under stream capture,
        where the table the graph would keep pointing at must already exist (a plan runs eagerly at least once before it is captured)."""
        convs = [op for op in ops if isinstance(op, _Conv)]
        if not convs:
            return
        arr = (L.WgradReduceDesc * len(convs))(*[op.reduce_desc for op in convs])
        total = L.load().e2e_wgrad_reduce_batch_prepare(arr, len(convs))
        if total <= 0:
            raise RuntimeError("launch plan: malformed backward-weight reduction descriptor")
        raw = bytes(arr)
        held = self._reduce_tables.get(which)
        if held is None or held[0] != raw:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("launch plan: the backward-weight reduction table changed under stream capture; run the plan eagerly once first")
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
            if held is not None:
                self._retired_tables.append(held[1])        # a graph captured earlier may still name the old table: it stays allocated
            held = self._reduce_tables[which] = (raw, table, total)
        L.call("e2e_wgrad_reduce_batched", L.ptr(held[1]), len(convs), held[2], L.stream())

    def split_offset(self, flat):
        """Offset in the flat parameter / gradient bucket where the parameters of ops[split_index:] begin; checks that they ARE the
        bucket's tail (the bucket follows the module order: stem, layer1..4, decoder)."""
        off_of = {id(p): o for p, o in zip(flat.params, flat.offsets)}

        def params_of(ops):
            out = []
            for op in ops:
                for name in ("weight", "bias"):
                    p = getattr(op, name, None)
                    if isinstance(p, torch.nn.Parameter) and p.requires_grad and id(p) in off_of:
                        out.append(off_of[id(p)])
                bn = getattr(op, "bn", None)
                if bn is not None:
                    out += [off_of[id(p)] for p in (bn.weight, bn.bias) if p.requires_grad and id(p) in off_of]
            return out
        late, early = params_of(self.ops[self.split_index:]), params_of(self.ops[:self.split_index])
        if not late or not early or min(late) <= max(early):
            raise RuntimeError("launch plan: the late layers' parameters are not the tail of the flat bucket")
        return min(late)
