"""Fused multi-tensor Adam: every parameter (and its gradient and moments) lives in ONE flat fp32 buffer, so a
step is a single streaming kernel over 4 arrays and the data-parallel gradient exchange is a single RCCL
all-reduce of one contiguous 57 MB bucket (no per-tensor launches, no flatten/unflatten copies)."""
import torch

from . import _lib as L


class FlatParams:
    """Re-homes a list of parameters into one contiguous, 16-byte aligned buffer (params become views) and keeps
    a matching flat gradient buffer whose slices are installed as `.grad`."""

    def __init__(self, params):
        params = [p for p in params]
        if not params:
            raise ValueError("no parameters")
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4            # keep every slice 16-byte aligned
        self.params, self.offsets, self.numel = params, offs, total
        self.data = torch.zeros(total, device=dev, dtype=torch.float32)
        # gradients + a 16-byte tail whose first element counts the ranks that contributed to the bucket (dist.exchange_gradients_)
        self.grad_ext = torch.zeros(total + 4, device=dev, dtype=torch.float32)
        self.grad = self.grad_ext[:total]
        self.grad_ext[total] = 1.0
        with torch.no_grad():
            for p, o in zip(params, offs):
                self.data[o:o + p.numel()].copy_(p.reshape(-1))
                p.data = self.data[o:o + p.numel()].view_as(p)
        # state the convolution module path reads through the parameters (e2ehip.conv): a counter of raw-pointer rewrites of the
        # parameters (their torch version counters do not move), and the direct-gradient switch + side stream of direct_weight_grads()
        self.epoch = [1]
        self.direct, self.overlap, self.side = False, False, None
        self.bind_grads()

    def bind_grads(self):
        for p, o in zip(self.params, self.offsets):
            p.grad = self.grad[o:o + p.numel()].view_as(p)
            p._e2e_grad_sink = p.grad              # conv backward may accumulate straight into it (conv.direct_weight_grads)
            p._e2e_grad_owner = self
            p._e2e_epoch = self.epoch

    def touched(self):
        """The parameters were rewritten through raw pointers (Adam kernel, broadcast): GEMM layouts cached on them are stale."""
        self.epoch[0] += 1

    def side_stream(self):
        if self.side is None:
            self.side = torch.cuda.Stream(self.data.device)
        return self.side

    def zero_grad(self):
        self.grad.zero_()
        self.bind_grads()

    def participants(self):
        return self.grad_ext[self.numel:self.numel + 1]


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) semantics (the reference's optimiser:
    utils/training_utils.py:23-25) in one HIP launch.  Like torch, parameters whose gradient never appears
    are not moved: a parameter is only updated from the first step in which its gradient slice is non-None,
    and frozen parameters (requires_grad False) are skipped -- both are handled by an `active` mask folded
    into the flat layout (inactive parameters are simply left out of the flat buffer)."""

    SCHEDULE_LEN = 1 << 18          # bias-correction table: 262 144 steps x 8 B (a 60-frame sequence takes 177)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        # weight_decay / amsgrad: torch.optim.Adam's remaining defaults, carried (at their off values) so that a state dict written here
        # is a complete torch.optim.Adam state dict
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        if len(self.param_groups) != 1:
            raise NotImplementedError("FusedAdam steps ONE parameter group (one lr / betas / eps for the flat bucket)")
        self._all = list(self.param_groups[0]["params"])
        self._pending = None
        self.flat = None
        self.steps = 0              # host-side count of step() CALLS; the step that counts lives on the device (graph replays)
        self._sched = self._sched_key = self._counter = None

    def _resident_state(self):
        """Device-resident step counter + the per-step scalars {lr / (1 - beta1^t), sqrt(1 - beta2^t)}, computed on the host in
        double exactly as torch.optim.Adam does and rounded to fp32 once; rebuilt in place when lr / betas change."""
        g = self.param_groups[0]
        key = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]))
        dev = self.flat.data.device
        if self._counter is None:
            self._counter = torch.ones(1, device=dev, dtype=torch.int32)
            self._sched = torch.empty(self.SCHEDULE_LEN, 2, device=dev, dtype=torch.float32)
        if self._sched_key != key:
            t = torch.arange(1, self.SCHEDULE_LEN + 1, dtype=torch.float64)
            bc1 = 1.0 - torch.pow(torch.tensor(key[1], dtype=torch.float64), t)
            bc2 = 1.0 - torch.pow(torch.tensor(key[2], dtype=torch.float64), t)
            self._sched.copy_(torch.stack([key[0] / bc1, torch.sqrt(bc2)], 1).to(torch.float32))
            self._sched_key = key
        return self._sched, self._counter

    def steps_done(self):
        """Optimiser steps taken so far, read from the device counter (host sync)."""
        return 0 if self._counter is None else int(self._counter.item()) - 1

    def _build(self):
        # parameters that take part: trainable AND with a gradient at the first step (torch skips grad None)
        act = [p for p in self._all if p.requires_grad and p.grad is not None]
        if not act:
            raise RuntimeError("FusedAdam.step(): no parameter has a gradient")
        grads = [p.grad.detach().clone() for p in act]
        self.flat = FlatParams(act)
        with torch.no_grad():
            for p, g in zip(act, grads):
                p.grad.copy_(g)
        self.m = torch.zeros_like(self.flat.data)
        self.v = torch.zeros_like(self.flat.data)
        self._apply_pending()

    def prebuild(self, params):
        """Fix the flat layout before the first backward (data-parallel runs: a rank may have to join a gradient exchange
        before its own first keyframe).  `params`: the parameters that will receive gradients, in a rank-independent order."""
        if self.flat is not None:
            return
        known = {id(p) for p in self._all}
        act = [p for p in params if p.requires_grad and id(p) in known]
        if not act:
            raise RuntimeError("FusedAdam.prebuild(): no trainable parameter")
        self.flat = FlatParams(act)
        self.m = torch.zeros_like(self.flat.data)
        self.v = torch.zeros_like(self.flat.data)
        self._apply_pending()

    # ---- torch.optim.Adam-compatible state (train_depth.py:849-863 resumes from `<optimizer>.pth`) --------------------------------
    def state_dict(self):
        """The optimiser state in torch.optim.Adam's own format -- {"state": {param index: {"step", "exp_avg", "exp_avg_sq"}},
        "param_groups": [...]} with indices into the parameter list the optimiser was built on -- so a file written here loads into the
        reference's Adam and vice versa.  Parameters that never received a gradient carry no state, as in torch."""
        groups = []
        start = 0
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d["params"] = list(range(start, start + len(g["params"])))
            start += len(g["params"])
            groups.append(d)
        state = {}
        if self.flat is None and getattr(self, "_pending", None):
            state = {i: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in self._pending.items()}     # loaded, not applied yet
        elif self.flat is not None and self.steps_done() > 0:                  # like torch: no state before the first step
            t = float(self.steps_done())
            idx = {id(p): i for i, p in enumerate(self._all)}
            for p, o in zip(self.flat.params, self.flat.offsets):
                n = p.numel()
                state[idx[id(p)]] = {"step": torch.tensor(t), "exp_avg": self.m[o:o + n].view_as(p).clone(), "exp_avg_sq": self.v[o:o + n].view_as(p).clone()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        """Counterpart of torch.optim.Optimizer.load_state_dict for the flat layout: hyper-parameters, moments and step count.
        WHICH parameters the flat bucket holds is decided where it always is -- at prebuild() (the caller names the parameters that
        take part) or at the first step() (those with a gradient) -- never by the file: loaded before that, the state waits and is
        applied when the bucket exists.  Adam here keeps ONE step counter for the bucket, torch one per parameter; they are equal in
        any file an Adam run over the same parameters wrote.  A bucket parameter WITHOUT state in a file whose others have stepped
        (torch would start it at step 0 with its own bias correction) cannot be represented and raises instead of being silently
        mis-stepped or -- round 3 -- silently left out of the bucket and never updated."""
        groups = state_dict["param_groups"]
        if len(groups) != 1 or len(self.param_groups) != 1:
            raise NotImplementedError("FusedAdam steps ONE parameter group (one lr / betas / eps for the flat bucket)")
        if len(groups[0]["params"]) != len(self._all):
            raise ValueError("loaded state dict has a different number of parameters")
        src = groups[0]
        if src.get("weight_decay", 0) or src.get("amsgrad", False) or src.get("maximize", False):
            raise NotImplementedError("FusedAdam: weight_decay / amsgrad / maximize are not on the reference's path (training_utils.py:23-25)")
        self.param_groups[0].update({k: v for k, v in src.items() if k in ("lr", "betas", "eps")})
        self._sched_key = None                              # lr / betas may have changed: rebuild the bias-correction table at the next step
        state = {int(k): v for k, v in state_dict["state"].items()}
        for i, st in state.items():
            if not 0 <= i < len(self._all):
                raise ValueError(f"state for parameter index {i}: the optimiser has {len(self._all)} parameters")
            if tuple(st["exp_avg"].shape) != tuple(self._all[i].shape):
                raise ValueError(f"parameter {i}: state of shape {tuple(st['exp_avg'].shape)} for a parameter of shape {tuple(self._all[i].shape)}")
        self._pending = state or None
        if self.flat is not None:
            self._apply_pending()

    def _apply_pending(self):
        state, self._pending = getattr(self, "_pending", None), None
        if not state:
            return
        idx = {id(p): i for i, p in enumerate(self._all)}
        offs = {idx[id(p)]: (o, p.numel()) for p, o in zip(self.flat.params, self.flat.offsets)}
        steps = {int(float(st["step"])) for st in state.values()}
        if len(steps) != 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused optimiser keeps one")
        t = steps.pop()
        missing = sorted(i for i in offs if i not in state)
        if missing and t != 0:
            raise ValueError(f"parameters {missing} are trained here but carry no state in the loaded file, whose other parameters are at "
                             f"step {t}: torch.optim.Adam would run them on their own step count, the fused optimiser keeps one")
        extra = sorted(i for i in state if i not in offs and self._all[i].requires_grad)
        if extra:
            raise ValueError(f"parameters {extra} carry optimiser state but are not part of the flat bucket (not among the parameters "
                             "named to prebuild() / without a gradient at the first step)")
        with torch.no_grad():
            for i, st in state.items():
                if i in offs:
                    o, n = offs[i]
                    self.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                    self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
        self._resident_state()
        self._counter.fill_(t + 1)

    def zero_grad(self, set_to_none=True):
        if self.flat is None:
            for p in self._all:
                p.grad = None
        else:
            self.flat.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        if self.flat is None:
            self._build()
        g = self.param_groups[0]
        self.steps += 1
        self.flat.touched()                            # the kernel rewrites the weights behind torch's version counters
        # the bucket holds the SUM over the data-parallel ranks and its tail the number of contributors (1 on a single GPU);
        # step count and bias corrections are device-resident, so this launch can be captured into a hipGraph and replayed
        sched, counter = self._resident_state()
        L.call("e2e_adam_step_resident", L.ptr(self.flat.data), L.ptr(self.flat.grad), L.ptr(self.flat.participants()), L.ptr(self.m), L.ptr(self.v),
               self.flat.numel, float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), L.ptr(sched), self.SCHEDULE_LEN, L.ptr(counter), L.stream())
