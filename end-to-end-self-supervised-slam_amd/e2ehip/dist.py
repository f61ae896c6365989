"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by SEQUENCE: every rank refines its own keyframe pairs against its own global map; the only
exchange per refinement step is the depth network's gradient, which FusedAdam keeps as ONE contiguous fp32
bucket (14 319 409 trainable elements = 57.3 MB, plus a tail element that counts the participating ranks), so the step
is ONE logical all-reduce with no flatten copies -- issued in two segments (exchange_gradients_late_ / _early_) so that 80 % of
it overlaps the second half of the backward pass.
At the end of a run the per-rank maps are gathered (variable length)."""
import os

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def data_parallel():
    """True when a refinement step has to go through the gradient exchange: more than one rank -- or ONE rank in an initialised
    process group with E2E_FORCE_EXCHANGE=1.  The latter runs the N-rank step (backward split into two graphs around the two-segment
    all-reduce, Adam as a third graph) over the REAL transport on a one-GPU box: the sum over one rank is the rank's own gradient and
    the participant count is 1, so such a run must reproduce the plain one bit for bit (tools/rehearse_rccl_one_rank.sh)."""
    return world() > 1 or (os.environ.get("E2E_FORCE_EXCHANGE") == "1" and dist.is_available() and dist.is_initialized())


def _host_staged():
    """gloo carrying GPU tensors (bench.py's one-GPU rehearsal of the N-rank path): collectives run on host copies."""
    return dist.get_backend() == "gloo"


def _all_reduce(t, op):
    if t.is_cuda and _host_staged():
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)


def _all_gather(outs, t):
    if t.is_cuda and _host_staged():
        hs = [o.cpu() for o in outs]
        dist.all_gather(hs, t.cpu())
        for o, h in zip(outs, hs):
            o.copy_(h)
    else:
        dist.all_gather(outs, t)


def broadcast_parameters_(flat, src=0):
    """Start of a data-parallel run: every rank takes rank `src`'s parameters (what DistributedDataParallel does in its constructor) --
    the averaged gradient updates only keep the replicas identical if they start identical (a randomly initialised network differs
    per rank as soon as the ranks consumed different amounts of random numbers, e.g. for their synthetic sequences)."""
    if world() == 1:
        return
    if flat.data.is_cuda and _host_staged():
        h = flat.data.cpu()
        dist.broadcast(h, src=src)
        flat.data.copy_(h)
    else:
        dist.broadcast(flat.data, src=src)
    flat.touched()


def exchange_gradients_(flat, participating=True):
    """The ONE collective of a refinement step (between loss.backward() and optimizer.step(), online_adaption.py:539-540):
    all-reduce(SUM) of FusedAdam's flat gradient bucket.  Keyframe decisions are data dependent (online_adaption.py:234), so a
    rank without a keyframe joins with a zero bucket; the number of contributing ranks rides in the bucket's tail element
    (`flat.grad_ext` = gradients + one 16-byte tail), so it is summed by the SAME all-reduce, and FusedAdam divides by it
    inside the Adam kernel (e2e_adam_step_mean) -- no second collective, no separate division pass.
    `flat`: e2ehip.optim.FlatParams.  world size 1: nothing to do (participants stays 1)."""
    if not participating:
        flat.grad_ext.zero_()
    flat.grad_ext[flat.numel] = 1.0 if participating else 0.0
    if data_parallel():
        _all_reduce(flat.grad_ext, dist.ReduceOp.SUM)
    return flat.grad_ext[flat.numel:flat.numel + 1]


def exchange_gradients_late_(flat, split, participating=True):
    """First half of the step's gradient exchange in data-parallel runs: all-reduce of the bucket's TAIL [split, end) -- the layers
    whose backward runs first (head, decoder, layer4: 80 % of the 57.3 MB) -- together with the participant count, started
    asynchronously so that it travels over xGMI while the rest of the backward pass computes.  Returns a handle for
    exchange_gradients_early_()."""
    if not participating:
        flat.grad_ext[split:].zero_()
    flat.grad_ext[flat.numel] = 1.0 if participating else 0.0
    seg = flat.grad_ext[split:]
    if seg.is_cuda and _host_staged():
        _all_reduce(seg, dist.ReduceOp.SUM)
        return None
    return dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True)


def exchange_gradients_early_(flat, split, handle, participating=True):
    """Second half: the bucket's head [0, split) (stem, layer1-3), then wait for the first half."""
    seg = flat.grad_ext[:split]
    if not participating:
        seg.zero_()
    _all_reduce(seg, dist.ReduceOp.SUM)
    if handle is not None:
        handle.wait()
    return flat.grad_ext[flat.numel:flat.numel + 1]


def allreduce_mean_(flat_grad, participating=True):
    """Stand-alone form for a bare tensor: in-place mean over the participating ranks (bucket and count in ONE all-reduce
    through a temporary with a tail element).  The driver uses exchange_gradients_ on the resident bucket instead."""
    if world() == 1:
        return flat_grad
    ext = torch.empty(flat_grad.numel() + 1, device=flat_grad.device, dtype=flat_grad.dtype)
    ext[:-1] = flat_grad.reshape(-1) if participating else 0.0
    ext[-1] = 1.0 if participating else 0.0
    _all_reduce(ext, dist.ReduceOp.SUM)
    flat_grad.copy_((ext[:-1] / ext[-1].clamp(min=1.0)).view_as(flat_grad))
    return flat_grad


def common_rounds(n_local, device):
    """Every rank must join the same number of gradient exchanges: the number of keyframe rounds of a run is the maximum
    over the ranks (ranks with fewer keyframes idle through the surplus rounds as non-participants)."""
    if world() == 1:
        return int(n_local)
    t = torch.tensor([int(n_local)], device=device, dtype=torch.int64)
    _all_reduce(t, dist.ReduceOp.MAX)
    return int(t.item())


def _broadcast(t, src):
    if t.is_cuda and _host_staged():
        h = t.cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)


def gather_maps(points, normals, colors, ccounts, dst=None):
    """End of a run: the per-rank maps (variable length) -> (points, normals, colors, ccounts, counts), rank after rank.
    dst=None: every rank receives the whole gathered map; dst=r: only rank r does (the others get None for the four arrays and the
    per-rank counts).  Exact sizes: after ONE small all-gather of the counts the receiver allocates sum(counts) rows per array and every
    rank's rows travel straight into their slice -- one broadcast (dst=None) or one send / receive (dst=r) per array and rank, no
    padding, no staging copy.  (Round 3 padded every rank's map to the largest one and materialised world x largest x 40 B on every
    rank, plus a packed copy and a concatenation: 8 x 470 MB x 3 for a full pass on 8 ranks, to report a size.)"""
    arrays = (points, normals, colors, ccounts)
    if world() == 1:
        return points, normals, colors, ccounts, torch.tensor([points.shape[0]])
    dev, w, me = points.device, world(), dist.get_rank()
    n = torch.tensor([points.shape[0]], device=dev, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(w)]
    _all_gather(counts, n)
    counts = torch.cat(counts).cpu()
    offs = [0]
    for c in counts.tolist():
        offs.append(offs[-1] + int(c))
    receive = dst is None or me == dst
    outs = [torch.empty((offs[-1],) + tuple(a.shape[1:]), device=dev, dtype=a.dtype) for a in arrays] if receive else [None] * 4
    for r in range(w):
        lo, hi = offs[r], offs[r + 1]
        if hi == lo:
            continue
        for k, a in enumerate(arrays):
            if dst is None:
                buf = outs[k][lo:hi]                                # a contiguous slice of the result: received in place
                if r == me:
                    buf.copy_(a)
                _broadcast(buf, r)
            elif me == dst:
                if r == me:
                    outs[k][lo:hi].copy_(a)
                else:
                    _recv(outs[k][lo:hi], r)
            elif r == me:
                _send(a.contiguous(), dst)
    return outs[0], outs[1], outs[2], outs[3], counts


def _send(t, dst):
    dist.send(t.cpu() if (t.is_cuda and _host_staged()) else t, dst=dst)


def _recv(t, src):
    if t.is_cuda and _host_staged():
        h = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(h, src=src)
        t.copy_(h)
    else:
        dist.recv(t, src=src)
