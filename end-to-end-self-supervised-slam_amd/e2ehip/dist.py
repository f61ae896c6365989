"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by SEQUENCE: every rank refines its own keyframe pairs against its own global map; the only
exchange per refinement step is the depth network's gradient, which FusedAdam keeps as ONE contiguous fp32
bucket (14 319 409 trainable elements = 57.3 MB), so the step is a single all-reduce with no flatten copies.
At the end of a run the per-rank maps are gathered (variable length)."""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_mean_(flat_grad, participating=True):
    """In-place average of the flat gradient bucket over the ranks that took a refinement step this round.
    Keyframe decisions are data dependent (online_adaption.py:234), so a rank without a keyframe joins with a
    zero bucket and the sum is divided by the number of participants (all-reduced alongside, one extra float)."""
    if world() == 1:
        return flat_grad
    if not participating:
        flat_grad.zero_()
    cnt = torch.tensor([1.0 if participating else 0.0], device=flat_grad.device)
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    flat_grad.div_(cnt.clamp(min=1.0))
    return flat_grad


def gather_maps(points, normals, colors, ccounts):
    """Variable-length all-gather of the per-rank maps -> concatenated (points, normals, colors, ccounts, counts)."""
    if world() == 1:
        return points, normals, colors, ccounts, torch.tensor([points.shape[0]])
    dev = points.device
    n = torch.tensor([points.shape[0]], device=dev, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world())]
    dist.all_gather(counts, n)
    counts = torch.cat(counts)
    cap = int(counts.max())
    packed = torch.zeros(cap, 10, device=dev, dtype=torch.float32)          # 40 B per point
    packed[: points.shape[0]] = torch.cat([points, normals, colors, ccounts.reshape(-1, 1)], 1)
    parts = [torch.empty_like(packed) for _ in range(world())]
    dist.all_gather(parts, packed)
    full = torch.cat([p[: int(c)] for p, c in zip(parts, counts)], 0)
    return full[:, 0:3], full[:, 3:6], full[:, 6:9], full[:, 9], counts.cpu()
