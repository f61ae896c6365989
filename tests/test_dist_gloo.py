"""N>1 path on CPU: world_size-2 gloo processes exercise the flat gradient bucket exchange (with and without a
keyframe on one rank) and the variable-length map gather."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from e2ehip import dist as edist
    from e2ehip.optim import FlatParams
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2, 3, 3))]
    flat = FlatParams(ps)
    assert all(p.data_ptr() >= flat.data.data_ptr() for p in ps) and flat.numel % 4 == 0
    # every rank: loss depends on rank -> different grads; all-reduce(mean) must equal the analytic mean
    loss = sum(((rank + 1.0) * p).sum() for p in ps)
    loss.backward()
    edist.allreduce_mean_(flat.grad)
    ok1 = all(torch.allclose(p.grad, torch.full_like(p, (1 + world) / 2.0)) for p in ps)
    # rank 1 has no keyframe this round: contributes zeros, divisor = 1 participant
    flat.zero_grad()
    (3.0 * ps[0]).sum().backward()
    edist.allreduce_mean_(flat.grad, participating=(rank == 0))
    ok2 = torch.allclose(ps[0].grad, torch.full_like(ps[0], 3.0)) and float(ps[1].grad.abs().sum()) == 0.0
    # map gather: rank r holds r+2 points
    n = rank + 2
    P, Nn, C, cc = (torch.full((n, 3), float(rank)), torch.ones(n, 3), torch.zeros(n, 3), torch.arange(n).float())
    gp, gn, gc, gcc, counts = edist.gather_maps(P, Nn, C, cc)
    ok3 = gp.shape[0] == sum(r + 2 for r in range(world)) and counts.tolist() == [r + 2 for r in range(world)] and \
        torch.equal(gp[:2], torch.zeros(2, 3)) and torch.equal(gp[2:], torch.ones(3, 3)) and torch.equal(gcc, torch.tensor([0., 1, 0, 1, 2]))
    q.put((rank, ok1, ok2, ok3))
    dist.destroy_process_group()


def test_two_rank_gloo_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True, True), (1, True, True, True)], res


def test_single_process_is_identity():
    sys.path[:0] = [os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")]
    from e2ehip import dist as edist
    g = torch.randn(16)
    ref = g.clone()
    assert torch.equal(edist.allreduce_mean_(g), ref)
    P = torch.randn(4, 3)
    out = edist.gather_maps(P, P, P, torch.ones(4))
    assert torch.equal(out[0], P) and out[4].tolist() == [4]
