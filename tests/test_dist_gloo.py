"""N>1 path on CPU: world_size-2 gloo processes exercise the flat gradient bucket exchange (with and without a keyframe on
one rank), the driver's SLAM._exchange_gradients / idle_round on CPU-side stand-ins for the flat bucket, the agreement on a
common number of keyframe rounds, the variable-length map gather, and bench.py's own multi-process launch (--gpus 2 --dry)."""
import json
import os
import subprocess
import sys
import types

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, PKG]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from e2ehip import dist as edist
    from e2ehip.optim import FlatParams
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2, 3, 3))]
    flat = FlatParams(ps)
    assert all(p.data_ptr() >= flat.data.data_ptr() for p in ps) and flat.numel % 4 == 0 and flat.grad_ext.numel() == flat.numel + 4
    # (1) every rank has a keyframe: ONE all-reduce carries the sums and the participant count
    loss = sum(((rank + 1.0) * p).sum() for p in ps)
    loss.backward()
    cnt = edist.exchange_gradients_(flat, participating=True)
    ok1 = float(cnt) == world and all(torch.allclose(p.grad / cnt, torch.full_like(p, (1 + world) / 2.0)) for p in ps)
    # (2) rank 1 has no keyframe this round: zero bucket, divisor = 1 participant
    flat.zero_grad()
    if rank == 0:
        (3.0 * ps[0]).sum().backward()
    cnt = edist.exchange_gradients_(flat, participating=(rank == 0))
    ok2 = float(cnt) == 1.0 and torch.allclose(ps[0].grad / cnt, torch.full_like(ps[0], 3.0)) and float(ps[1].grad.abs().sum()) == 0.0
    # (3) the driver's own methods on a stand-in that carries the flat bucket (no GPU, no network)
    from online_adaption import SLAM
    steps = []
    fake_opt = types.SimpleNamespace(flat=flat, zero_grad=flat.zero_grad, step=lambda: steps.append(float(flat.participants())), _build=lambda: None)
    fake = types.SimpleNamespace(optimizer=fake_opt, step_plan=None, args=types.SimpleNamespace(OPTIMIZATION=types.SimpleNamespace(refinement_steps=3)))
    fake._exchange_gradients = types.MethodType(SLAM._exchange_gradients, fake)
    if rank == 0:                                    # rank 0 refines a keyframe (3 steps), rank 1 idles through the same round
        for _ in range(3):
            flat.zero_grad()
            (2.0 * ps[1]).sum().backward()
            fake._exchange_gradients()
            fake_opt.step()
    else:
        SLAM.idle_round(fake)
    ok3 = steps == [1.0, 1.0, 1.0] and torch.allclose(ps[1].grad, torch.full_like(ps[1], 2.0))      # same averaged bucket on both ranks
    # (4) common number of keyframe rounds; stand-alone tensor form
    ok4 = edist.common_rounds(5 + 2 * rank, torch.device("cpu")) == 5 + 2 * (world - 1)
    g = torch.full((6,), float(rank + 1))
    ok4 = ok4 and torch.allclose(edist.allreduce_mean_(g), torch.full((6,), (1 + world) / 2.0))
    # (5) map gather: rank r holds r+2 points
    n = rank + 2
    P, Nn, C, cc = (torch.full((n, 3), float(rank)), torch.ones(n, 3), torch.zeros(n, 3), torch.arange(n).float())
    gp, gn, gc, gcc, counts = edist.gather_maps(P, Nn, C, cc)
    ok5 = gp.shape[0] == sum(r + 2 for r in range(world)) and counts.tolist() == [r + 2 for r in range(world)] and \
        torch.equal(gp[:2], torch.zeros(2, 3)) and torch.equal(gp[2:], torch.ones(3, 3)) and torch.equal(gcc, torch.tensor([0., 1, 0, 1, 2]))
    # ... to one rank only (exact sizes, point-to-point): the others receive nothing but the counts
    gp1, _, _, gcc1, counts1 = edist.gather_maps(P, Nn, C, cc, dst=1)
    ok5 = ok5 and counts1.tolist() == counts.tolist() and ((gp1 is None and gcc1 is None) if rank != 1 else
                                                          (torch.equal(gp1, gp) and torch.equal(gcc1, gcc)))
    # (6) the launch plan's two-segment form: tail [split, end) + participant count asynchronously, then the head [0, split);
    #     rank 1 idles (zero bucket) -- and the parameter broadcast that makes the replicas start identical
    flat.zero_grad()
    split = flat.offsets[1]
    if rank == 0:
        sum((4.0 * p).sum() for p in ps).backward()
    h = edist.exchange_gradients_late_(flat, split, participating=(rank == 0))
    cnt = edist.exchange_gradients_early_(flat, split, h, participating=(rank == 0))
    ok6 = float(cnt) == 1.0 and all(torch.allclose(p.grad, torch.full_like(p, 4.0)) for p in ps)
    with torch.no_grad():
        flat.data.add_(float(rank))                  # replicas drift apart ...
    edist.broadcast_parameters_(flat)                # ... rank 0's parameters win
    ref = [torch.zeros_like(flat.data) for _ in range(world)]
    dist.all_gather(ref, flat.data)
    ok6 = ok6 and all(torch.equal(r, ref[0]) for r in ref)
    q.put((rank, ok1, ok2, ok3, ok4, ok5, ok6))
    dist.destroy_process_group()


def test_two_rank_gloo_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0,) + (True,) * 6, (1,) + (True,) * 6], res


def test_single_process_is_identity():
    sys.path[:0] = [PKG]
    from e2ehip import dist as edist
    from e2ehip.optim import FlatParams
    g = torch.randn(16)
    ref = g.clone()
    assert torch.equal(edist.allreduce_mean_(g), ref)
    flat = FlatParams([torch.nn.Parameter(torch.randn(3, 3))])
    flat.grad.fill_(2.0)
    assert float(edist.exchange_gradients_(flat)) == 1.0 and torch.equal(flat.grad, torch.full((12,), 2.0))
    assert edist.common_rounds(7, torch.device("cpu")) == 7
    P = torch.randn(4, 3)
    out = edist.gather_maps(P, P, P, torch.ones(4))
    assert torch.equal(out[0], P) and out[4].tolist() == [4]


def test_bench_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` with no launcher environment must start 2 ranks itself (the driver's 8-GPU run relies on the
    same path through torch.distributed.run); --dry keeps the ranks on the CPU (gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "7"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 7 and out["dry"] is True
    assert out["config"]["rounds"] == 3 and out["config"]["map_points_per_rank"] == [2, 3] and out["config"]["map_points_gathered"] == 5
    # step 7 (index 6) belongs to keyframe round 2: only rank 1 (2 + 1 = 3 keyframes) still participates
    assert out["config"]["participants_last_step"] == 1.0


def _one_rank_worker(port, q):
    sys.path[:0] = [ROOT, PKG]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from e2ehip import dist as edist
    from e2ehip.optim import FlatParams
    a = edist.data_parallel()                            # no process group: never
    os.environ["E2E_FORCE_EXCHANGE"] = "1"
    b = edist.data_parallel()                            # the switch alone is not enough either
    dist.init_process_group("gloo", rank=0, world_size=1)
    c = edist.data_parallel()
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    flat = FlatParams(ps)
    (2.0 * ps[0]).sum().backward()
    (3.0 * ps[1]).sum().backward()
    before = flat.grad_ext.clone()
    split = 16
    h = edist.exchange_gradients_late_(flat, split, True)
    cnt = edist.exchange_gradients_early_(flat, split, h, True)
    same = torch.equal(flat.grad_ext[: flat.numel], before[: flat.numel]) and float(cnt) == 1.0
    del os.environ["E2E_FORCE_EXCHANGE"]
    d = edist.data_parallel()
    dist.destroy_process_group()
    q.put((a, b, c, same, d))


def test_forced_exchange_on_one_rank_is_the_identity():
    """E2E_FORCE_EXCHANGE=1 + a process group of size 1 (tools/rehearse_rccl_one_rank.sh on the GPU box): the step takes the N-rank path,
    and the two-segment exchange leaves the bucket as it was with a participant count of 1."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(29700 + os.getpid() % 200, q))
    p.start()
    res = q.get(timeout=120)
    p.join(60)
    assert res == (False, False, True, True, False)


def _ordering_worker(rank, world, port, q):
    """The two-segment exchange as RefineStepPlan.step issues it (late backward -> asynchronous all-reduce of the bucket's tail -> early
    backward -> all-reduce of the head -> wait -> Adam), with the asynchronous collective made as late as the protocol allows: the
    handle performs the tail's all-reduce only inside wait().  Whatever touches the tail between the issue and the wait, and whoever reads
    it before the wait, then changes the outcome deterministically."""
    sys.path[:0] = [ROOT, PKG]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from e2ehip import dist as edist
    from e2ehip.optim import FlatParams
    torch.manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(6, 4)), torch.nn.Parameter(torch.randn(8)), torch.nn.Parameter(torch.randn(3, 2, 3, 3)), torch.nn.Parameter(torch.randn(12))]
    flat = FlatParams(ps)
    split = flat.offsets[2]                              # "late layers" = the last two parameters = the bucket's tail
    g = torch.Generator().manual_seed(10 + rank)
    grad = torch.randn(flat.numel, generator=g)
    want = grad.clone()
    dist.all_reduce(want)                                # the one-shot reference: sum over the ranks
    real, events = dist.all_reduce, []

    class Lazy:
        def __init__(self, t, op):
            self.t, self.op = t, op

        def wait(self):
            events.append("wait tail")
            real(self.t, op=self.op)

    def patched(t, op=dist.ReduceOp.SUM, async_op=False):
        if async_op:
            events.append("issue tail")
            return Lazy(t, op)
        events.append("reduce head")
        return real(t, op=op)

    def step(early_work):
        flat.grad_ext.zero_()
        flat.grad[split:] = grad[split:]                 # late backward: head / decoder / layer4 gradients -- the tail
        h = edist.exchange_gradients_late_(flat, split, True)
        early_work()                                     # early backward: layer3 ... stem, while the tail travels
        cnt = edist.exchange_gradients_early_(flat, split, h, True)
        return flat.grad.clone(), float(cnt)             # what Adam reads

    edist.dist.all_reduce = patched
    try:
        def correct():
            flat.grad[:split] = grad[:split]             # the early layers' parameters are the bucket's head (NetPlan.split_offset checks it)
        got, cnt = step(correct)
        ok_result = torch.equal(got, want) and cnt == float(world)
        ok_order = events == ["issue tail", "reduce head", "wait tail"]       # the wait sits behind the head's reduction, before the return
        # negative controls: the test bites.  (a) early work that strays into the tail; (b) a consumer that does not wait
        def strays():
            correct()
            flat.grad[split + 1] += 1.0
        bad, _ = step(strays)
        bites_a = not torch.equal(bad, want)
        flat.grad_ext.zero_()
        flat.grad[split:] = grad[split:]
        h = edist.exchange_gradients_late_(flat, split, True)
        correct()
        unwaited = flat.grad.clone()                      # read before exchange_gradients_early_ / wait
        edist.exchange_gradients_early_(flat, split, h, True)
        bites_b = not torch.equal(unwaited[split:], want[split:])
    finally:
        edist.dist.all_reduce = real
    q.put((rank, ok_result, ok_order, bites_a, bites_b))
    dist.destroy_process_group()


def test_two_segment_exchange_ordering():
    """the late segment's asynchronous all-reduce is issued between two graph replays; nothing tested that the bucket's
    tail is left alone until handle.wait() and that wait() precedes the optimiser.  world_size 2, gloo, the asynchronous handle deferred to its
    wait(): the exchanged bucket equals the one-shot all-reduce bit for bit, the call order is issue-tail / reduce-head / wait-tail, and
    both ordering mistakes (early work writing into the tail; reading before the wait) change the result."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ordering_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True, True, True), (1, True, True, True, True)], res
