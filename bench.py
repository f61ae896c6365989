#!/usr/bin/env python3
"""bench.py -- refinement-step throughput of the MI355X hot path (see DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]

Workload (BASELINE.json configs[1]): synthetic 640x480 RGB-D pair, image-space part of one
refinement step = fused warp + masked SSIM/L1 photometric (+ l2 depth regulariser) forward AND
backward, inputs resident in HBM.  One "step" = e2e_warp_photo_fwd (+1-block reduce) +
e2e_warp_photo_bwd, replayed from a captured HIP graph.  N>1: one process per GPU, each rank owns
its own pair (the path shards by sequence; this workload has no exchange step), value = total
steps / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# HBM bytes per launch of the dominant kernel from separate rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE, KiB -> bytes;
# profiles/r01_final_summaries.md).  The kernel loads 4- and 12-byte items, a pattern for which gfx950's FETCH_SIZE
# halving (seen on 16-B/lane streams) is not calibrated, so the counters are taken at face value.
TRAFFIC_NOTE = {(False, 1): (11065 + 2438) * 1024}
# Why the HBM fraction of this kernel is low: it is bound by VALU issue, not by bytes.  SQ_INSTS_VALU per launch (same PMC
# runs) x 4 cycles per wave64 instruction / 1024 SIMDs, against the measured launch duration at the ~2.3 GHz the counters show.
VALU_NOTE = {(False, 1): {"wave_instructions": 3520800, "issue_cycles_per_simd": 3520800 * 4 // 1024, "clock_ghz": 2.3}}


def cpu_baseline(H, W, budget_s=12.0):
    """The oracle (CPU restatement, kind 'port') on this node's host cores, same workload."""
    from oracle import warp_loss
    from synth import make_pair
    ncpu = os.cpu_count() or 1
    s = make_pair(H, W, seed=1234)
    g = torch.Generator().manual_seed(5)
    dsrc = s["depth"] + 0.1 * torch.rand(s["depth"].shape, generator=g)
    it, is_ = s["depth"] + 0.05, dsrc + 0.05
    src, tgt = s["src"].permute(0, 3, 1, 2), s["tgt"].permute(0, 3, 1, 2)

    def step():
        d = s["depth"].clone().requires_grad_(True)
        ds = dsrc.clone().requires_grad_(True)
        synth, valid, _ = warp_loss.inverse_warp(d, src, s["K"], s["invK"], s["T"], "border")
        lp, _ = warp_loss.masked_photometric_mean(synth, tgt, valid)
        loss = lp + 1e-2 * (warp_loss.depth_regularizer(it, d, "l2") + warp_loss.depth_regularizer(is_, ds, "l2"))
        loss.backward()
        return loss

    # torch's intra-op pool oversubscribes badly on a many-core host for these small ops: probe a few
    # thread counts briefly and time the best one (cores = threads actually used).
    best = None
    for th in sorted({t for t in (8, 16, 32, 64) if t <= ncpu} | {min(ncpu, 8)}):
        torch.set_num_threads(th)
        step()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        rate = 3 / (time.perf_counter() - t0)
        if best is None or rate > best[0]:
            best = (rate, th)
    cores = best[1]
    torch.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 400:
            break
    return {"value": n / el, "unit": "steps/s", "cores": cores, "kind": "port", "host_cpus": ncpu,
            "sample": f"{n} fwd+bwd steps of the same 1x{H}x{W} warp+photometric+reg workload ({el:.1f} s), torch CPU, {cores} threads (best of a short sweep)"}


def cpu_baseline_full(H, W, threads=16):
    """The oracle's refinement step (depth net forward + backward on the pair, median scaling, warp + photometric +
    regulariser, Adam) on this node's host cores -- WITHOUT the 3-D nearest-neighbour loss and without the map step: the
    reference's CPU KNN alone takes 10^2-10^3 s per step at this size, far outside a bounded sample."""
    from e2ehip.synthetic import make_sequence
    from oracle import depthnet, refine
    ncpu = os.cpu_count() or 1
    cores = min(threads, ncpu)
    torch.set_num_threads(cores)
    colors, depths, intr, poses = make_sequence(2, H, W, seed=1234)[:4]          # CPU tensors, colours in [0, 1]
    r = refine.Refiner(depthnet.random_state_dict(0))
    t0 = time.perf_counter()
    recs = r.refine_pair(colors, depths, poses, intr, update_map=False)
    el = time.perf_counter() - t0
    return {"value": len(recs) / el, "unit": "steps/s", "cores": cores, "kind": "port", "host_cpus": ncpu,
            "sample": f"{len(recs)} refinement steps of one 1x{H}x{W} keyframe pair ({el:.1f} s), torch CPU, {cores} threads; 3-D KNN loss and map step excluded"}


def full_step_bench(a, rank, world, dev):
    """Whole refinement steps (BASELINE configs[2] shape on a synthetic sequence): per keyframe pair 3 x (network fwd+bwd on the
    pair, median scale, fused warp/photometric/regulariser, 3-D nearest-neighbour loss against the map, Adam) + the map update."""
    import torch.distributed as dist
    from e2ehip.synthetic import make_sequence
    from online_adaption import SLAM, default_config
    H, W = a.height, a.width
    cfg = default_config(H, W, 3)
    cfg.DEBUG.print_metrics = False
    cfg.DEMO.frame_threshold = 0.0
    cfg.MODEL.map_capacity = (8 + 2 * (a.warmup + a.steps)) * H * W
    slam = SLAM(cfg, sequence=make_sequence(3, H, W, seed=1234 + rank))
    slam.main()                                   # pairs (0,1), (1,2): builds the map, warms everything up
    for _ in range(a.warmup):
        slam.refinement(1, 2)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        slam.refinement(1, 2)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    nsteps = a.steps * cfg.OPTIMIZATION.refinement_steps
    if rank == 0:
        from e2ehip import nn_ops
        extra = {}
        if world == 1 and not a.no_cpu_baseline:
            extra["cpu_baseline"] = cpu_baseline_full(H, W)
        print(json.dumps({**extra, "metric": "online refinement steps/sec @640x480", "value": world * nsteps / el, "unit": "steps/s", "n_gpus": world,
                          "steps": nsteps, "warmup": a.warmup * 3, "ms_per_step": 1e3 * el / nsteps, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "full refinement step (depth net fwd+bwd on the pair, median scale, fused warp+photometric+reg, "
                                                 "3-D KNN loss, Adam; map update every 3rd step)", "height": H, "width": W,
                                     "conv_backend": "hip" if nn_ops._use_hip(torch.zeros(1, device=dev)) else "miopen-scaffold",
                                     "map_points": int(slam.map.M)}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 2000 for --workload warp, 30 for full)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default: 200 / 6)")
    ap.add_argument("--batch", type=int, default=1, help="keyframe pairs per launch (reference: 1)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--workload", default="warp", choices=["warp", "full"],
                    help="warp: BASELINE configs[1] (default). full: whole refinement step incl. depth network, 3-D loss, Adam, map update")
    ap.add_argument("--device-geometry", action="store_true", help="derive the geometry in the kernel from device K/inv_K/T (default: host kernel arguments)")
    ap.add_argument("--steps-per-graph", type=int, default=3,
                    help="steps captured per hipGraph replay (default 3 = the refinement steps of one keyframe, README.md:146-158 of the reference)")
    ap.add_argument("--two-kernel", action="store_true", help="lossgrad + second-stage reduce per step instead of the chained single-kernel form")
    ap.add_argument("--grad-only", action="store_true", help="diagnostic: skip the second-stage loss reduction (NOT the benchmark configuration)")
    ap.add_argument("--split", action="store_true", help="two-kernel form (e2e_warp_photo_fwd + _bwd) instead of the single-launch lossgrad")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 2000 if a.workload == "warp" else 30
    if a.warmup is None:
        a.warmup = 200 if a.workload == "warp" else 6

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    if a.workload == "full":
        return full_step_bench(a, rank, world, dev)
    from e2ehip import _lib as L
    from e2ehip.fused import LossGradPlan, WarpPhotoPlan
    from synth import make_pair
    L.load()
    B, H, W = a.batch, a.height, a.width
    s = make_pair(H, W, seed=1234 + rank, B=B)
    g = torch.Generator().manual_seed(5)
    dsrc = s["depth"] + 0.1 * torch.rand(s["depth"].shape, generator=g)
    t = {k: v.to(dev).contiguous() for k, v in s.items()}
    src, tgt = t["src"].permute(0, 3, 1, 2), t["tgt"].permute(0, 3, 1, 2)     # NHWC memory, NCHW view
    plan = (WarpPhotoPlan(B, H, W, dev, "border", True, "l2") if a.split else LossGradPlan(B, H, W, dev, "border", True, "l2", 1.0, 1e-2)).bind(
        t["depth"], dsrc.to(dev), (s["depth"] + 0.05).to(dev), (dsrc + 0.05).to(dev), src, tgt, t["K"], t["invK"], t["T"])
    if not a.split and B == 1 and not a.device_geometry:
        # poses / intrinsics are dataset inputs the host already holds: the pair's 12 geometry numbers go in as kernel arguments
        plan.set_host_geometry(s["K"][0], s["invK"][0], s["T"][0])

    def step():
        if a.split:
            plan.forward()
            plan.backward()
        else:
            plan.step(want_loss=not a.grad_only)

    side = torch.cuda.Stream(dev)
    graph = None
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
        side.synchronize()
        G = 1 if a.no_graph else max(1, a.steps_per_graph)
        graph_g = None
        if not a.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                step()
            if G > 1:                                   # one replay = the G refinement steps of one keyframe
                graph_g = torch.cuda.CUDAGraph()
                chained = not (a.split or a.grad_only or a.two_kernel)
                if chained:
                    step_losses = [torch.zeros_like(plan.loss) for _ in range(G)]
                with torch.cuda.graph(graph_g, stream=side):
                    if chained:
                        # one kernel per step: launch k adds its fixed-point loss sums to slot set k and finalises step k-1's
                        # loss; the last step of the replay is finished by the flush -- all G losses are final at replay end
                        for k in range(G):
                            plan.step_chain(k % 8, (k - 1) % 8 if k else -1, step_losses[k - 1] if k else None)
                        plan.flush_chain((G - 1) % 8, step_losses[G - 1])
                    else:
                        for _ in range(G):
                            step()
        run1 = graph.replay if graph is not None else step

        def run_steps(n):                               # EXACTLY n steps: n // G replays of the G-step graph + the rest one by one
            if graph_g is not None:
                for _ in range(n // G):
                    graph_g.replay()
                n = n % G
            for _ in range(n):
                run1()
        run_steps(a.warmup)

        def barrier():
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        barrier()
        t0 = time.perf_counter()
        run_steps(a.steps)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        barrier()
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())

        # per-kernel launch durations with HIP events on the launch stream (eager, same buffers)
        def kernel_ms(fn, reps=200):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for e0, e1 in ev:
                e0.record(side)
                fn()
                e1.record(side)
            side.synchronize()
            ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
            return sum(ts[reps // 10: -reps // 10]) / len(ts[reps // 10: -reps // 10])
        N = B * H * W
        if a.split:
            fwd_ms = kernel_ms(plan.forward)
            bwd_ms = kernel_ms(plan.backward)
            # algorithmic bytes per launch: fwd reads depth 4N + src 12N + tgt 12N (+ reg 12N), writes synth 12N
            # + valid 4N; bwd reads depth 4N + src 12N + tgt 12N + synth 12N + valid 4N (+ reg 12N), writes g_tgt 4N (+ g_src 4N)
            fwd_bytes, bwd_bytes = (44 + 12) * N, (48 + 16) * N
            dom, dom_ms, dom_bytes = ("e2e_warp_photo_bwd", bwd_ms, bwd_bytes) if bwd_ms >= fwd_ms else ("e2e_warp_photo_fwd(+reduce)", fwd_ms, fwd_bytes)
            launch_ms = {"warp_photo_fwd+reduce": fwd_ms, "warp_photo_bwd": bwd_ms}
            alg = {"fwd": fwd_bytes, "bwd": bwd_bytes}
        else:
            # the dominant kernel ALONE (k_warp_photo_lossgrad, no second-stage reduce).  One HIP event pair per launch
            # over-reports a 11 us kernel by ~2 us (event + eager launch overhead), so the figure used for the roofline
            # is a captured run of 20 back-to-back launches between ONE event pair, divided by 20 -- this is what
            # rocprofv3's average duration agrees with (profiles/r01_final_summaries.md); the per-launch number stays in
            # launch_ms for reference.
            dom_eager_ms = kernel_ms(lambda: plan.step(want_loss=False))
            both_ms = kernel_ms(plan.step)
            dom_ms = dom_eager_ms
            if not a.no_graph:
                RUN = 20
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, stream=side):
                    for _ in range(RUN):
                        plan.step(want_loss=False)
                dom_ms = kernel_ms(gk.replay, reps=60) / RUN
            # single launch: reads depth 4N + src 12N + tgt 12N + reg (init_t, init_s, depth_s) 12N; writes g_tgt 4N + g_src 4N
            dom, dom_bytes = "k_warp_photo_lossgrad", 48 * N
            launch_ms = {"k_warp_photo_lossgrad": dom_ms, "k_warp_photo_lossgrad (one event pair per eager launch)": dom_eager_ms,
                         "lossgrad+reduce (both launches of a step, eager)": both_ms}
            alg = {"lossgrad": dom_bytes, "survey_8d_fused_minimum_equiv": (92 + 24) * N}
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    if rank == 0:
        out = {
            "metric": "online refinement steps/sec @640x480", "value": world * a.steps * B / el, "unit": "steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic 640x480 RGB-D pair, warp+photometric(+l2 depth-reg) fwd+bwd kernels only",
                       "pairs_per_launch": B, "height": H, "width": W, "launch": "eager" if graph is None else f"hipGraph replay, {G} step(s) per replay" + (", chained launches (1 kernel per step + 1 flush per replay)" if (G > 1 and not (a.split or a.grad_only or a.two_kernel)) else ""),
                       "kernels_per_step": 3 if a.split else (2 if (G == 1 or a.two_kernel or a.grad_only) else round(1 + 1 / G, 3)),
                       "geometry": "device matrices" if (a.split or B != 1 or a.device_geometry) else "host kernel arguments"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": TRAFFIC_NOTE.get((a.split, B), None),
                         "launch_ms": launch_ms, "algorithmic_bytes": alg},
        }
        vn = VALU_NOTE.get((a.split, B))
        if vn is not None:
            out["roofline"]["valu_issue"] = dict(vn, frac_of_launch=vn["issue_cycles_per_simd"] / (launch_ms[dom] * 1e-3 * vn["clock_ghz"] * 1e9))
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(H, W)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
