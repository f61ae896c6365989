#!/usr/bin/env python3
"""bench.py -- online refinement steps/sec of the MI355X hot path (DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload seq|warp] [--odom gt|icp|gradicp] [--no-cpu-baseline]

Default workload = BASELINE.json configs[2]: a synthetic 640x480 RGB-D sequence of seq_len = 60 run through the build's
online_adaption.SLAM with the keyframe schedule (every frame of the synthetic orbit is a keyframe), 3 refinement steps per
keyframe -- depth network forward + backward on the keyframe pair, median scaling, fused warp + photometric + depth
regulariser, 3-D nearest-neighbour loss against the global map, Adam -- and the PointFusion map update after every
keyframe.  One "step" = one refinement step (one iteration of online_adaption.py:274); the map updates run inside the
timed region (one per 3 steps) but are not counted as steps.  EXACTLY K steps are timed after W untimed ones.

N > 1: one process per GPU (the driver launches them with torch.distributed.run; run by hand with --gpus N and no RANK in
the environment this file spawns them itself BEFORE touching the GPU).  Each rank refines its own sequence (seed 1234 +
rank) against its own map; the ranks share the depth network through ONE all-reduce of the flat 57.3 MB gradient bucket
per refinement step (RCCL over xGMI), and gather their maps at the end.  value = steps of all ranks / max-over-ranks time.

--workload warp keeps the round-1 kernel-only workload (BASELINE configs[1]: warp + photometric kernels on one pair).
"""
import argparse
import contextlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA, dense (= the fp32 vector peak)
CONV_ENTRY_POINTS = ("e2e_conv2d_fwd", "e2e_conv2d_bwd_data", "e2e_conv2d_bwd_data_acc", "e2e_conv2d_bwd_data_fused", "e2e_conv2d_bwd_weight",
                     "e2e_conv2d_bwd_weight_scaled", "e2e_conv2d_bwd_weight_scaled_deferred")
# launches that belong to the family's TIME but carry no FLOPs and are no convolution call of their own: the slab reductions the deferred
# backward-weight calls leave to the end of a backward pass
CONV_TAIL_ENTRY_POINTS = ("e2e_wgrad_reduce_batched",)
WARP_ENTRY_POINTS = ("e2e_warp_photo_lossgrad_hostgeo", "e2e_warp_photo_lossgrad", "e2e_warp_photo_lossgrad_chain")
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "r04_bench_pmc_traffic.json")
WHOLE_PASS = os.path.join(ROOT, "profiles", "r04_bench_seq_fullpass.json")     # `--steps 177 --warmup 6` line of tools/evidence.sh
NET_SEED = 20241004         # seed of the depth network's random initialisation (there are no pretrained weights on the GPU box)


def source_stamp():
    """sha256 over the sources that decide which kernels a refinement step launches and with what arguments (csrc/*, the launch plan).
    tools/bench_pmc.sh stores it next to the PMC traffic it measures; a mismatch means the committed traffic figure is stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")))
    files += [os.path.join(pkg, "e2ehip", f) for f in ("netplan.py", "stepplan.py", "fusionmap.py", "ops.py")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed refinement steps (default: one pass of the sequence = 3 x (seq_len - 1))")
    ap.add_argument("--warmup", type=int, default=None, help="untimed refinement steps (default 6)")
    ap.add_argument("--workload", default="seq", choices=["seq", "warp"],
                    help="seq: BASELINE configs[2], whole refinement steps over a 60-frame sequence (default). warp: configs[1] kernels only")
    ap.add_argument("--seq-len", type=int, default=60)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--odom", default="gt", choices=["gt", "icp", "gradicp"], help="MODEL.odom of the map step (gradicp: reports the ATE)")
    ap.add_argument("--tum", action="store_true", help="TUM-shaped sequence (fx = fy = 525, 10 %% zero-depth holes, threshold 0.12): configs[3]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel event-timing pass")
    ap.add_argument("--dry", action="store_true", help="CPU rehearsal of the multi-process path (gloo, no GPU, no kernels): launch / exchange / gather only")
    ap.add_argument("--print-stamp", action="store_true", help="print source_stamp() and exit (tools/bench_pmc.sh)")
    ap.add_argument("--same-sequence", action="store_true", help="[N > 1] every rank refines the SAME sequence (seed 1234): the averaged update must then "
                                                                 "equal the one-rank update -- a correctness check of the gradient exchange, not a benchmark")
    # round-1 kernel-only workload
    ap.add_argument("--batch", type=int, default=1, help="[warp] keyframe pairs per launch (reference: 1)")
    ap.add_argument("--no-graph", action="store_true", help="[warp] eager launches instead of hipGraph replay")
    ap.add_argument("--steps-per-graph", type=int, default=3, help="[warp] steps captured per hipGraph replay")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# multi-process launch
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(a):
    """--gpus N with no launcher environment: start N ranks (one per GPU) and relay rank 0's JSON line.  Runs before any
    GPU call in this process (a process that has touched the GPU must not exec / fork workers on this pool)."""
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = CPU restatement, kind "port"), rank 0 at N = 1 only
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline_seq(H, W, tum=False):
    """One refinement step of the same workload on this node's host cores: (i) >= 5 steps after 2 warm-ups of the oracle's
    refinement step WITHOUT the 3-D loss (depth network fwd+bwd on the pair, median scale, warp + photometric + regulariser,
    Adam) + (ii) the 3-D loss's nearest-neighbour search timed on 4 096 query rows against a map of the size the GPU run
    sees after two keyframes and scaled linearly to the 307 200 rows of a frame (SURVEY.md 8d iii: the full search takes
    minutes per step) + (iii) the map update (unprojection + PointFusion association / fuse) once, divided by the 3 steps
    of a keyframe."""
    import torch
    from e2ehip.synthetic import make_sequence, tum_intrinsics
    from oracle import depthnet, knn, pointfusion, refine, warp_loss
    ncpu = os.cpu_count() or 1
    cores = min(16, ncpu)           # torch's intra-op pool oversubscribes badly on a many-core host (21 s/step at 256 threads)
    torch.set_num_threads(cores)
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    colors, depths, intr, poses = make_sequence(3, H, W, seed=1234, K=tum_intrinsics(H, W) if tum else None, holes=0.1 if tum else 0.0)
    cfg = refine.Config()
    cfg.three3d_loss = False
    cfg.refinement_steps = 7
    r = refine.Refiner(depthnet.random_state_dict(0), cfg)
    stamps = []
    orig = r.opt.step

    def step_and_stamp(*x, **k):
        out = orig(*x, **k)
        stamps.append(time.perf_counter())
        return out
    r.opt.step = step_and_stamp
    t0 = time.perf_counter()
    r.refine_pair(colors[:, :2], depths[:, :2], poses[:, :2], intr, update_map=False)
    t_steps = (stamps[-1] - stamps[1]) / 5.0                       # steps 3..7: 5 steps after 2 warm-ups
    # (iii) map update: two frames into an empty map = what the first keyframe does
    t1 = time.perf_counter()
    with torch.no_grad():
        d2, _ = warp_loss.median_scale([1 / depthnet.disp_forward(r.sd, colors[:, i]) for i in range(2)], depths[:, :2])
        state = pointfusion.empty_state()
        for i in range(2):
            state, _ = pointfusion.pointfusion_step(state, colors[0, i], d2[i][0, 0], intr[0, 0], poses[0, i])
    t_map = time.perf_counter() - t1
    # (ii) nearest neighbours: 4 096 rows of the next frame's cloud against that map
    M = state["points"].shape[0]
    maps = pointfusion.vertex_normal_maps(d2[1][0, 0], intr[0, 0], poses[0, 1])
    q = maps["Vg"].reshape(-1, 3)[torch.randperm(H * W, generator=torch.Generator().manual_seed(0))[:4096]].contiguous()
    knn.knn1(q[:64], state["points"])
    t2 = time.perf_counter()
    knn.knn1(q, state["points"])
    t_knn = (time.perf_counter() - t2) * (H * W / 4096.0)
    step_s = t_steps + t_knn + t_map / 3.0
    return {"value": 1.0 / step_s, "unit": "steps/s", "cores": cores, "kind": "port", "host_cpus": ncpu,
            "seconds": {"step_without_3d_loss": t_steps, "knn_scaled": t_knn, "map_update_per_keyframe": t_map, "total_bounded_sample": time.perf_counter() - t0},
            "sample": f"oracle refinement step at 1x{H}x{W}: 5 steps after 2 warm-ups without the 3-D loss ({t_steps:.2f} s/step) + brute-force 1-NN of 4096 "
                      f"query rows against a {M}-point map scaled x{H * W / 4096:.0f} to the frame's {H * W} rows ({t_knn:.1f} s/step) + one 2-frame PointFusion "
                      f"map update / 3 ({t_map:.2f} s per keyframe); torch CPU + OpenMP C loop, {cores} threads"}


# ---------------------------------------------------------------------------------------------------------------------
# default workload: BASELINE configs[2]
# ---------------------------------------------------------------------------------------------------------------------
def seq_bench(a, rank, world, dev):
    import torch
    import torch.distributed as dist
    from e2ehip import dist as edist
    from e2ehip.profile import KernelTimer
    from e2ehip.synthetic import make_sequence, tum_intrinsics
    from online_adaption import SLAM, default_config
    H, W, L = a.height, a.width, a.seq_len
    spk = 3
    cfg = default_config(H, W, L)
    cfg.DEBUG.print_metrics = False
    cfg.MODEL.odom = a.odom
    cfg.DATA.name = "TUM" if a.tum else "ICL"
    cfg.DEMO.frame_threshold = 0.12 if a.tum else 0.05             # README.md:157 of the reference
    step_m = 0.13 if a.tum else 0.06                               # camera step of the synthetic orbit: every frame is a keyframe
    seq = make_sequence(L, H, W, seed=1234 + (0 if a.same_sequence else rank), step=step_m, K=tum_intrinsics(H, W) if a.tum else None, holes=0.1 if a.tum else 0.0,
                        scene="corner" if a.odom != "gt" else "plane")
    K = a.steps if a.steps is not None else spk * (L - 1)
    Wm = a.warmup if a.warmup is not None else 6
    torch.manual_seed(NET_SEED)                         # the network initialisation is part of the workload: map size (KNN / association cost) follows it
    with contextlib.redirect_stdout(sys.stderr):        # the driver mirrors the reference's start-up prints; stdout carries ONE JSON line
        slam = SLAM(cfg, sequence=seq)
    slam.set_refinement_mode()
    slam.first_iter = True
    sched = slam.keyframe_schedule()
    if edist.data_parallel():                            # N > 1 ranks, or the one-rank exchange rehearsal (E2E_FORCE_EXCHANGE=1)
        slam.optimizer.prebuild(slam.models["depth"].used_parameters())
    state = {"i": 0, "passes": 0, "visited": []}

    def run_steps(n):                                   # EXACTLY n refinement steps, continuing along the keyframe schedule
        while n > 0:
            if state["i"] >= len(sched):                # sequence exhausted: next pass over it with the refined network
                slam.reset_map()
                state["i"], state["passes"] = 0, state["passes"] + 1
            k = min(spk, n)
            state["visited"].append((state["passes"], state["i"], slam.map.count[0:1].clone()))    # (pass, keyframe, map size before the keyframe:
                                                                                                 #  a device-side copy, read after the timed region)
            nxt = sched[state["i"] + 1] if state["i"] + 1 < len(sched) else None
            slam.refinement(*sched[state["i"]], max_steps=k, next_pair=nxt)
            slam.first_iter = False
            state["i"] += 1
            n -= k

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run_steps(Wm)
    barrier()
    done0 = slam.refinement_steps_done
    v0 = len(state["visited"])
    t0 = time.perf_counter()
    run_steps(K)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    barrier()
    assert slam.refinement_steps_done - done0 == K
    timed = state["visited"][v0:]
    covered = {"first": {"pass": timed[0][0], "keyframe": timed[0][1] + 1, "map_points_before": int(timed[0][2])},
               "last": {"pass": timed[-1][0], "keyframe": timed[-1][1] + 1, "map_points_before": int(timed[-1][2])},
               "keyframes_in_timed_region": len(timed), "of_keyframes_per_pass": len(sched),
               "note": "keyframe k = k-th pair of the schedule (1-based); KNN / association cost grows with the map, so a short timed region "
                       "that starts at keyframe 3 reads faster than a whole pass (value_whole_pass: profiles/r04_bench_seq_fullpass.json = --steps 177)"}
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    map_points = int(slam.map.M)
    ate = slam.absolute_trajectory_error() if a.odom != "gt" else None
    odo_info = None
    if a.odom != "gt" and getattr(slam, "_odo", None) is not None:
        its, _, ntgt, nact = slam._odo.check()
        odo_info = {"iterations_last_keyframe": its, "target_points": ntgt, "active_map_points": nact, "source_points": slam._odo.n_src,
                    "grid_cells_per_axis": slam._odo.cells}
    replicas_identical = True
    if world > 1:                                       # data parallel: the shared depth network must be bit-identical on every rank
        cs = slam.optimizer.flat.data.double().sum().reshape(1).to("cpu" if dist.get_backend() == "gloo" else dev)
        allcs = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(allcs, cs)
        replicas_identical = all(float(c) == float(allcs[0]) for c in allcs)
        if not replicas_identical:
            print(f"ERROR: ranks disagree about the network parameters: {[float(c) for c in allcs]}", file=sys.stderr)

    # ---- per-kernel durations of ONE more keyframe, measured live with HIP events on the launch streams --------------
    roof = {}
    if not a.no_roofline:
        overlap, slam.overlap_wgrad = slam.overlap_wgrad, False     # one stream: a kernel's duration is its own, not a shared GPU's
        with KernelTimer() as kt:
            run_steps(spk)
        slam.overlap_wgrad = overlap
        rows = kt.summary()
        conv_ms = sum(rows[n]["ms"] for n in CONV_ENTRY_POINTS + CONV_TAIL_ENTRY_POINTS if n in rows)
        conv_fl = sum(rows[n]["flops"] for n in CONV_ENTRY_POINTS if n in rows)
        conv_calls = sum(rows[n]["calls"] for n in CONV_ENTRY_POINTS if n in rows)
        conv_bytes = sum(rows[n]["bytes"] for n in CONV_ENTRY_POINTS if n in rows)
        warp = [rows[n] for n in WARP_ENTRY_POINTS if n in rows]
        all_ms = sum(r["ms"] for r in rows.values())
        if conv_ms > 0:
            tf = conv_fl / (conv_ms * 1e-3) / 1e12
            # HBM-side bytes per C-ABI convolution call from the committed PMC passes of this same workload (tools/bench_pmc.sh; counters
            # cannot be read live): bytes of all conv-family kernels per refinement step x steps per keyframe / calls per keyframe
            traffic, traffic_note, pmc = None, "no PMC record", PMC_TRAFFIC
            if os.path.isfile(pmc):
                with open(pmc) as f:
                    t = json.load(f)
                if t.get("source_stamp") == source_stamp():
                    traffic = t["conv_gemm_family_bytes_total"] / t["refinement_steps"] * spk / max(conv_calls, 1)
                    traffic_note = (f"bytes per C-ABI convolution call from {os.path.relpath(pmc, ROOT)} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                    f"this workload over {t['refinement_steps']} steps, taken at source stamp {t['source_stamp']} = this build)")
                else:
                    traffic_note = f"{os.path.relpath(pmc, ROOT)} was taken at source stamp {t.get('source_stamp')}, this build is {source_stamp()}: stale, not reported"
            roof["roofline"] = {
                "bound": "mfma", "kernel": "depth-network convolution GEMMs (k_conv_gemm forward / backward-data, k_wgrad_gemm* backward-weight incl. their "
                                           "split-K / slab reductions), fp32 v_mfma_f32_32x32x2_f32",
                "achieved": tf, "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFS, "traffic": traffic,
                "traffic_note": traffic_note,
                "traffic_algorithmic": conv_bytes / max(conv_calls, 1),
                "traffic_algorithmic_note": "compulsory bytes per C-ABI convolution call: input domain + weights + result, each once (fp32); `traffic` "
                                            "additionally holds the split-K / backward-weight slabs (written, then read by the reduction kernels) and one "
                                            "copy of a launch's weight slices per XCD L2",
                "launches": conv_calls, "avg_launch_us": 1e3 * conv_ms / max(conv_calls, 1), "algorithmic_gflop_per_keyframe": conv_fl / 1e9,
                "ms_per_keyframe": conv_ms, "share_of_event_timed_kernel_time": conv_ms / all_ms,
                "by_entry_point": {n: {"calls": rows[n]["calls"], "ms": round(rows[n]["ms"], 4), "tflops": rows[n]["flops"] / (rows[n]["ms"] * 1e-3) / 1e12}
                                   for n in CONV_ENTRY_POINTS + CONV_TAIL_ENTRY_POINTS if n in rows},
                "method": "HIP events around every C-ABI call of one keyframe (3 steps + map update) after the timed region, on the stream each call runs on, "
                          "backward-weight overlap off; ~2 us of event overhead per call is included"}
        if warp and slam.step_plan is not None:
            # the fused loss kernel alone: 20 back-to-back launches on the step's own resident buffers inside ONE captured graph between ONE
            # event pair (an event pair per launch adds ~6 us to a ~10 us kernel; rocprofv3's per-kernel average agrees with this figure)
            wby, RUN, lp = warp[0]["bytes"] // max(warp[0]["calls"], 1), 20, slam.step_plan.loss
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                lp.step()
                side.synchronize()
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, stream=side):
                    for _ in range(RUN):
                        lp.step()
                ts = []
                for _ in range(30):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                    gk.replay()
                    e1.record(side)
                    side.synchronize()
                    ts.append(e0.elapsed_time(e1) / RUN)
            torch.cuda.current_stream(dev).wait_stream(side)
            ts.sort()
            wms = sum(ts[5:-5]) / len(ts[5:-5])
            gbs = wby / (wms * 1e-3) / 1e9
            roof["roofline_warp"] = {"bound": "hbm", "kernel": "k_warp_photo_lossgrad (+ its loss finalisation)", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": 1e3 * wms, "algorithmic_bytes_per_launch": wby,
                                     "method": f"{RUN} back-to-back launches per captured graph between one event pair, trimmed mean of 30 replays"}
        roof["kernel_time_ms_per_keyframe"] = {n: round(r["ms"], 4) for n, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:16]}

    # ---- odometry against the oracle on the SAME inputs (outside the timed region; --odom icp | gradicp) ------------------
    # "ATE vs reference": for a few more keyframes the inputs of the frame-to-model odometry (map before the step, predicted depth, previous
    # pose) are snapshotted, the oracle (oracle/icp.py on the host cores) registers the same frame against the same map, and both poses
    # are scored against the dataset pose -- two trajectory errors over one prefix, plus the largest difference between the two estimates
    odom_parity = None
    if a.odom != "gt" and world == 1 and not a.no_cpu_baseline:
        from oracle import icp as oicp
        rows = []
        for _ in range(3):
            if state["i"] >= len(sched):
                slam.reset_map()
                state["i"], state["passes"] = 0, state["passes"] + 1
            prev, cur = sched[state["i"]]
            pts, nrm = (t.clone().cpu() for t in slam.map.live()[:2]) if not slam.first_iter else (None, None)
            nxt = sched[state["i"] + 1] if state["i"] + 1 < len(sched) else None
            slam.refinement(prev, cur, next_pair=nxt)
            slam.first_iter = False
            state["i"] += 1
            if pts is None:
                continue
            torch.cuda.synchronize(dev)
            depth = slam.step_plan.depth[1, 0].cpu()
            est, gt_pose = slam.estimated_poses[-1][0].cpu().double().numpy(), slam.poses[0, cur].cpu().double().numpy()
            t_o = time.perf_counter()
            ref, _ = oicp.frame_to_model(pts, nrm, depth, slam.intrinsics[0, 0].cpu(), slam.poses[0, prev].cpu(), mode=a.odom, numiters=cfg.MODEL.numiters)
            rows.append({"keyframe": state["i"], "map_points": int(pts.shape[0]), "err_gpu_m": float(((est[:3, 3] - gt_pose[:3, 3]) ** 2).sum() ** 0.5),
                         "err_oracle_m": float(((ref[:3, 3] - gt_pose[:3, 3]) ** 2).sum() ** 0.5), "max_abs_pose_diff": float(abs(est - ref).max()),
                         "oracle_seconds": time.perf_counter() - t_o})
        if rows:
            rms = lambda k: float((sum(r[k] ** 2 for r in rows) / len(rows)) ** 0.5)
            odom_parity = {"keyframes": rows, "ate_m_gpu": rms("err_gpu_m"), "ate_m_oracle": rms("err_oracle_m"),
                           "max_abs_pose_diff": max(r["max_abs_pose_diff"] for r in rows),
                           "note": "same inputs on both sides (map before the step, the GPU's predicted depth, previous dataset pose): oracle/icp.py frame_to_model "
                                   "on the host vs e2ehip.icp.ResidentOdometry inside the captured map step; errors against the dataset pose"}

    # ---- end of run: per-rank map sizes and the map gather (outside the timed region) ---------------------------------
    sizes, gathered = [map_points], map_points
    if world > 1:
        t1 = time.perf_counter()
        g = edist.gather_maps(*slam.map.live(), dst=0)                 # exact sizes, to rank 0 only
        torch.cuda.synchronize(dev)
        sizes, gathered = [int(c) for c in g[4]], int(sum(int(c) for c in g[4]))
        if rank == 0 and int(g[0].shape[0]) != gathered:
            raise RuntimeError("map gather: rank 0 received a different number of points than the ranks reported")
        roof["map_gather_ms"] = 1e3 * (time.perf_counter() - t1)
    if rank == 0:
        out = {"metric": "online refinement steps/sec @640x480, seq_len=60", "value": world * K / el, "unit": "steps/s", "n_gpus": world, "steps": K,
               "warmup": Wm, "ms_per_step": 1e3 * el / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": f"BASELINE configs[{3 if a.tum else 2}]: synthetic {'TUM-shaped (fx=fy=525, 10% depth holes) ' if a.tum else ''}{W}x{H} RGB-D sequence, seq_len={L}, "
                                      "online_adaption.SLAM keyframe schedule, 3 refinement steps/keyframe (depth net fwd+bwd on the pair, median scale, fused "
                                      "warp+photometric+depth-reg, 3-D KNN loss vs the global map, Adam) + PointFusion map update per keyframe"
                                      + (", one sequence per GPU, 1 all-reduce of the 57.3 MB gradient bucket per step" if world > 1 else ""),
                          "height": H, "width": W, "seq_len": L, "keyframes_per_pass": len(sched), "refinement_steps_per_keyframe": spk, "odom": a.odom,
                          "sequence_passes_started": state["passes"] + 1, "map_points_rank0": map_points, "map_points_per_rank": sizes,
                          "map_points_gathered": gathered, "ate_m": ate, "odometry": odo_info, "replicas_identical": replicas_identical, "keyframes_covered": covered,
                          "network_seed": NET_SEED, "parameter_checksum": float(slam.optimizer.flat.data.double().sum()) if slam.optimizer.flat is not None else None,
                          "same_sequence_on_every_rank": bool(a.same_sequence),
                          "exchange_forced_on_one_rank": world == 1 and edist.data_parallel()}}
        if not replicas_identical:                      # a data-parallel run whose replicas diverged measured nothing
            out["value"] = None
            out["error"] = "replicas diverged: parameter checksums differ between ranks"
        out.update(roof)
        if odom_parity is not None:
            out["odometry_vs_oracle"] = odom_parity
        out["source_stamp"] = source_stamp()
        # the default timed region (20 steps from keyframe 3, a small map) reads faster than a whole pass of the 60-frame sequence: the
        # whole-pass figure travels with this line, with the build it was measured on (tools/evidence.sh runs it first)
        if world == 1 and a.workload == "seq" and os.path.isfile(WHOLE_PASS):
            try:
                with open(WHOLE_PASS) as f:
                    wp = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])
                out["value_whole_pass"] = {"value": wp["value"], "unit": wp["unit"], "steps": wp["steps"], "ms_per_step": wp["ms_per_step"],
                                           "keyframes_covered": wp["config"].get("keyframes_covered"), "source_stamp": wp.get("source_stamp"),
                                           "same_build_as_this_line": wp.get("source_stamp") == out["source_stamp"],
                                           "file": os.path.relpath(WHOLE_PASS, ROOT), "command": "python3 bench.py --gpus 1 --steps 177 --warmup 6 --no-cpu-baseline"}
            except Exception as e:                      # a malformed record must not cost the bench line
                out["value_whole_pass"] = {"error": repr(e)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_seq(H, W, a.tum)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    elif edist.data_parallel():
        dist.destroy_process_group()
    if not replicas_identical:
        sys.exit(3)


# ---------------------------------------------------------------------------------------------------------------------
# CPU rehearsal of the multi-process path (tests/test_dist_gloo.py): launch, bucket exchange, schedule agreement, gather
# ---------------------------------------------------------------------------------------------------------------------
def dry_run(a, rank, world):
    import torch
    import torch.distributed as dist
    from e2ehip import dist as edist
    from e2ehip.optim import FlatParams
    if world > 1:
        dist.init_process_group("gloo")
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(33, 7)), torch.nn.Parameter(torch.randn(5))]
    flat = FlatParams(ps)
    rounds = edist.common_rounds(2 + rank, torch.device("cpu"))         # rank r has 2 + r keyframes
    K = a.steps or 4
    t0 = time.perf_counter()
    for i in range(K):
        flat.zero_grad()
        mine = (i // 3) < 2 + rank
        if mine:
            sum(((rank + 1.0) * p).sum() for p in ps).backward()
        edist.exchange_gradients_(flat, participating=mine)
    el = time.perf_counter() - t0
    cnt = float(flat.participants())
    g = edist.gather_maps(torch.full((rank + 2, 3), float(rank)), torch.ones(rank + 2, 3), torch.zeros(rank + 2, 3), torch.ones(rank + 2))
    if rank == 0:
        print(json.dumps({"metric": "online refinement steps/sec @640x480, seq_len=60", "value": world * K / el, "unit": "steps/s", "n_gpus": world,
                          "steps": K, "warmup": 0, "ms_per_step": 1e3 * el / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic", "dry": True,
                          "config": {"workload": "DRY (CPU, gloo): launch + gradient-bucket exchange + map gather only", "rounds": rounds,
                                     "participants_last_step": cnt, "map_points_per_rank": [int(c) for c in g[4]], "map_points_gathered": int(g[0].shape[0])}}))
    if world > 1:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# round-1 workload: BASELINE configs[1], warp + photometric kernels only
# ---------------------------------------------------------------------------------------------------------------------
def warp_bench(a, rank, world, dev):
    import torch
    import torch.distributed as dist
    from e2ehip import _lib as L
    from e2ehip.fused import LossGradPlan
    from synth import make_pair
    L.load()
    steps = a.steps if a.steps is not None else 2000
    warm = a.warmup if a.warmup is not None else 200
    B, H, W = a.batch, a.height, a.width
    s = make_pair(H, W, seed=1234 + rank, B=B)
    g = torch.Generator().manual_seed(5)
    dsrc = s["depth"] + 0.1 * torch.rand(s["depth"].shape, generator=g)
    t = {k: v.to(dev).contiguous() for k, v in s.items()}
    src, tgt = t["src"].permute(0, 3, 1, 2), t["tgt"].permute(0, 3, 1, 2)     # NHWC memory, NCHW view
    plan = LossGradPlan(B, H, W, dev, "border", True, "l2", 1.0, 1e-2).bind(
        t["depth"], dsrc.to(dev), (s["depth"] + 0.05).to(dev), (dsrc + 0.05).to(dev), src, tgt, t["K"], t["invK"], t["T"])
    if B == 1:
        plan.set_host_geometry(s["K"][0], s["invK"][0], s["T"][0])
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        for _ in range(3):
            plan.step()
        side.synchronize()
        G = 1 if a.no_graph else max(1, a.steps_per_graph)
        graph = graph_g = None
        if not a.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                plan.step()
            if G > 1:
                graph_g = torch.cuda.CUDAGraph()
                step_losses = [torch.zeros_like(plan.loss) for _ in range(G)]
                with torch.cuda.graph(graph_g, stream=side):
                    for k in range(G):
                        plan.step_chain(k % 8, (k - 1) % 8 if k else -1, step_losses[k - 1] if k else None)
                    plan.flush_chain((G - 1) % 8, step_losses[G - 1])
        run1 = graph.replay if graph is not None else plan.step

        def run_steps(n):
            if graph_g is not None:
                for _ in range(n // G):
                    graph_g.replay()
                n = n % G
            for _ in range(n):
                run1()
        run_steps(warm)

        def barrier():
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
        barrier()
        t0 = time.perf_counter()
        run_steps(steps)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        barrier()
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        # the dominant kernel alone: a captured run of 20 back-to-back launches between ONE event pair (one pair per eager launch
        # over-reports an 11 us kernel by ~2 us); rocprofv3's average agrees with this figure
        RUN = 20
        gk = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gk, stream=side):
            for _ in range(RUN):
                plan.step(want_loss=False)
        ts = []
        for _ in range(60):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            gk.replay()
            e1.record(side)
            side.synchronize()
            ts.append(e0.elapsed_time(e1) / RUN)
        ts.sort()
        dom_ms = sum(ts[6:-6]) / len(ts[6:-6])
    N = B * H * W
    achieved = 48 * N / (dom_ms * 1e-3) / 1e9
    if rank == 0:
        print(json.dumps({
            "metric": "online refinement steps/sec @640x480", "value": world * steps * B / el, "unit": "steps/s", "n_gpus": world, "steps": steps,
            "warmup": warm, "ms_per_step": 1e3 * el / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic 640x480 RGB-D pair, warp+photometric(+l2 depth-reg) fwd+bwd kernels only",
                       "pairs_per_launch": B, "height": H, "width": W,
                       "launch": "eager" if graph is None else f"hipGraph replay, {G} step(s) per replay" + (", chained launches" if G > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": "k_warp_photo_lossgrad", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": 1e3 * dom_ms, "algorithmic_bytes_per_launch": 48 * N}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    a = parse_args()
    if a.print_stamp:
        print(source_stamp())
        return
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.dry:
        return dry_run(a, rank, world)
    import torch
    # E2E_REHEARSE_ONE_GPU=1: every rank on cuda:0 with gloo as transport (the gradient bucket is staged through the host) -- a rehearsal
    # of the N-rank GPU code path (graphs split around the exchange, idle rounds, map gather) on a one-GPU box; never a measurement
    rehearse = os.environ.get("E2E_REHEARSE_ONE_GPU") == "1"
    dev = torch.device("cuda", 0 if rehearse else local)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif os.environ.get("E2E_FORCE_EXCHANGE") == "1":
        # ONE rank through the N-rank step over the real RCCL transport (e2ehip.dist.data_parallel): a correctness rehearsal, not a benchmark
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29500 + os.getpid() % 2000}", rank=0, world_size=1, device_id=dev)
    if a.workload == "warp":
        return warp_bench(a, rank, world, dev)
    return seq_bench(a, rank, world, dev)


if __name__ == "__main__":
    main()
