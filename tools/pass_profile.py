#!/usr/bin/env python3
"""Where a whole pass of the sequence spends its time: wall time of every keyframe (3 refinement steps + map update; one device sync per
keyframe, so the figures read ~1 % slower than bench.py's unsynchronised pass) next to the map size, and -- at a few keyframes -- the
event-timed C-ABI entry points that grow with the map (KNN index build / query, PointFusion association / fuse-append).

    python tools/pass_profile.py [--probe 3,15,30,45,58] > profiles/rNN_pass_profile.txt

Same workload construction as bench.py's `seq` (BASELINE configs[2], synthetic data, NET_SEED)."""
import argparse
import contextlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (adds the package directory to sys.path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--probe", default="3,15,30,45,58", help="keyframes (1-based) whose entry points are event-timed in a SECOND pass over the same schedule")
    ap.add_argument("--seq-len", type=int, default=60)
    ap.add_argument("--gc", default="default", choices=["default", "freeze", "off"], help="Python's cyclic collector during the passes")
    a = ap.parse_args()
    import torch
    from e2ehip.profile import KernelTimer
    from e2ehip.synthetic import make_sequence
    from online_adaption import SLAM, default_config
    H, W, L = 480, 640, a.seq_len
    dev = torch.device("cuda", 0)
    cfg = default_config(H, W, L)
    cfg.DEBUG.print_metrics = False
    cfg.MODEL.odom = "gt"
    cfg.DATA.name = "ICL"
    cfg.DEMO.frame_threshold = 0.05
    seq = make_sequence(L, H, W, seed=1234, step=0.06, K=None, holes=0.0, scene="plane")
    torch.manual_seed(bench.NET_SEED)
    with contextlib.redirect_stdout(sys.stderr):
        slam = SLAM(cfg, sequence=seq)
    slam.set_refinement_mode()
    slam.first_iter = True
    sched = slam.keyframe_schedule()
    probes = {int(x) for x in a.probe.split(",") if x}
    # where a slow keyframe comes from: the device time between two events around it (GPU-side work only) next to the wall time, and the
    # time the host spent inside Python's cyclic garbage collector during it (a gen-2 collection scans every live object of the process:
    # tens of milliseconds with a network, plans and graphs alive -- and the GPU idles meanwhile: a keyframe is 4 graph replays the host issues)
    import gc
    gc_log = {"t0": 0.0, "spent": 0.0, "events": []}

    def gc_cb(phase, info):
        if phase == "start":
            gc_log["t0"] = time.perf_counter()
        else:
            dt = time.perf_counter() - gc_log["t0"]
            gc_log["spent"] += dt
            gc_log["events"].append((info["generation"], 1e3 * dt))
    gc.callbacks.append(gc_cb)
    if a.gc == "freeze":
        gc.collect()
        gc.freeze()
    elif a.gc == "off":
        gc.disable()
    print(f"# {bench.source_stamp()}  pass profile, {len(sched)} keyframes, 3 steps each")
    for p in range(2):
        if p:
            slam.reset_map()
        print(f"# pass {p}: " + ("wall time per keyframe (sync after each)" if p == 0 else "event-timed entry points at the probed keyframes (refined network, map rebuilt)"))
        for i, pair in enumerate(sched):
            nxt = sched[i + 1] if i + 1 < len(sched) else None
            m_before = int(slam.map.count[0])
            torch.cuda.synchronize(dev)
            gc_log["spent"], gc_log["events"] = 0.0, []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0 = time.perf_counter()
            if p == 1 and (i + 1) in probes:
                with KernelTimer() as kt:
                    slam.refinement(*pair, max_steps=3, next_pair=nxt)
                torch.cuda.synchronize(dev)
                rows = kt.summary()
                tot = sum(r["ms"] for r in rows.values())
                grow = {n: r for n, r in rows.items() if "knn" in n or "_pf_" in n or "vertex" in n}
                print(f"keyframe {i + 1:3d}  map {m_before:9d}  kernels {tot:7.3f} ms  " + "  ".join(f"{n.replace('e2e_', '')} {r['ms']:.3f}/{r['calls']}" for n, r in sorted(grow.items())))
            else:
                slam.refinement(*pair, max_steps=3, next_pair=nxt)
                e1.record()
                torch.cuda.synchronize(dev)
                if p == 0:
                    wall = 1e3 * (time.perf_counter() - t0)
                    gcs = "" if not gc_log["events"] else "  gc " + " ".join(f"gen{g}:{ms:.2f}ms" for g, ms in gc_log["events"])
                    print(f"keyframe {i + 1:3d}  map {m_before:9d}  wall {wall:7.3f} ms  device {e0.elapsed_time(e1):7.3f} ms{gcs}")
            slam.first_iter = False
    sys.stdout.flush()


if __name__ == "__main__":
    main()
