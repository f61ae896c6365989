#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: GPU busy time (union of kernel intervals), per-queue busy time, idle gaps
between consecutive kernels, and the top kernels by summed duration -- over the window of the LAST `frac` of the trace (steady
state).    python tools/trace_timeline.py <kernel_trace.csv> [frac=0.5]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]) for r in rows))
t0 = ev[0][0] + (ev[-1][1] - ev[0][0]) * (1 - frac)
ev = [e for e in ev if e[0] >= t0]
span = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, q, n in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
perq = defaultdict(int)
for s, e, q, n in ev:
    perq[q] += e - s
print(f"window {span / 1e6:.2f} ms, {len(ev)} kernels; GPU busy (union) {busy / 1e6:.2f} ms = {100 * busy / span:.1f} %; idle {100 - 100 * busy / span:.1f} %")
print("sum of kernel durations per queue (ms):", {q: round(v / 1e6, 2) for q, v in perq.items()})
gaps.sort()
if gaps:
    print(f"gaps between busy intervals: n={len(gaps)} median {gaps[len(gaps) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, total {sum(gaps) / 1e6:.2f} ms; >20us: {sum(1 for g in gaps if g > 20000)} totalling {sum(g for g in gaps if g > 20000) / 1e6:.2f} ms")
tot = defaultdict(lambda: [0, 0])
for s, e, q, n in ev:
    tot[n.split("(")[0][:70]][0] += e - s
    tot[n.split("(")[0][:70]][1] += 1
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {n:70s} {c:6d} calls {d / 1e6:8.2f} ms")

# where the GPU waits: idle gaps (no kernel running on any queue) attributed to the kernel that ended before and the one that
# started after, largest totals first
pairs = defaultdict(lambda: [0, 0])
cur_e, last = None, None
for s, e, q, n in ev:
    if cur_e is not None and s > cur_e:
        key = (last.split("(")[0][:40], n.split("(")[0][:40])
        pairs[key][0] += s - cur_e
        pairs[key][1] += 1
    if cur_e is None or e > cur_e:
        cur_e, last = e, n
print("idle gaps by (kernel before -> kernel after):")
for (a, b), (d, c) in sorted(pairs.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {a:40s} -> {b:40s} {c:6d} gaps {d / 1e6:8.2f} ms  avg {d / c / 1e3:7.1f} us")

# optional: dump the launches around the LAST occurrence of a kernel (substring match): argv[3] = name, argv[4] = how many before/after
if len(sys.argv) > 3:
    name, span_n = sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 30
    idx = max(i for i, e in enumerate(ev) if name in e[3])
    t_ref = ev[max(0, idx - span_n)][0]
    print(f"launches around the last {name}:  start_us end_us dur_us queue name")
    for s, e, q, n in ev[max(0, idx - span_n): idx + span_n]:
        print(f"  {(s - t_ref) / 1e3:9.1f} {(e - t_ref) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q}  {n.split('(')[0][:60]}")
