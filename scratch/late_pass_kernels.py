"""Per-kernel durations of the map-size dependent kernels at the END of a whole pass (map ~11.7 M points), from a rocprofv3 kernel trace of
`bench.py --steps 177 --warmup 6`: average of the last 4 calls of each kernel next to the average of calls 3..6 (map ~0.7 M points).
usage: python3 scratch/late_pass_kernels.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if any(k in n for k in ("k_grid", "k_scan", "k_pf_", "k_cp_", "k_knn1", "k_gather_active", "k_active", "k_fill_u")):
        per[n.split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':62s} calls   early_us    late_us")
for n, d in sorted(per.items(), key=lambda kv: -sum(kv[1][-4:])):
    e = d[3:7] if len(d) > 10 else d[:1]
    print(f"{n:62s} {len(d):5d} {sum(e)/len(e):10.1f} {sum(d[-4:])/len(d[-4:]):10.1f}")
for n in ("k_knn1_rest", "k_grid_query"):
    d = per.get(n, [])
    print(n, "last 12 calls (us):", " ".join(f"{v:.0f}" for v in d[-12:]), "| calls 10..21:", " ".join(f"{v:.0f}" for v in d[9:21]))
