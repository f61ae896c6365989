"""python3 scratch/stats_table.py <kernel_stats.csv> <steps>: per-kernel calls, average, microseconds per step, share"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms over {steps:.0f} steps = {tot / steps / 1e3:.1f} us per step")
for r in rows[:48]:
    t = float(r["TotalDurationNs"])
    print(f"{r['Name'][:78]:78s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs']) / 1e3:8.1f} per_step_us {t / steps / 1e3:8.1f} {100 * t / tot:5.1f}%")
