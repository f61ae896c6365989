"""python3 scratch/stats_diff.py <kernel_stats_A.csv> <kernel_stats_B.csv>: calls / total ms of the kernels whose total differs by > 0.3 ms"""
import csv, sys
def load(f): return {r["Name"][:70]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load(sys.argv[1]), load(sys.argv[2])
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print(f"total kernel ms: A {ta:.1f}  B {tb:.1f}")
for k in sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1])):
    ca, ma = a.get(k, (0, 0.0)); cb, mb = b.get(k, (0, 0.0))
    if abs(ma - mb) > 0.3: print(f"{k:70s} A {ca:5d} calls {ma:8.2f} ms | B {cb:5d} calls {mb:8.2f} ms | B-A {mb - ma:+.2f}")
