#!/bin/bash
# usage: tools/full_prof.sh <keyframes> <warmup>  -- kernel breakdown of the full refinement step at a given map size
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fullprof_$1; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/fp_$1 -o full --output-format csv -- python3 bench.py --workload full --no-cpu-baseline --steps $1 --warmup $2 > $OUT/bench.log 2>&1
f=$(find /tmp/fp_$1 -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
python3 - "$f" > $OUT/summary.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:28]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
cat $OUT/summary.txt; tail -1 $OUT/bench.log
