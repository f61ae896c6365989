#!/bin/bash
# usage: tools/prof_gradicp.sh <tag>  -- per-kernel time of bench.py --odom gradicp (rocprofv3 --kernel-trace --stats)
export TMPDIR=/tmp
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/gradicp_$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/gi_$TAG -o g --output-format csv -- python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --odom gradicp > $OUT/bench.json 2> $OUT/rocprof.err
f=$(find /tmp/gi_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $OUT/kernel_stats.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --odom gradicp ; total kernel time {tot/1e6:.1f} ms")
for r in rows[:45]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
grep -E "knn|icp|grid|transform|pf_|vertex|active|gather_active|scan|cp_count" $OUT/kernel_stats.txt | cut -c1-170
