#!/bin/bash
# usage: tools/full_prof.sh <tag> [bench args...]  -- rocprofv3 kernel-trace summary + timeline of bench.py's default (sequence) workload
export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/fullprof_$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/fp_$TAG -o full --output-format csv -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
f=$(find /tmp/fp_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
t=$(find /tmp/fp_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py "$t" ${TL_FRAC:-0.3} $TL_DUMP > $OUT/timeline.txt 2>&1
python3 - "$f" > $OUT/summary.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:40]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
cat $OUT/timeline.txt; cat $OUT/summary.txt; tail -1 $OUT/bench.log
