#!/bin/bash
# per-layer kernel breakdown: rocprofv3 --kernel-trace --stats on tools/conv_prof_one.py for the heavy layers
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/convprof; mkdir -p $OUT
for L in "$@"; do
  rocprofv3 --kernel-trace --stats -d /tmp/cp_$L -o $L --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/conv_prof_one.py $L > /dev/null 2>&1
  f=$(find /tmp/cp_$L -name "*kernel_stats.csv" | head -1)
  echo "== $L" >> $OUT/summary.txt
  python3 - "$f" >> $OUT/summary.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
done
cat $OUT/summary.txt
