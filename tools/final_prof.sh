#!/bin/bash
# usage: tools/final_prof.sh <tag>  -- the evidence set of a round for bench.py's DEFAULT command: the JSON line, the rocprofv3 --kernel-trace
# --stats summary + timeline of the same command, and the PMC traffic passes (tools/bench_pmc.sh).  Writes under gpurun_out/final_<tag>/.
export TMPDIR=/tmp
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench done" > $OUT/progress.txt
rocprofv3 --kernel-trace --stats -d /tmp/fin_$TAG -o full --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
echo "rocprof done" >> $OUT/progress.txt
f=$(find /tmp/fin_$TAG -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv
t=$(find /tmp/fin_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py "$t" 0.25 > $OUT/timeline.txt 2>&1
python3 - "$f" > $OUT/kernel_stats.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ; total kernel time {tot/1e6:.1f} ms")
for r in rows[:60]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
bash tools/bench_pmc.sh $TAG > $OUT/pmc.log 2>&1
cp gpurun_out/benchpmc_$TAG/traffic.txt $OUT/pmc_traffic.txt
tail -1 $OUT/bench.json | cut -c1-400
