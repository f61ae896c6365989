#!/bin/bash
# usage: tools/evidence.sh <tag>  -- the measured evidence of a round, under gpurun_out/evidence_<tag>/ (copy what is to be judged to profiles/):
#   bench_default.json          python3 bench.py --gpus 1 --steps 20 --warmup 5                      (the driver's command)
#   bench_fullpass.json         ... --steps 177 --warmup 6: one whole pass of the 60-frame sequence (59 keyframes x 3 steps)
#   bench_tum.json              ... --tum: BASELINE configs[3] (TUM intrinsics, 10 % depth holes, threshold 0.12)
#   bench_gradicp.json          ... --odom gradicp: frame-to-model GradICP odometry in the map step, reports the ATE
#   bench_overlap.json          E2E_WGRAD_OVERLAP=1: backward-weight chains on a second stream (A/B of the one-stream default)
#   kernel_stats.txt, timeline.txt   rocprofv3 --kernel-trace --stats of `bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline`
#   pmc_traffic.{txt,json}      tools/bench_pmc.sh (FETCH_SIZE / WRITE_SIZE passes, stamped with the source hash)
#   gemm_tune.txt               tools/gemm_tune.py both
#   pass_profile.txt            tools/pass_profile.py: wall time per keyframe over a whole pass + the map-size dependent entry points
#   conv_stamps.txt             scratch/conv_stamps.py on the -DE2E_CONV_STAMPS build (if scratch/_stamped/ holds one)
#   pytest_gpu.log              python3 -m pytest tests -m gpu -q
# usage: tools/evidence.sh <tag> [bench|suite]   -- two parts, each within one gpurun call (20 minutes): "bench" = PMC traffic, the bench lines and
# the rocprofv3 kernel statistics; "suite" = GEMM table, pass profile, GPU test log.  Default: both.
export TMPDIR=/tmp
TAG=$1
PART=${2:-all}
OUT=$GRAFT_REPO_ROOT/gpurun_out/evidence_$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "$PART" != "suite" ]; then
# PMC traffic FIRST: bench.py reports roofline.traffic from profiles/r04_bench_pmc_traffic.json only if that record carries the stamp of the build it runs
bash tools/bench_pmc.sh $TAG > $OUT/pmc.log 2>&1
cp gpurun_out/benchpmc_$TAG/traffic.txt $OUT/pmc_traffic.txt; cp gpurun_out/benchpmc_$TAG/traffic.json $OUT/pmc_traffic.json
cp gpurun_out/benchpmc_$TAG/traffic.json profiles/r04_bench_pmc_traffic.json
echo "pmc done" > $OUT/progress.txt
# whole pass BEFORE the default line: the default line embeds it (value_whole_pass) with the stamp of the build it was measured on
python3 bench.py --gpus 1 --steps 177 --warmup 6 --no-cpu-baseline > $OUT/bench_fullpass.json 2> $OUT/bench_fullpass.err; echo "fullpass $?" >> $OUT/progress.txt
cp $OUT/bench_fullpass.json profiles/r04_bench_seq_fullpass.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default $?" >> $OUT/progress.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --tum --no-cpu-baseline --no-roofline > $OUT/bench_tum.json 2> $OUT/bench_tum.err; echo "tum $?" >> $OUT/progress.txt
E2E_WGRAD_OVERLAP=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench_overlap.json 2> $OUT/bench_overlap.err; echo "overlap $?" >> $OUT/progress.txt
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-roofline --odom gradicp > $OUT/bench_gradicp.json 2> $OUT/bench_gradicp.err; echo "gradicp $?" >> $OUT/progress.txt
rocprofv3 --kernel-trace --stats -d /tmp/ev_$TAG -o full --output-format csv -- python3 bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
echo "rocprof $?" >> $OUT/progress.txt
f=$(find /tmp/ev_$TAG -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv
t=$(find /tmp/ev_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py "$t" 0.45 k_pf_append 40 > $OUT/timeline.txt 2>&1
python3 - "$f" > $OUT/kernel_stats.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline --no-roofline ; total kernel time {tot/1e6:.1f} ms")
for r in rows[:70]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
fi
if [ "$PART" != "bench" ]; then
# are the ATen fill launches of the kernel trace per step or start-up?  the same trace over 9 and over 65 steps: equal counts = allocation-time zero fills
for n in 6 60; do
  rocprofv3 --kernel-trace --stats -d /tmp/evf_${TAG}_$n -o t --output-format csv -- python3 bench.py --gpus 1 --steps $n --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2> $OUT/rocprof_fill_$n.err
done
python3 - /tmp/evf_${TAG}_6 /tmp/evf_${TAG}_60 > $OUT/fill_launches.txt <<'PY'
import csv, glob, sys
for d, steps in ((sys.argv[1], 9), (sys.argv[2], 63)):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "FillFunctor" in r["Name"]]
    print(f"bench.py --steps {steps - 3} --warmup 3 ({steps} refinement steps): {sum(int(r['Calls']) for r in rows)} FillFunctor launches, {sum(float(r['TotalDurationNs']) for r in rows) / 1e3:.0f} us in total")
    for r in rows:
        print(f"    {r['Calls']:>5s} x {r['Name'][:110]}")
PY
echo "fills done" >> $OUT/progress.txt
timeout -k 10 500 python3 tools/gemm_tune.py both > $OUT/gemm_tune.txt 2>&1; echo "tune $?" >> $OUT/progress.txt
timeout -k 10 300 python3 tools/pass_profile.py > $OUT/pass_profile.txt 2> $OUT/pass_profile.err; echo "pass profile $?" >> $OUT/progress.txt
if [ -f scratch/_stamped/libe2eslam_hip_stamped.so ]; then python3 scratch/conv_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/conv_stamps.txt; echo "stamps $?" >> $OUT/progress.txt; fi
timeout -k 10 600 python3 -m pytest tests -m gpu -q --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest $?" >> $OUT/progress.txt
fi
cat $OUT/progress.txt; head -c 300 $OUT/bench_default.json; echo; head -c 300 $OUT/bench_fullpass.json; echo; head -c 300 $OUT/bench_gradicp.json
