#!/bin/bash
# usage: tools/bench_pmc_sq.sh <tag>  -- SQ / GRBM counters of every kernel of bench.py's default workload (two rocprofv3 --pmc passes of 8 SQ
# counters each; no tracing domains combined with --pmc): MFMA pipe busy share, LDS bank conflicts, wait shares, waves per launch.
export TMPDIR=/tmp
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/benchsq_$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --no-roofline --steps 9 --warmup 3"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/sq1_$TAG -o pmc -- python3 bench.py $ARGS > $OUT/p1.log 2>&1
echo "pass 1 done" > $OUT/progress.txt
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/sq2_$TAG -o pmc -- python3 bench.py $ARGS > $OUT/p2.log 2>&1
echo "pass 2 done" >> $OUT/progress.txt
python3 - /tmp/sq1_$TAG /tmp/sq2_$TAG > $OUT/sq_summary.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
def m(k, c):
    v = acc[k].get(c)
    return sum(v) / len(v) if v else float("nan")
rows = []
for k in acc:
    n = len(acc[k].get("SQ_WAVES", []))
    rows.append((m(k, "GRBM_GUI_ACTIVE") * n, k, n))
rows.sort(reverse=True)
print("per-launch means; SQ_*_CYCLES / WAIT / ACTIVE count quad-cycles summed over waves or SIMDs as rocprofv3 reports them (gfx950)")
print(f"{'kernel':64s} {'launches':>8s} {'gui_active':>11s} {'waves':>8s} {'mfma_busy/sq_busy':>18s} {'insts_mfma':>11s} {'insts_valu':>11s} {'insts_lds':>10s} "
      f"{'lds_conflict/lds_active':>24s} {'wait_any/wave_cyc':>18s} {'wait_inst/wave_cyc':>19s} {'active_any/wave_cyc':>20s}")
for _, k, n in rows[:34]:
    wc = m(k, "SQ_WAVE_CYCLES")
    print(f"{k:64s} {n:8d} {m(k, 'GRBM_GUI_ACTIVE'):11.0f} {m(k, 'SQ_WAVES'):8.0f} {m(k, 'SQ_VALU_MFMA_BUSY_CYCLES') / max(m(k, 'SQ_BUSY_CYCLES'), 1):18.3f} "
          f"{m(k, 'SQ_INSTS_MFMA'):11.0f} {m(k, 'SQ_INSTS_VALU'):11.0f} {m(k, 'SQ_INSTS_LDS'):10.0f} "
          f"{m(k, 'SQ_LDS_BANK_CONFLICT') / max(m(k, 'SQ_LDS_IDX_ACTIVE'), 1):24.3f} {m(k, 'SQ_WAIT_ANY') / max(wc, 1):18.3f} "
          f"{m(k, 'SQ_WAIT_INST_ANY') / max(wc, 1):19.3f} {m(k, 'SQ_ACTIVE_INST_ANY') / max(wc, 1):20.3f}")
PY
cat $OUT/sq_summary.txt | cut -c1-260
