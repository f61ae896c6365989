#!/bin/bash
# usage: tools/bench_pmc.sh <tag>   -- HBM-side traffic of every kernel of bench.py's default workload: two separate rocprofv3 --pmc
# passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950), per-kernel mean per launch.  FETCH_SIZE / WRITE_SIZE are KiB;
# FETCH_SIZE is doubled for the kernels whose loads are 16 B per lane (gfx950 tallies their 128-B requests at 64 B:
# MI355X_MICROARCH.md, HBM section).  The counters sit at the L2's fabric side: Infinity-Cache hits are included.
export TMPDIR=/tmp
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/benchpmc_$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
STEPS=9; WARM=3
ARGS="--no-cpu-baseline --no-roofline --steps $STEPS --warmup $WARM"
STAMP=$(python3 bench.py --print-stamp)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/bp_f_$TAG -o pmc -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1
echo "fetch pass done" > $OUT/progress.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/bp_w_$TAG -o pmc -- python3 bench.py $ARGS > $OUT/write.log 2>&1
echo "write pass done" >> $OUT/progress.txt
python3 - /tmp/bp_f_$TAG /tmp/bp_w_$TAG $((STEPS + WARM)) $STAMP > $OUT/traffic.txt <<'PY'
import csv, glob, sys, collections, json
def load(d, counter):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc
F, W = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
WIDE = ("k_conv_gemm", "k_conv3x3_thin", "k_wgrad3x3_thin", "k_conv7x7", "k_wgrad7x7", "k_wgrad_gemm", "k_wgrad_reduce", "k_adam", "k_weight_layouts", "k_gather_adjoint", "k_act_bwd", "k_maxpool", "k_head", "k_conv_splitk")
rows = []
for k in sorted(set(F) | set(W)):
    f = sum(F.get(k, [0])) / max(len(F.get(k, [])), 1) * 1024
    w = sum(W.get(k, [0])) / max(len(W.get(k, [])), 1) * 1024
    wide = any(s in k for s in WIDE)
    rows.append((k, len(F.get(k, [])), f, f * (2 if wide else 1), w))
rows.sort(key=lambda r: -(r[3] + r[4]) * r[1])
print(f"{'kernel':70s} {'launches':>8s} {'FETCH raw MB':>13s} {'FETCH corr MB':>14s} {'WRITE MB':>9s}   (per launch)")
for k, n, f, fc, w in rows[:40]:
    print(f"{k[:70]:70s} {n:8d} {f / 1e6:13.2f} {fc / 1e6:14.2f} {w / 1e6:9.2f}")
conv = [r for r in rows if "k_conv_gemm" in r[0] or "_thin" in r[0] or "7x7" in r[0] or "k_wgrad_gemm" in r[0] or "k_wgrad_reduce" in r[0] or "k_conv_splitk" in r[0]]
n = sum(r[1] for r in conv)
tot = sum((r[3] + r[4]) * r[1] for r in conv)
print(json.dumps({"refinement_steps": int(sys.argv[3]), "source_stamp": sys.argv[4], "conv_gemm_family_kernel_launches": n, "conv_gemm_family_bytes_total": tot, "conv_gemm_family_bytes_per_kernel_launch": tot / max(n, 1),
                  "source": "tools/bench_pmc.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --no-cpu-baseline --no-roofline; FETCH_SIZE x2 for the 16-B-per-lane loaders (gfx950 correction), KiB -> bytes; L2 fabric-side requests, Infinity-Cache hits included"}))
PY
tail -1 $OUT/traffic.txt > $OUT/traffic.json
cat $OUT/traffic.txt
