#!/bin/bash
# usage: tools/conv_pmc.sh <layer>   -- PMC passes for the GEMM kernels of one conv layer (tools/conv_prof_one.py)
L=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/convpmc_$L; mkdir -p $OUT
P="python3 $GRAFT_REPO_ROOT/tools/conv_prof_one.py $L"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA --output-format csv -d /tmp/cpmc1 -- $P > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/cpmc2 -- $P > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d /tmp/cpmc3 -- $P > $OUT/p3.log 2>&1
python3 - <<'PY' > $OUT/summary.txt
import csv,glob,collections
for d in ("/tmp/cpmc1","/tmp/cpmc2","/tmp/cpmc3"):
    fs=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
    if not fs: print(d,"no csv"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"][:60]
        if "gemm" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k)
        for c,vals in v.items(): print("   ",c,sum(vals)/len(vals),"n",len(vals))
PY
cat $OUT/summary.txt; tail -3 $OUT/p3.log
for d in /tmp/cpmc1 /tmp/cpmc2 /tmp/cpmc3; do f=$(find $d -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/$(basename $d)_trace.csv; done
