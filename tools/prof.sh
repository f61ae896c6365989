#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]   -- kernel-trace stats + PMC passes for bench.py
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc1 -- python3 bench.py --no-cpu-baseline --no-graph --steps 20 --warmup 5 "$@" > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 bench.py --no-cpu-baseline --no-graph --steps 20 --warmup 5 "$@" > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py --no-cpu-baseline --no-graph --steps 20 --warmup 5 "$@" > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 bench.py --no-cpu-baseline --no-graph --steps 20 --warmup 5 "$@" > $OUT/bench_pmc4.log 2>&1
find $OUT -name "*.csv" | head -50
