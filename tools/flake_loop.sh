#!/bin/bash
# usage: tools/flake_loop.sh <tag> <n> [pytest args]   -- repeat the GPU test files that have shown order-dependent failures, in suite order
OUT=gpurun_out/flake_$1; mkdir -p $OUT; N=$2; shift; shift
for i in $(seq 1 $N); do
  timeout -k 10 500 python -m pytest tests/test_gpu_driver.py tests/test_gpu_netplan.py tests/test_gpu_network.py tests/test_gpu_train_depth.py -q "$@" > $OUT/run_$i.log 2>&1
  echo "run $i exit $? : $(tail -1 $OUT/run_$i.log)" >> $OUT/progress.txt
  grep "^FAILED" $OUT/run_$i.log >> $OUT/progress.txt
done
cat $OUT/progress.txt
