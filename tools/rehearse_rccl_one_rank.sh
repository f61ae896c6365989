#!/bin/bash
# usage: tools/rehearse_rccl_one_rank.sh <tag>   -- on a ONE-GPU box: the N-rank refinement step over the REAL RCCL transport with a process
# group of size 1 (E2E_FORCE_EXCHANGE=1: backward as two captured graphs around the two-segment all-reduce of the gradient bucket -- async
# all-reduce of the bucket's tail on RCCL's stream, synchronous one of its head --, Adam as a third graph).  The sum over one rank is the
# rank's own gradient, so the run must end with EXACTLY the parameter checksum and map size of the plain run over the same steps.
# Writes gpurun_out/rccl_one_rank_<tag>/{plain,forced}.json and verdict.json.  A correctness rehearsal, never a measurement.
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/rccl_one_rank_$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ARGS="--gpus 1 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline"
python3 bench.py $ARGS > $OUT/plain.json 2> $OUT/plain.err; echo "plain $?" > $OUT/progress.txt
E2E_FORCE_EXCHANGE=1 python3 bench.py $ARGS > $OUT/forced.json 2> $OUT/forced.err; echo "forced $?" >> $OUT/progress.txt
python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
def line(f):
    for l in open(f):
        if l.startswith("{"):
            return json.loads(l)
a, b = line(out + "/plain.json"), line(out + "/forced.json")
v = {"what": "tools/rehearse_rccl_one_rank.sh: one rank, plain step vs the N-rank step forced over RCCL (process group of size 1)",
     "plain": {"steps_per_s": a["value"], "parameter_checksum": a["config"]["parameter_checksum"], "map_points": a["config"]["map_points_rank0"]},
     "forced": {"steps_per_s": b["value"], "parameter_checksum": b["config"]["parameter_checksum"], "map_points": b["config"]["map_points_rank0"],
                "exchange_forced_on_one_rank": b["config"].get("exchange_forced_on_one_rank")},
     "identical": a["config"]["parameter_checksum"] == b["config"]["parameter_checksum"] and a["config"]["map_points_rank0"] == b["config"]["map_points_rank0"]}
json.dump(v, open(out + "/verdict.json", "w"), indent=1)
print(json.dumps(v))
PY
