#!/bin/bash
# usage: tools/rehearse_two_ranks.sh <tag>  -- the N-rank GPU code path of bench.py on a ONE-GPU box (E2E_REHEARSE_ONE_GPU=1: both ranks on cuda:0,
# gloo as transport, the gradient bucket staged through the host -- a rehearsal of the launch / graph / exchange / gather logic, never a measurement):
#   a) two ranks, two different sequences                         -> replicas_identical must be true
#   b) two ranks, the SAME sequence (--same-sequence)             -> the averaged update must equal the one-rank update: parameter_checksum(b) == (c)
#   c) one rank, that sequence
OUT=$GRAFT_REPO_ROOT/gpurun_out/rehearsal_$1; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ARGS="--steps 12 --warmup 3 --no-cpu-baseline --no-roofline"
E2E_REHEARSE_ONE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 $ARGS > $OUT/a_two_ranks_two_sequences.json 2> $OUT/a.err; echo "a exit $?" >> $OUT/progress.txt
E2E_REHEARSE_ONE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --same-sequence $ARGS > $OUT/b_two_ranks_same_sequence.json 2> $OUT/b.err; echo "b exit $?" >> $OUT/progress.txt
timeout -k 10 400 python3 bench.py --gpus 1 $ARGS > $OUT/c_one_rank.json 2> $OUT/c.err; echo "c exit $?" >> $OUT/progress.txt
python3 - $OUT <<'PY'
import json, sys
d = sys.argv[1]
def load(p):
    return next(json.loads(l) for l in open(p) if l.startswith("{"))      # gloo prints its connection banner on stdout
a, b, c = (load(f"{d}/{n}") for n in ("a_two_ranks_two_sequences.json", "b_two_ranks_same_sequence.json", "c_one_rank.json"))
print("a) replicas_identical", a["config"]["replicas_identical"], "map points per rank", a["config"]["map_points_per_rank"])
print("b) replicas_identical", b["config"]["replicas_identical"], "checksum", repr(b["config"]["parameter_checksum"]))
print("c) one rank            checksum", repr(c["config"]["parameter_checksum"]))
print("averaged update of two identical gradients == one-rank update:", b["config"]["parameter_checksum"] == c["config"]["parameter_checksum"])
PY
