#!/bin/bash
# usage: scratch/build_variant.sh <name> <extra conv.hip flags...>  ->  scratch/_variants/<name>/libe2eslam_hip.so (conv.hip rebuilt with the flags, other objects from csrc/_obj)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); P=$ROOT/end-to-end-self-supervised-slam_amd
name=$1; shift
mkdir -p $ROOT/scratch/_variants/$name
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-result -ffp-contract=fast "$@" -I$ROOT/include -I$P/csrc -x hip -c $P/csrc/conv.hip -o $ROOT/scratch/_variants/$name/conv.o
objs=$(ls $P/csrc/_obj/*.o | grep -v "/conv.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scratch/_variants/$name/libe2eslam_hip.so $objs $ROOT/scratch/_variants/$name/conv.o
echo built $name
