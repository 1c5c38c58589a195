#!/bin/bash
# a tagged diagnostic library from the product objects + a few units recompiled with extra flags:
#   tools/ab/build_variant.sh TAG "-DPCR_X=1 ..." unit.hip [unit.hip ...]     (never ships: lib/libpcr_hip_TAG.so is git-ignored,
#   and tests/test_abi.py refuses a tree that carries one at release time only by convention -- remove it after the A/B)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=$1; FLAGS=$2; shift 2
LIB=$ROOT/point-cloud-reid_amd/pcr_amd/lib
mkdir -p $LIB/obj_$TAG
for u in "$@"; do
  extra=""
  case $u in point_ops.hip|edge_kernels.hip) extra="-ffp-contract=off";; esac
  hipcc --offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -std=c++17 -I$ROOT/include -I$ROOT/point-cloud-reid_amd/csrc $FLAGS $extra \
    -c $ROOT/point-cloud-reid_amd/csrc/$u -o $LIB/obj_$TAG/$u.o &
done
wait
objs=""
for o in $LIB/obj/*.o; do
  b=$(basename $o)
  if [ -f $LIB/obj_$TAG/$b ]; then objs="$objs $LIB/obj_$TAG/$b"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB/libpcr_hip_$TAG.so $objs
ls -la $LIB/libpcr_hip_$TAG.so
