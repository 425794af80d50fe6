#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
for cfg in "4 65536" "1 131072"; do
set -- $cfg
for L in "" lanent; do
  echo -n "nS $1 batch $2 lane lib ${L:-product}: "
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} NS=$1 BATCH=$2 MAPPING=lane timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "per iteration" | tail -3 | tr '\n' ' '; echo
done; done; done | tee $OUT/lane_nt_stores.log
