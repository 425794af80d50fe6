#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
for L in "" uprio1 uprio3; do
  echo "== lib ${L:-product}"
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} timeout -k 10 120 python scripts/fold_time.py 2>&1 | grep "per sweep" | tail -1
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} timeout -k 10 120 python scripts/fbs_time.py 2>&1 | grep solve | tail -1
done; done > $OUT/fold_uprio.log 2>&1
cat $OUT/fold_uprio.log
