#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04q
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
for L in "" spf2 srec srec2; do
  echo "== lib ${L:-product}"
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} timeout -k 10 120 python scripts/fold_time.py 2>&1 | grep "per sweep" | tail -2
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} timeout -k 10 120 python scripts/fbs_time.py 2>&1 | grep solve | tail -1
done; done > $OUT/fold_variants.log 2>&1
cat $OUT/fold_variants.log
OCS_LIB_OVERRIDE=$ROOT/optimal-control-solvers_amd/lib/libocs_srec2.so timeout -k 10 300 python -m pytest tests/test_gpu_fb_sweep.py -m gpu -q 2>&1 | tail -2
