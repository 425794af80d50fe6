#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04r
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_fb_sweep.py tests/test_gpu_user_problems.py -m gpu -q -x 2>&1 | tail -3 || exit 1
for rep in 1 2; do
for L in "" upf0; do
  echo "== lib ${L:-product}"
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} timeout -k 10 120 python scripts/fold_time.py 2>&1 | grep "per sweep" | tail -2
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} timeout -k 10 120 python scripts/fbs_time.py 2>&1 | grep solve | tail -1
done; done > $OUT/fold_upf.log 2>&1
cat $OUT/fold_upf.log
