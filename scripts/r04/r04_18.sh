#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04w
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_user_problems.py -m gpu -q -x -k "six_state" 2>&1 | tail -15 || exit 1
timeout -k 10 200 bash scripts/big_plugin_prof.sh 2>&1 | tee $OUT/big_plugin_kernels.txt
grep "fb_sweep\|adjoint" gpurun_out/big_plugin_prof.log | tail -2
for rep in 1 2; do
for L in "" g1lds; do
  echo "== lib ${L:-product}"
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} timeout -k 10 120 python scripts/fold_time.py 2>&1 | grep "per sweep" | tail -2
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} timeout -k 10 120 python scripts/fbs_time.py 2>&1 | grep solve | tail -1
done; done > $OUT/fold_g1lds.log 2>&1
cat $OUT/fold_g1lds.log
OCS_LIB_OVERRIDE=$ROOT/optimal-control-solvers_amd/lib/libocs_g1lds.so timeout -k 10 300 python -m pytest tests/test_gpu_fb_sweep.py -m gpu -q 2>&1 | tail -2
