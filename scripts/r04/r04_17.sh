#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04v
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_user_problems.py -m gpu -q -x -k "six_state" 2>&1 | tail -15 || exit 1
timeout -k 10 200 bash scripts/big_plugin_prof.sh 2>&1 | tee $OUT/big_plugin_kernels.txt
grep "fb_sweep\|adjoint" gpurun_out/big_plugin_prof.log | tail -2
