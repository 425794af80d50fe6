#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04y
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_user_problems.py tests/test_gpu_fb_sweep.py -m gpu -q -x 2>&1 | tail -5 || exit 1
timeout -k 10 200 python scripts/fbs_user_time.py 2>&1 | grep "per sweep" | tee $OUT/fbs_user_time.log
timeout -k 10 200 python scripts/big_plugin_time.py 2>&1 | grep "fb_sweep\|adjoint" | tail -2 | tee $OUT/big_plugin.log
