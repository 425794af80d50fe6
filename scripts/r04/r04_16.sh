#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04u
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_fb_sweep.py tests/test_gpu_user_problems.py -m gpu -q -x 2>&1 | tail -3 || exit 1
timeout -k 10 200 python scripts/big_plugin_time.py 2>&1 | grep "fb_sweep\|adjoint" | tail -2 | tee $OUT/big_plugin.log
timeout -k 10 200 bash scripts/big_plugin_prof.sh 2>&1 | tee $OUT/big_plugin_kernels.txt
