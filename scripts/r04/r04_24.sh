#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_stress
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_rk4_parity.py -m gpu -q -x -k "randomised" 2>&1 | tail -40 || exit 1
SEED=11 timeout -k 10 900 python tests/stress_rk4.py 400 > $OUT/stress_rk4.log 2>&1; echo "stress_rk4 rc=$?"; tail -2 $OUT/stress_rk4.log
SEED=7 timeout -k 10 900 python tests/stress_fold.py 40 > $OUT/stress_fold.log 2>&1; echo "stress_fold rc=$?"; tail -2 $OUT/stress_fold.log
