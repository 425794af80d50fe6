#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_stress
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_controls_shooting.py -m gpu -q -x -k "randomised" 2>&1 | tail -30 || exit 1
SEED=12 timeout -k 10 900 python tests/stress_nlp.py 400 > $OUT/stress_nlp.log 2>&1; echo "stress_nlp rc=$?"; tail -2 $OUT/stress_nlp.log; grep -c refused $OUT/stress_nlp.log
SEED=13 timeout -k 10 900 python tests/stress_rk4.py 1500 > $OUT/stress_rk4_1500.log 2>&1; echo "stress_rk4 rc=$?"; tail -1 $OUT/stress_rk4_1500.log
