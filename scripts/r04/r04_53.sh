#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_rk4_parity.py tests/test_golden.py tests/test_gpu_fb_sweep.py tests/test_gpu_controls_shooting.py -m gpu -q -x 2>&1 | tail -4 || exit 1
SEED=51 timeout -k 10 600 python tests/stress_rk4.py 1500 2>&1 | tail -1
SEED=52 timeout -k 10 600 python tests/stress_nlp.py 400 2>&1 | tail -1
SEED=53 timeout -k 10 600 python tests/stress_fold.py 30 2>&1 | tail -1
for cfg in "4 1000 4100" "4 1000 100" "4 1000 1000" "2 1000 1000" "1 1000 1000" "4 1000 4096"; do
set -- $cfg
  echo -n "nS $1 N $2 batch $3 auto: "
  NS=$1 NSTEPS=$2 BATCH=$3 MAPPING=auto timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "per iteration" | tail -3 | tr '\n' ' '; echo
done | tee gpurun_out/pair_ragged_after.log
