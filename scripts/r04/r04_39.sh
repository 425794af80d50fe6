#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_stress
mkdir -p $OUT
cd $ROOT
OCS_LANE_XRC_MIN=1 SEED=21 timeout -k 10 900 python tests/stress_rk4.py 1500 > $OUT/stress_rk4_xrc_forced.log 2>&1; echo "stress_rk4 (lane re-integration forced at every batch) rc=$?"; tail -1 $OUT/stress_rk4_xrc_forced.log
OCS_LANE_XRC_MIN=1 timeout -k 10 900 python -m pytest tests/test_gpu_rk4_parity.py tests/test_golden.py tests/test_gpu_controls_shooting.py tests/test_gpu_user_problems.py -m gpu -q -x 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests/test_gpu_rk4_parity.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -3
for cfg in "4 65536" "4 32768" "2 65536" "2 131072"; do
set -- $cfg
  echo -n "nS $1 batch $2 auto: "
  NS=$1 BATCH=$2 MAPPING=auto timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "per iteration" | tail -3 | tr '\n' ' '; echo
done | tee $ROOT/gpurun_out/r04_map/auto_after_xrc.log
