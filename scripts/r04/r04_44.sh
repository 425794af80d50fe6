#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
OCS_LANE_XRC_MIN=1 timeout -k 10 900 python -m pytest tests/test_gpu_user_problems.py -m gpu -q -x 2>&1 | tail -3 || exit 1
for X in 0 32768; do
  echo -n "vector plugin nS 2 batch 65536 lane OCS_LANE_XRC_MIN=$X: "
  OCS_LANE_XRC_MIN=$X BATCH=65536 MAPPING=lane timeout -k 10 200 python scripts/user_pair_time.py 2>&1 | tail -1
  echo -n "vector plugin nS 4 batch 65536 lane OCS_LANE_XRC_MIN=$X: "
  OCS_LANE_XRC_MIN=$X NS=4 BATCH=65536 MAPPING=lane timeout -k 10 200 python scripts/user_pair_time.py 2>&1 | tail -1
done | tee $OUT/user_lane_xrc.log
