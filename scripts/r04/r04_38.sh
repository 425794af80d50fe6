#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
OCS_LANE_XRC_MIN=1 timeout -k 10 600 python -m pytest tests/test_gpu_rk4_parity.py -m gpu -q -x 2>&1 | tail -3 || exit 1
for cfg in "4 65536" "4 32768" "4 16384" "2 65536" "2 131072" "1 131072"; do
set -- $cfg
for X in 0 1; do
  echo -n "nS $1 batch $2 lane OCS_LANE_XRC_MIN=$X: "
  OCS_LANE_XRC_MIN=$X NS=$1 BATCH=$2 MAPPING=lane timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "per iteration" | tail -3 | tr '\n' ' '; echo
done; done | tee $OUT/lane_xrc.log
