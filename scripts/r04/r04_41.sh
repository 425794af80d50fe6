#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
OCS_LANE_XRC_MIN=1 timeout -k 10 600 python -m pytest tests/test_gpu_rk4_parity.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -3 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_rk4_parity.py tests/test_gpu_fb_sweep.py -m gpu -q -x 2>&1 | tail -3 || exit 1
for cfg in "4 65536" "4 32768" "2 65536"; do
set -- $cfg
  echo -n "nS $1 batch $2 auto: "
  NS=$1 BATCH=$2 MAPPING=auto timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "per iteration" | tail -3 | tr '\n' ' '; echo
done
