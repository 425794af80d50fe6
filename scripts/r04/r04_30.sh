#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for NS in 1 2 4; do
for WG in 512 1024 2048 4096; do
B=$((WG * 64 / NS))
  echo "== nS $NS workgroups $WG batch $B mapping auto"
  NS=$NS BATCH=$B MAPPING=auto timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "per iteration" | tail -3
done; done > $OUT/pair_auto_after.log 2>&1
cat $OUT/pair_auto_after.log
timeout -k 10 900 python -m pytest tests/test_gpu_rk4_parity.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -3
