#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for B in 4096 4112 5120 6144 8192 8208 12288; do
  echo -n "nS 4 batch $B ($((B/16)) workgroups) auto: "
  NS=4 BATCH=$B MAPPING=auto timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "per iteration" | tail -3 | tr '\n' ' '; echo
done | tee $OUT/p2_occupancy_steps.log
