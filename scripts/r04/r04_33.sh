#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for B in 16384 32768 65536 131072 262144; do
for L in 512 100000; do
  echo -n "batch $B OCS_FOLD_MAX_WG=$L: "
  OCS_FOLD_MAX_WG=$L BATCH=$B timeout -k 10 200 python scripts/fbs_time.py 2>&1 | grep solve | tail -1
done; done | tee $OUT/fbs_by_batch_fold_limit.log
