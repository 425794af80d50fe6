#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for NS in 2 4; do
for WG in 512 1024 2048 4096; do
B=$((WG * 64 / NS))
for L in 512 100000; do
  echo -n "nS $NS batch $B ($WG workgroups) OCS_FOLD_MAX_WG=$L: "
  NS=$NS OCS_FOLD_MAX_WG=$L BATCH=$B timeout -k 10 200 python scripts/fbs_time.py 2>&1 | grep solve | tail -1
done; done; done | tee $OUT/fbs_by_batch_fold_limit_ns24.log
