#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for B in 8192 16384 24576 32768 65536; do
for V in 1 1000000; do
  echo -n "vector plugin fb_sweep batch $B OCS_COSTATE_VSCAN_MAX=$V: "
  OCS_COSTATE_VSCAN_MAX=$V BATCH=$B timeout -k 10 200 python scripts/fbs_vector_time.py 2>&1 | tail -1
done; done | tee $OUT/fbs_vector_by_batch.log
