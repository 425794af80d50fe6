#!/bin/bash
# BL-2 problem by batch and mapping (pass pair steady-state loops, scripts/pair_loop.py)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for B in 8192 16384 32768 65536; do
for M in auto lane pipeline scan; do
  echo "== batch $B mapping $M"
  BATCH=$B MAPPING=$M timeout -k 10 200 python scripts/pair_loop.py 2>&1 | grep "per iteration" | tail -3
done; done > $OUT/pair_by_batch_mapping.log 2>&1
cat $OUT/pair_by_batch_mapping.log
