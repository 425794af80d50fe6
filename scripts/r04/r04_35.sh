#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for B in 4096 16384 32768 65536 131072; do
for M in auto lane pipeline scan; do
  echo -n "vector plugin nS 2 batch $B mapping $M: "
  BATCH=$B MAPPING=$M timeout -k 10 200 python scripts/user_pair_time.py 2>&1 | tail -1
done; done | tee $OUT/user_vector_pair_by_batch_mapping.log
