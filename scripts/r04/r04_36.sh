#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for cfg in "4 4096" "4 8192" "4 16384" "4 24576" "4 32768" "1 16384" "1 32768" "1 49152" "2 24576"; do
set -- $cfg
for M in lane scan; do
  echo -n "vector plugin nS $1 batch $2 mapping $M: "
  NS=$1 BATCH=$2 MAPPING=$M timeout -k 10 200 python scripts/user_pair_time.py 2>&1 | tail -1
done; done | tee $OUT/user_vector_pair_ns14.log
