#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for cfg in "4 65536" "2 131072" "2 65536" "4 32768"; do
set -- $cfg
for M in auto lane scan auto lane; do
  echo "== nS $1 batch $2 mapping $M"
  NS=$1 BATCH=$2 MAPPING=$M timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "per iteration" | tail -3 | tr '\n' ' '; echo
done; done > $OUT/pair_ab_same_box.log 2>&1
cat $OUT/pair_ab_same_box.log
