#!/bin/bash
# pass pair by workgroup count and mapping for nS = 1, 2 (scripts/pair_loop.py)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for NS in 1 2; do
for WG in 512 1024 2048 4096; do
B=$((WG * 64 / NS))
for M in lane pipeline scan; do
  echo "== nS $NS workgroups $WG batch $B mapping $M"
  NS=$NS BATCH=$B MAPPING=$M timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "per iteration" | tail -3
done; done; done > $OUT/pair_by_wg_mapping_ns12.log 2>&1
cat $OUT/pair_by_wg_mapping_ns12.log
