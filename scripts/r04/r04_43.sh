#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for B in 4096 4112 8192 12288 16384; do
  echo -n "nS 4 batch $B product: "
  NS=4 BATCH=$B MAPPING=auto timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "^f " | tail -1
  echo -n "nS 4 batch $B 8-wave workgroups, no LDS pad beyond 256 workgroups: "
  OCS_P2_NO_LDS_PAD=1 OCS_LIB_OVERRIDE=$ROOT/optimal-control-solvers_amd/lib/libocs_p2pad.so NS=4 BATCH=$B MAPPING=auto timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "^f " | tail -1
done | tee $OUT/p2_pair_on_cu.log
