#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
for cfg in "4 96 4096" "4 104 4096" "4 10000 4096" "4 10003 4096" "1 10000 16384" "4 1000 4100" "4 1000 100" "2 1000 4128"; do
set -- $cfg
  echo -n "nS $1 N $2 batch $3 auto: "
  NS=$1 NSTEPS=$2 BATCH=$3 MAPPING=auto timeout -k 10 300 python scripts/pair_loop.py 2>&1 | grep "per iteration" | tail -3 | tr '\n' ' '; echo
done | tee gpurun_out/pair_odd_shapes.log
