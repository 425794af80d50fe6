#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
for NS in 2 3 4; do
  echo "== LogisticK nS $NS, Chebyshev-16"
  NS=$NS BATCHES=2048,4096,8192,16384,32768,65536 MODES=off,lane,auto timeout -k 10 600 python scripts/bl4_time.py 2>&1 | grep "batch" | cut -c1-90
done | tee $OUT/fusion_by_batch_ns234.log
