#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_map
mkdir -p $OUT
cd $ROOT
BATCHES=4096,8192,16384,32768,49152,65536,131072 MODES=auto,on,lane timeout -k 10 600 python scripts/bl4_time.py 2>&1 | grep "batch\|vs" > $OUT/bl4_by_batch_mode.log
cat $OUT/bl4_by_batch_mode.log
echo "--- fb_sweep by batch (BL-3 family)"
for B in 16384 32768 65536 131072; do BATCH=$B timeout -k 10 200 python scripts/fbs_time.py 2>&1 | grep solve | tail -1; done | tee $OUT/fbs_by_batch.log
