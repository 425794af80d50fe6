#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
for cfg in "1 1024" "1 1000" "1 16384" "1 16000" "2 1000" "2 1024" "4 1000" "4 1024"; do
set -- $cfg
  echo -n "fb_sweep nS $1 batch $2: "
  NS=$1 BATCH=$2 timeout -k 10 200 python scripts/fbs_time.py 2>&1 | grep solve | tail -1
done | tee gpurun_out/fbs_ragged.log
