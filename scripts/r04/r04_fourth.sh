#!/bin/bash
# round 4, fourth GPU call: time-parallel LQ passes after the carry fix: parity, timing, kernel trace of the shard
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_lq.py tests/test_gpu_user_problems.py -m gpu -x -q > $OUT/pytest_lq.log 2>&1; echo "pytest rc $?"; tail -8 $OUT/pytest_lq.log
{
for B in 64 512 1024 2048 4096; do
  echo "== batch $B mapping 0"; BATCH=$B timeout -k 10 200 python scripts/lq_time.py 2>&1 | grep forward | tail -2
done
for W in 2048 4096; do
  echo "== batch 1024, OCS_LQ_CHUNK_WAVES=$W"; OCS_LQ_CHUNK_WAVES=$W BATCH=1024 timeout -k 10 200 python scripts/lq_time.py 2>&1 | grep forward | tail -2
  echo "== batch 4096 mapping 4, OCS_LQ_CHUNK_WAVES=$W"; OCS_LQ_CHUNK_WAVES=$W MAPPING=4 BATCH=4096 timeout -k 10 200 python scripts/lq_time.py 2>&1 | grep forward | tail -2
done
} > $OUT/lq_time.log 2>&1
cat $OUT/lq_time.log
cd /tmp && export TMPDIR=/tmp
BATCH=1024 REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1024 -- python3 $ROOT/scripts/lq_time.py > $OUT/trace1024.log 2>&1 || echo "trace failed"
f=$(ls -t $OUT/trace1024/*/*kernel_stats.csv | head -1); head -16 $f | cut -d, -f1-4 | cut -c1-150
