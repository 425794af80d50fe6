#!/bin/bash
# round 4, third GPU call: the time-parallel LQ passes -- parity tests, then timing by batch and mapping
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04c
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_lq.py tests/test_gpu_multi_device.py -m gpu -x -q > $OUT/pytest_lq.log 2>&1; echo "pytest rc $?"; tail -15 $OUT/pytest_lq.log
{
for B in 1024 2048 4096 8192; do
  for M in 0 3 4; do
    echo "== batch $B mapping $M"; BATCH=$B MAPPING=$M timeout -k 10 200 python scripts/lq_time.py 2>&1 | grep forward | tail -2
  done
done
for W in 512 2048; do
  echo "== batch 1024 mapping 4, OCS_LQ_CHUNK_WAVES=$W"; OCS_LQ_CHUNK_WAVES=$W BATCH=1024 MAPPING=4 timeout -k 10 200 python scripts/lq_time.py 2>&1 | grep forward | tail -2
done
for B in 64 256 512; do
  for M in 0 3; do
    echo "== batch $B mapping $M"; BATCH=$B MAPPING=$M timeout -k 10 200 python scripts/lq_time.py 2>&1 | grep forward | tail -1
  done
done
} > $OUT/lq_time.log 2>&1
cat $OUT/lq_time.log
