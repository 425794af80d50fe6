#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04n
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_gpu.log
for B in 512 1024 2048 4096 8192; do echo "== batch $B"; BATCH=$B timeout -k 10 200 python scripts/lq_time.py 2>&1 | grep forward | tail -2; done > $OUT/lq_time.log 2>&1; cat $OUT/lq_time.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err || tail -20 $OUT/bench.err
