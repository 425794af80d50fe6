#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_user_problems.py tests/test_gpu_rk4_parity.py -m gpu -q 2>&1 | tail -4
for B in 1000 1024; do
  echo -n "vector plugin nS 2 batch $B auto: "; BATCH=$B MAPPING=auto timeout -k 10 200 python scripts/user_pair_time.py 2>&1 | tail -1
  echo -n "vector plugin fb_sweep batch $B: "; BATCH=$B timeout -k 10 200 python scripts/fbs_vector_time.py 2>&1 | tail -1
done | tee gpurun_out/vector_ragged.log
