#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_fb_sweep.py tests/test_gpu_user_problems.py tests/test_golden.py -m gpu -q 2>&1 | tail -12
SEED=61 timeout -k 10 900 python tests/stress_fold.py 90 2>&1 | grep -v "ok$" | tail -8
ORACLE_ALL=1 SEED=62 timeout -k 10 900 python tests/stress_fold.py 12 2>&1 | tail -3
bash scripts/r04/r04_54.sh 2>&1 | tail -8
