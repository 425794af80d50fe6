#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_user_problems.py tests/test_gpu_fb_sweep.py tests/test_gpu_lq.py -m gpu -q 2>&1 | tail -6
