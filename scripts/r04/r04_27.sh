#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_user_problems.py -m gpu -q -x -k "reference_symbolic" 2>&1 | tail -30
