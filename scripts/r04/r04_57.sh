#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_multi_device.py -m gpu -q -x 2>&1 | grep -B30 "AssertionError" | tail -50
SEED=71 timeout -k 10 900 python tests/stress_fold.py 60 2>&1 | grep -v " ok$" | tail -8
