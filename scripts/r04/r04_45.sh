#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fb_sweep.py -m gpu -q -x -k "beyond_512" 2>&1 | tail -25
