#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_controls_shooting.py -m gpu -q -x -k "tail_leg" 2>&1 | tail -20
