#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tail -6
SEED=71 timeout -k 10 900 python tests/stress_fold.py 60 2>&1 | tail -1
