#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fb_sweep.py -m gpu -q -x -k "interpolant or pchip" 2>&1 | tail -3 || exit 1
timeout -k 10 300 python scripts/api_survey_time.py 2>&1 | grep "vectorInterpolant" | tee gpurun_out/api_survey_interp_after.log
