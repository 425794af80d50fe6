#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_controls_shooting.py tests/test_gpu_lq.py tests/test_gpu_user_problems.py tests/test_golden.py tests/test_gpu_multi_device.py -m gpu -q -x 2>&1 | tail -4 || exit 1
timeout -k 10 600 python scripts/api_survey2_time.py 2>&1 | grep "RK4Infinite" | tee gpurun_out/api_survey2_after.log
