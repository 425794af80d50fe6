#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_controls_shooting.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -3 || exit 1
SEED=31 timeout -k 10 600 python tests/stress_nlp.py 600 2>&1 | tail -1
timeout -k 10 300 python scripts/control_kernels_time.py 2>&1 | grep -v amdgpu.ids | grep Chebyshev | tee gpurun_out/control_kernels_time_after.log
timeout -k 10 300 python scripts/api_survey_time.py 2>&1 | grep "nlp_objective" | tee gpurun_out/api_survey_after.log
