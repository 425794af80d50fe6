#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_fb_sweep.py tests/test_gpu_user_problems.py tests/test_golden.py tests/test_gpu_multi_device.py -m gpu -q 2>&1 | tail -8
timeout -k 10 400 python scripts/fbs_offnode_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fbs_offnode_after.log
