#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_user_problems.py tests/test_abi_exports.py -q -x -k "control_char or false_decl or abi or export or header" 2>&1 | tail -15
