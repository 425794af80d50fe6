#!/bin/bash
# round 4, twelfth GPU call: bench.py (final kernels), evidence of scripts/profile_r04.sh, full -m gpu suite
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04p
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err || tail -20 $OUT/bench.err
timeout -k 10 900 bash scripts/profile_r04.sh r04 > $OUT/profile_r04.log 2>&1; tail -3 $OUT/profile_r04.log
rm -rf $ROOT/gpurun_out/prof_r04/*/runc/*kernel_trace.csv $ROOT/gpurun_out/sq_r04*/*/runc/*kernel_trace.csv
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_gpu.log
