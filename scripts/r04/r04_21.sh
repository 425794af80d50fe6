#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04z
mkdir -p $OUT
cd $ROOT
( time timeout -k 10 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 ) 2>&1 | tee $OUT/gpu_tests.log || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log; tail -c 600 $OUT/bench_line.json
