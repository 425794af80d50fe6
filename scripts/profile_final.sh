#!/bin/bash
# rocprofv3 kernel stats of the default bench.py run (the command the bench line comes from); through gpurun, repo root
TAG=${1:-r03f}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench.err || echo "failed"
f=$(ls -t $OUT/trace/*/*kernel_stats.csv | head -1); head -14 $f | cut -d, -f1-5
