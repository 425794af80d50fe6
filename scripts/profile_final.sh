#!/bin/bash
# rocprofv3 kernel stats of bench.py restricted to its headline loop (--no-cpu-baseline --no-fb-sweep: the secondary legs run the
# same kernel templates at other batch sizes and would mix into the averages); through gpurun, repo root
TAG=${1:-r03f}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-fb-sweep --no-live-traffic > $OUT/bench_line.json 2> $OUT/bench.err || echo "failed"
f=$(ls -t $OUT/trace/*/*kernel_stats.csv | head -1); head -14 $f | cut -d, -f1-5
