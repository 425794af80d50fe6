#!/bin/bash
# round 4, first GPU call: sweep tests after the depth change, two bench runs (reproducibility of every leg), the sweep
# loop at depth 1 / 2 / 3, the pass pair over rotating buffer sets with kernel stats of both modes
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_fb_sweep.py -m gpu -x -q > $OUT/pytest_fbs.log 2>&1 || { tail -30 $OUT/pytest_fbs.log; exit 1; }
tail -3 $OUT/pytest_fbs.log
timeout -k 10 300 python bench.py > $OUT/bench1.json 2> $OUT/bench1.err || { tail -20 $OUT/bench1.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench2.json 2> $OUT/bench2.err || { tail -20 $OUT/bench2.err; exit 1; }
for d in 1 2 3; do echo "depth $d"; OCS_FBS_DEPTH=$d timeout -k 10 120 python scripts/fbs_time.py; done > $OUT/fbs_depth.log 2>&1
cat $OUT/fbs_depth.log
for r in 1 3; do ROTATE=$r timeout -k 10 120 python scripts/pair_rotate.py; done > $OUT/rotate.log 2>&1
for w in f b; do for r in 1 3; do WHAT=$w ROTATE=$r timeout -k 10 120 python scripts/pair_rotate.py | tail -1; done; done >> $OUT/rotate.log 2>&1
cat $OUT/rotate.log
cd /tmp && export TMPDIR=/tmp
for r in 1 3; do
  ROTATE=$r rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_rot$r -- python3 $ROOT/scripts/pair_rotate.py > $OUT/trace_rot$r.log 2>&1 || echo "trace $r failed"
  f=$(ls -t $OUT/trace_rot$r/*/*kernel_stats.csv | head -1); head -5 $f | cut -d, -f1-5
done
