#!/bin/bash
# rocprofv3 kernel stats of the BL-4 kernels: scripts/prof_bl4.sh TAG "8192 65536" [mode]
set -o pipefail
TAG=${1:-bl4}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
MODE=${3:-on}
cd /tmp && export TMPDIR=/tmp
for B in ${2:-8192 65536}; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_$B
  mkdir -p $OUT
  BATCHES=$B MODES=$MODE REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/scripts/bl4_time.py > $OUT/trace.log 2>&1 || exit 1
  f=$(ls -t $OUT/trace/*/*kernel_stats.csv | head -1)
  echo "== batch $B"; head -6 $f | cut -d, -f1-8
done
