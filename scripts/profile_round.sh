#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats                     -> per-kernel average durations
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes)
#   3. the same two counters on a calibration copy kernel with this path's access width (8 B/lane)
# Results land in gpurun_out/prof_$TAG/; scripts/summarize_profile.py turns them into profiles/.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-fb-sweep --no-live-traffic --steps 10 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 1
CAL="python3 $ROOT/scripts/calibrate_traffic.py"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -- $CAL > $OUT/cal_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_write -- $CAL > $OUT/cal_write.log 2>&1 || exit 1
echo "profile $TAG done"
