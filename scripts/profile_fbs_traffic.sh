#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate, as MI355X_MICROARCH.md prescribes) over the fb_sweep solve of BL-3; the
# calibration factors are those of the same round's profile_round.sh run (profiles/<TAG>_traffic.json).
#   bash scripts/profile_fbs_traffic.sh TAG   then: python scripts/summarize_fbs_traffic.py TAG
set -o pipefail
TAG=${1:-r02d}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/fbstraffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROG="python3 $ROOT/scripts/fbs_prof.py"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $PROG > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $PROG > $OUT/pmc_write.log 2>&1 || exit 1
echo "fbs traffic $TAG done"
