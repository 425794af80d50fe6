#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes + kernel stats of the two secondary BL-2 entries of bench.py (through gpurun, repo root):
#   the lane mapping at batch 65536 (Logistic4) and the one-state shapes at batch 4096;  then: python scripts/summarize_variants.py TAG
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_${TAG}v
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1 args=$2; shift 2
  rocprofv3 $args --kernel-trace --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1 || echo "$name failed"
}
export NS=4 BATCH=65536 MAPPING=lane
pass lane65536_trace "--stats" $ROOT/scripts/pair_loop.py
pass lane65536_fetch "--pmc FETCH_SIZE" $ROOT/scripts/pair_loop.py
pass lane65536_write "--pmc WRITE_SIZE" $ROOT/scripts/pair_loop.py
export NS=1 BATCH=4096 MAPPING=auto
pass ns1_trace "--stats" $ROOT/scripts/pair_loop.py
pass ns1_fetch "--pmc FETCH_SIZE" $ROOT/scripts/pair_loop.py
pass ns1_write "--pmc WRITE_SIZE" $ROOT/scripts/pair_loop.py
echo "profile_variants $TAG done"
