#!/bin/bash
# round 4, second GPU call: the whole -m gpu suite with the rewritten ocs_multi, host cost of the multi-device entry
# points, cache-policy variants of the adjoint scan / state pass in both buffer regimes, the streaming probe
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04b
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_gpu.log
OCS_MULTI_ALLOW_DUPLICATES=1 timeout -k 10 120 python scripts/multi_overhead.py > $OUT/multi_overhead.log 2>&1; cat $OUT/multi_overhead.log
for L in "" ldnt ldsc1 ldsc0sc1 xnt xsc1; do
  for r in 1 3; do
    echo "== lib ${L:-product} ROTATE=$r"
    OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} ROTATE=$r timeout -k 10 120 python scripts/pair_rotate.py 2>&1 | grep ROTATE | tail -2
  done
done > $OUT/policy_variants.log 2>&1
cat $OUT/policy_variants.log
timeout -k 10 200 scripts/probe/seg_stream > $OUT/seg_stream.log 2>&1; cat $OUT/seg_stream.log
