#!/bin/bash
# round 4, tenth GPU call: role placement variants of the sweep's state pass (a control wave on the recursion wave's SIMD, with and
# without priority for the recursion wave); LQ tests with the tail-leg adjoint chunked; kernel times of a sweep per variant
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04m
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_lq.py -m gpu -q > $OUT/pytest_lq.log 2>&1; echo "pytest lq rc $?"; tail -3 $OUT/pytest_lq.log
for L in "" rm1 rm1p rm2 rm2p; do
  echo "== lib ${L:-product}"
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} timeout -k 10 120 python scripts/fold_time.py 2>&1 | grep "per sweep" | tail -2
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} timeout -k 10 120 python scripts/fbs_time.py 2>&1 | grep solve | tail -1
done > $OUT/rolemap.log 2>&1
cat $OUT/rolemap.log
cd /tmp && export TMPDIR=/tmp
for L in "" rm2p; do
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_${L:-product} -- python3 $ROOT/scripts/fold_time.py > $OUT/trace_${L:-product}.log 2>&1
  f=$(ls -t $OUT/trace_${L:-product}/*/*kernel_stats.csv | head -1); head -3 $f | cut -d, -f1-4 | cut -c1-120
done
