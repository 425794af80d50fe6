#!/bin/bash
# round 4, sixth GPU call: the checkpoint-recompute variant of the adjoint scan (lib/libocs_xrc.so) -- parity, then both
# buffer regimes against the product; the two failing tests of the fifth call
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04f
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_controls_shooting.py tests/test_gpu_user_problems.py -m gpu -q -k "constraint_hooks or generated_from_symbols" > $OUT/pytest_new.log 2>&1; echo "pytest new rc $?"; tail -4 $OUT/pytest_new.log
OCS_LIB_OVERRIDE=$ROOT/optimal-control-solvers_amd/lib/libocs_xrc.so timeout -k 10 600 python -m pytest tests/test_gpu_rk4_parity.py tests/test_golden.py -m gpu -q > $OUT/pytest_xrc.log 2>&1; echo "pytest xrc rc $?"; tail -4 $OUT/pytest_xrc.log
for L in "" xrc; do
  for r in 1 3; do
    echo "== lib ${L:-product} ROTATE=$r"
    OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} ROTATE=$r timeout -k 10 120 python scripts/pair_rotate.py 2>&1 | grep ROTATE | tail -2
  done
  echo "== lib ${L:-product} nS=1 ROTATE=1 / 3"
  for r in 1 3; do OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} NS=1 ROTATE=$r timeout -k 10 120 python scripts/pair_rotate.py 2>&1 | grep ROTATE | tail -1; done
done > $OUT/xrc.log 2>&1
cat $OUT/xrc.log
