#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04t
mkdir -p $OUT
cd $ROOT
for ch in "" 1 2 4; do
  echo "== OCS_COSTATE_CH=${ch:-default}"
  OCS_JIT_DEFINES=${ch:+OCS_COSTATE_CH=$ch} timeout -k 10 200 python scripts/big_plugin_time.py 2>&1 | grep "fb_sweep\|adjoint" | tail -2
done > $OUT/costate_chunk.log 2>&1
cat $OUT/costate_chunk.log
