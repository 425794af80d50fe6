#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
for rep in 1 2 3; do
for L in "" notile; do
  echo -n "state pass, batch 4096, lib ${L:-product}: "
  OCS_LIB_OVERRIDE=${L:+$ROOT/optimal-control-solvers_amd/lib/libocs_$L.so} NS=4 BATCH=4096 timeout -k 10 200 python scripts/pair_loop.py 2>&1 | grep "^f " | tr '\n' ' '; echo
done; done | tee gpurun_out/p2_tile_base_ab.log
