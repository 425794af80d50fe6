#!/bin/bash
# A/B of two builds of the library on the headline pass pair and the folded sweep:  ALT=<path to the other libocs.so> bash scripts/ab_lib.sh
ALT=${ALT:-optimal-control-solvers_amd/lib/libocs_prio.so}
for rep in 1 2; do
  for L in "" "$ALT"; do
    echo "== lib ${L:-default} (rep $rep)"
    OCS_LIB_OVERRIDE=$L NS=4 BATCH=4096 python scripts/pair_loop.py 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
    OCS_LIB_OVERRIDE=$L python scripts/fold_time.py 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1
  done
done
