#!/bin/bash
# Tuning builds of libocs.so that differ from the product in ONE object: lib/libocs_<name>.so from
#   name:object:"-D flags"   triples (the other objects are the product's).  The variants are git-ignored and travel to the
# GPU box with the snapshot; scripts select one through OCS_LIB_OVERRIDE.  Usage: bash scripts/build_variants.sh (edit VARIANTS)
set -e
cd "$(dirname "$0")/../optimal-control-solvers_amd/csrc"
make -s
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-parameter"
OBJS=$(ls _obj/*.o | grep -v "_var_")
VARIANTS=${VARIANTS:-"ldnt:ocs_scan_kernels:-DOCS_SCAN_LD_AUX=2 ldsc1:ocs_scan_kernels:-DOCS_SCAN_LD_AUX=16 ldsc0sc1:ocs_scan_kernels:-DOCS_SCAN_LD_AUX=17 xnt:ocs_pipeline2_kernels:-DOCS_P2_X_ST_AUX=2 xsc1:ocs_pipeline2_kernels:-DOCS_P2_X_ST_AUX=16"}
for v in $VARIANTS; do
  name=${v%%:*}; rest=${v#*:}; obj=${rest%%:*}; defs=${rest#*:}; defs=${defs//,/ }
  extra=""
  [ "$obj" = "ocs_lq_kernels" ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  $HIPCC $FLAGS $extra $defs -c -o _obj/_var_${name}.o $obj.hip
  others=$(echo "$OBJS" | grep -v "_obj/$obj.o")
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../lib/libocs_${name}.so $others _obj/_var_${name}.o -ldl -lpthread
  echo "built lib/libocs_${name}.so ($obj with $defs)"
done
