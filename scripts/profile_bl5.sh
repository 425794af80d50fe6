#!/bin/bash
# BL-5 evidence for the current LQ kernels (through gpurun, repo root): kernel stats, FETCH / WRITE, matrix-pipe counters of
# scripts/lq_time.py (LQ32, nC = 4, 4000 + 4000 steps, batch 8192);  then: python scripts/summarize_bl5.py TAG
TAG=${1:-r03e}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_${TAG}_bl5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1 args=$2; shift 2
  rocprofv3 $args --kernel-trace --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1 || echo "$name failed"
}
pass trace "--stats" $ROOT/scripts/lq_time.py
pass fetch "--pmc FETCH_SIZE" $ROOT/scripts/lq_time.py
pass write "--pmc WRITE_SIZE" $ROOT/scripts/lq_time.py
pass mfma "--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES" $ROOT/scripts/lq_time.py
pass valu "--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY" $ROOT/scripts/lq_time.py
echo "profile_bl5 $TAG done"
