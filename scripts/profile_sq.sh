#!/bin/bash
# SQ counter passes (rocprofv3 --pmc, kernel trace only) over a python program; one pass per counter group (8 SQ
# slots per pass on gfx950).  Usage (through gpurun, from the repo root):
#   bash scripts/profile_sq.sh TAG scripts/bench_passes.py      (environment variables select the workload)
# Results: gpurun_out/sq_TAG/passN/...counter_collection.csv; scripts/summarize_sq.py turns them into profiles/TAG_sq.json
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
PROG="$ROOT/$1"; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH"
P4="GRBM_GUI_ACTIVE"
n=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  n=$((n+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$n -- python3 $PROG "$@" > $OUT/pass$n.log 2>&1 || echo "pass $n failed (see pass$n.log)"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $PROG "$@" > $OUT/trace.log 2>&1 || echo "trace failed"
echo "sq profile $TAG done"
