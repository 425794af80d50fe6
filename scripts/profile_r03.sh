#!/bin/bash
# Round-3 evidence in one GPU call (run through gpurun from the repo root): bash scripts/profile_r03.sh [TAG]
#   headline (bench.py): kernel stats + calibrated FETCH/WRITE traffic            (profile_round.sh)
#   BL-4 at 8192 and 65536 candidates: kernel stats, FETCH_SIZE / WRITE_SIZE passes
#   BL-5 (LQ32, 4000 + 4000 steps, batch 8192): kernel stats, FETCH / WRITE, matrix-pipe counters
#   fb_sweep BL-3: kernel stats, FETCH / WRITE
#   SQ counters of the BL-4 kernels at 8192
# then, back home: python scripts/summarize_r03.py TAG
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
bash $ROOT/scripts/profile_round.sh $TAG || echo "profile_round failed"
cd /tmp && export TMPDIR=/tmp
pass() {   # pass NAME "ROCPROF ARGS" PROGRAM...
  local name=$1 args=$2; shift 2
  rocprofv3 $args --kernel-trace --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1 || echo "$name failed"
}
for B in 8192 65536; do
  export BATCHES=$B MODES=auto REPS=10
  pass bl4_${B}_trace "--stats" $ROOT/scripts/bl4_time.py
  pass bl4_${B}_fetch "--pmc FETCH_SIZE" $ROOT/scripts/bl4_time.py
  pass bl4_${B}_write "--pmc WRITE_SIZE" $ROOT/scripts/bl4_time.py
done
unset BATCHES MODES REPS
pass bl5_trace "--stats" $ROOT/scripts/lq_time.py
pass bl5_fetch "--pmc FETCH_SIZE" $ROOT/scripts/lq_time.py
pass bl5_write "--pmc WRITE_SIZE" $ROOT/scripts/lq_time.py
pass bl5_mfma "--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES" $ROOT/scripts/lq_time.py
pass fbs_trace "--stats" $ROOT/scripts/fbs_prof.py
pass fbs_fetch "--pmc FETCH_SIZE" $ROOT/scripts/fbs_prof.py
pass fbs_write "--pmc WRITE_SIZE" $ROOT/scripts/fbs_prof.py
BATCHES=8192 MODES=auto REPS=5 bash $ROOT/scripts/profile_sq.sh ${TAG}bl4 scripts/bl4_time.py
echo "profile_r03 $TAG done"
