#!/bin/bash
# Round-4 evidence in one GPU call (through gpurun, repo root): bash scripts/profile_r04.sh [TAG]
#   everything of scripts/profile_r03.sh (headline kernel stats + calibrated traffic, BL-4, BL-5 at 8192, fb_sweep, SQ of BL-4)
#   + the BL-5 shard (1024 trajectories, time-parallel chunks): kernel stats, FETCH / WRITE, matrix-pipe counters
#   + the BL-2 pass pair over three rotating buffer sets: kernel stats, FETCH / WRITE
#   + SQ counters of the headline kernels (k_forward_p2, k_backward_scan)
# then, back home: python scripts/summarize_profile.py TAG && python scripts/summarize_r03.py TAG && python scripts/summarize_r04.py TAG
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
bash $ROOT/scripts/profile_r03.sh $TAG || echo "profile_r03 part failed"
cd /tmp && export TMPDIR=/tmp
pass() {   # pass NAME "ROCPROF ARGS" PROGRAM...
  local name=$1 args=$2; shift 2
  rocprofv3 $args --kernel-trace --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1 || echo "$name failed"
}
export BATCH=1024 REPS=4
pass bl5s_trace "--stats" $ROOT/scripts/lq_time.py
pass bl5s_fetch "--pmc FETCH_SIZE" $ROOT/scripts/lq_time.py
pass bl5s_write "--pmc WRITE_SIZE" $ROOT/scripts/lq_time.py
pass bl5s_mfma "--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES" $ROOT/scripts/lq_time.py
unset BATCH REPS
export ROTATE=3 K=60
pass rot_trace "--stats" $ROOT/scripts/pair_rotate.py
pass rot_fetch "--pmc FETCH_SIZE" $ROOT/scripts/pair_rotate.py
pass rot_write "--pmc WRITE_SIZE" $ROOT/scripts/pair_rotate.py
unset ROTATE K
BATCHES=4096 NSTEPS=1000 bash $ROOT/scripts/profile_sq.sh ${TAG}hl scripts/bench_passes.py
echo "profile_r04 $TAG done"
