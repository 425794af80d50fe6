#!/bin/bash
# One GPU call that collects the round's evidence: kernel stats + calibrated FETCH/WRITE traffic of bench.py's headline
# kernels (profile_round.sh), SQ counters of the BL-2 pass pair (profile_sq.sh), kernel stats of the fb_sweep solve.
#   bash scripts/profile_all.sh TAG      then, back home: python scripts/summarize_profile.py TAG; python scripts/summarize_sq.py TAG
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash $ROOT/scripts/profile_round.sh $TAG
NSTEPS=1000 BATCHES=4096 bash $ROOT/scripts/profile_sq.sh $TAG scripts/bench_passes.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/fbs_$TAG -- python3 $ROOT/scripts/fbs_prof.py > $ROOT/gpurun_out/fbs_$TAG.log 2>&1
bash $ROOT/scripts/profile_sq.sh ${TAG}fbs scripts/fbs_prof.py    # SQ counters of the fb_sweep kernels
echo "profile_all $TAG done"
