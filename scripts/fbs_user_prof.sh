#!/bin/bash
# kernel times of fb_sweep: registry problem / hipRTC user problems (scripts/fbs_user_time.py)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/fbs_user_prof -o fbsu -- python $GRAFT_REPO_ROOT/scripts/fbs_user_time.py > $GRAFT_REPO_ROOT/gpurun_out/fbs_user_prof.log 2>&1
