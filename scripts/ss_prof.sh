#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ss_prof -- python3 $GRAFT_REPO_ROOT/scripts/ss_batch_time.py > $GRAFT_REPO_ROOT/gpurun_out/ss_prof.log 2>&1
f=$(ls -t $GRAFT_REPO_ROOT/gpurun_out/ss_prof/*/*kernel_stats.csv | head -1); head -12 $f | cut -d, -f1-4
