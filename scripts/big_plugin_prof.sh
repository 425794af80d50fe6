#!/bin/bash
# kernel times of the pass pair and of fb_sweep on the six-state plugin (scripts/big_plugin_time.py)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/big_plugin_prof -o bigp -- python3 $GRAFT_REPO_ROOT/scripts/big_plugin_time.py > $GRAFT_REPO_ROOT/gpurun_out/big_plugin_prof.log 2>&1
f=$(ls -t $GRAFT_REPO_ROOT/gpurun_out/big_plugin_prof/*kernel_stats.csv | head -1)
cut -d, -f1-6 $f | head -12
