#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fbs_offnode_prof -o offn -- python3 $GRAFT_REPO_ROOT/scripts/fbs_offnode_time.py > $GRAFT_REPO_ROOT/gpurun_out/fbs_offnode_prof.log 2>&1
python3 - <<PY
import csv, glob
f=sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/fbs_offnode_prof/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'][:100]:100s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.1f} ms")
PY
