#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lq_fbs_prof -o lqf -- python3 $GRAFT_REPO_ROOT/scripts/lq_fbs_time.py > $GRAFT_REPO_ROOT/gpurun_out/lq_fbs_prof.log 2>&1
python3 - <<PY
import csv, glob
f=sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/lq_fbs_prof/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:95]:95s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:10.1f} us")
PY
f=$(ls -t $GRAFT_REPO_ROOT/gpurun_out/lq_fbs_prof/*kernel_trace.csv | head -1)
python3 - <<PY
import csv
seen={}
for r in csv.DictReader(open("$f")):
    k=(r['Kernel_Name'][:60], r['VGPR_Count'], r['Accum_VGPR_Count'], r['Scratch_Size'])
    seen[k]=seen.get(k,0)+1
for k,v in seen.items():
    if 'UserP' in k[0]: print(k, v)
PY
