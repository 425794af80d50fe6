cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fbs1 -- python3 $GRAFT_REPO_ROOT/scripts/fbs_prof.py > $GRAFT_REPO_ROOT/gpurun_out/fbs1.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/fbs1/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'ocs' in r['Name']: print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
