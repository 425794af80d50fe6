#!/bin/bash
# round 4, seventh GPU call: bench.py with the live traffic collection, new tests, then the evidence of scripts/profile_r04.sh
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04i
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err || tail -20 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04i/bench.json"))
print("headline", d["ms_per_step"], d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], (d["roofline"]["traffic_source"] or "")[:60])
print("live", json.dumps(d.get("traffic_collected_by_this_run"))[:900])
PY
timeout -k 10 300 python -m pytest tests/test_gpu_user_problems.py -m gpu -q -k "false_declarations or six_state" 2>&1 | tail -3
echo "== profile_r04"; date
timeout -k 10 900 bash scripts/profile_r04.sh r04 > $OUT/profile_r04.log 2>&1; tail -5 $OUT/profile_r04.log; date
