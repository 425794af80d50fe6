#!/bin/bash
# round 4, fifth GPU call: the whole -m gpu suite, then bench.py
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04e
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -12 $OUT/pytest_gpu.log
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || tail -20 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04e/bench.json"))
print("headline", d["ms_per_step"], d["roofline"]["frac"], "fb", d["fb_sweep"]["batch_sweeps_per_s"], d["fb_sweep"]["ms_per_solve"]["median"])
o=d["other_configs"]
for k,v in o.items():
    if "strong" in k:
        for q in ("BL-2","BL-3 fb_sweep","BL-4","BL-5"): print(q, {a:b for a,b in v[q].items() if not isinstance(b,dict)})
    else: print(k[:40], {a:b for a,b in v.items() if a.startswith("ms_") and not isinstance(b,dict)})
PY
