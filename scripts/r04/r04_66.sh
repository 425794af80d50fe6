#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
mkdir -p gpurun_out
OCS_CONTROL_PTS_SORTED=0 timeout -k 10 300 python scripts/r04/offnode_ab.py /tmp/offn_legacy.npz 2>&1 | tail -1
timeout -k 10 300 python scripts/r04/offnode_ab.py /tmp/offn_sorted.npz 2>&1 | tail -1
python - <<'PY'
import numpy as np
a, b = np.load("/tmp/offn_legacy.npz"), np.load("/tmp/offn_sorted.npz")
worst, bad = 0.0, 0
for k in a.files:
    if k.endswith("sweeps"):
        if not np.array_equal(a[k], b[k]): bad += 1; print("sweep counts differ in case", k, np.nonzero(a[k] != b[k])[0][:5])
    else:
        ok = np.isfinite(b[k])
        if ok.any(): worst = max(worst, float(np.max(np.abs(a[k][ok] - b[k][ok]) / np.maximum(1, np.abs(b[k][ok])))))
print(f"sorted error-point kernel against the point-by-point one, 24 random cases: sweep-count mismatches {bad}, worst relative difference {worst:.2e}")
PY
