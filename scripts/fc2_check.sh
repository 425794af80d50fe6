#!/bin/bash
# two-role lane kernels (k_forward_fc2 / k_backward_fc2) against the one-wave lane kernels: bit equality + timing
set -e
mkdir -p gpurun_out/fc2
B=${BATCHES:-100,8192,65536}
OCS_FC2=0 BATCHES=$B MODES=lane DUMP=gpurun_out/fc2/old python scripts/bl4_time.py
OCS_FC2=2 BATCHES=$B MODES=lane,on DUMP=gpurun_out/fc2/new python scripts/bl4_time.py
python - <<PY
import numpy as np
for b in "$B".split(','):
    a = np.load(f"gpurun_out/fc2/old_{b}_lane.npz"); c = np.load(f"gpurun_out/fc2/new_{b}_lane.npz")
    print(b, "J equal", np.array_equal(a['J'], c['J']), "dJdv equal", np.array_equal(a['dJdv'], c['dJdv']),
          "max rel", np.max(np.abs(a['dJdv']-c['dJdv'])/np.maximum(1,np.abs(a['dJdv']))))
PY
