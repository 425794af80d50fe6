#!/bin/bash
# windows of an ODD batch: the tiled sweep kernels then run with an odd row distance (ld)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 300 python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
ocs = g.load_package()
from oracle import oracle
for nS, batch in ((4, 4099), (4, 4098), (2, 4099), (1, 4099), (4, 301)):
    N = 64
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    tspan = oracle.linspace(0, 1.0, N + 1)
    rng = np.random.default_rng(7)
    x0 = rng.uniform(0.8, 2.0, (nS, 5000))[:, :batch].copy(); cs = rng.uniform(1.0, 2.0, 5000)[:batch].copy()
    prob = ocs.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]]); prob.set_batch_params([0], cs[None, :])
    base = {"nERROR_PTS": N + 1, "nINTERP_PTS": 17, "nSWEEPS": 40}
    ra = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, nWINDOWS=2))
    rb = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, fused_update_off=1))
    bad = np.nonzero(ra["sweeps"] != rb["sweeps"])[0]
    err = [float(np.max(np.abs(ra[k] - rb[k]) / np.maximum(1, np.abs(rb[k])))) for k in ("x", "lam", "u", "J")]
    print(f"nS {nS} batch {batch} nWINDOWS=2 vs kernel-by-kernel: differing sweep counts {bad.size} {bad[:6].tolist()}; rel diff x/lam/u/J {['%.1e' % e for e in err]}")
PY
