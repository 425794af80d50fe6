"""fb_sweep with error points off the nodes on random shapes; writes the results to an .npz (run with OCS_CONTROL_PTS_SORTED=0 and
without, then compare: scripts/r04/r04_66.sh)"""
import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import __graft_entry__ as g
ocs = g.load_package()
out = {}
rng = np.random.default_rng(5)
for case in range(24):
    nS = int(rng.integers(1, 5)); N = int(rng.choice([8, 24, 40, 64, 100, 131])); batch = int(rng.choice([1, 5, 64, 70, 200]))
    nE = int(rng.choice([7, 33, N, N + 2, 2 * N + 1, 3 * N]))
    T = N * float(rng.choice([0.01, 0.02]))
    tspan = ocs.linspace(0, T, N + 1) if case % 2 else np.concatenate([[0.0], np.cumsum(rng.uniform(0.5, 1.5, N))]) * (T / N)
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    prob = ocs.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]])
    x0 = rng.uniform(0.8, 2.0, (nS, batch))
    r = ocs.fb_sweep_batch(prob, x0, tspan, {"nERROR_PTS": nE, "nINTERP_PTS": 17, "nSWEEPS": 30})
    for k in ("x", "lam", "u", "J", "sweeps"):
        out[f"{case}_{k}"] = np.asarray(r[k])
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1])
