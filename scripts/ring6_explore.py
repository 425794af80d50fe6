"""fb_sweep convergence of the six-state ring problem (tests/user_problems.ring6_symbolic) by horizon and damping."""
import os, sys, importlib, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
ocs = g.load_package()
sym = importlib.import_module("ocs_amd.symbolic")
from tests.user_problems import ring6_symbolic
gg, f, vals = ring6_symbolic(sym)
prob = ocs.make_from_symbolic(gg, f, 6, 3, vals, [[0.0, 1.0]] * 3)
rng = np.random.default_rng(66)
X0 = rng.uniform(0.6, 1.8, (6, 96))
for T in (2.0, 4.0):
    for om in (0.0, 0.2, 0.35, 0.5):
        N = 160
        ts = ocs.linspace(0, T, N + 1)
        r = ocs.fb_sweep_batch(prob, X0, ts, {"nERROR_PTS": N + 1, "nINTERP_PTS": 41, "nSWEEPS": 120, "uRelax": om})
        sw = r["sweeps"]
        print(f"T={T} uRelax={om}: converged {np.mean(sw > 0):.2f}, sweeps {sw[sw > 0].min() if (sw > 0).any() else 0}..{sw.max()}", flush=True)
