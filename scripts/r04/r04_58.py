import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import __graft_entry__ as g
ocs = g.load_package()
from oracle import oracle
from tests import stress_fold
for ragged in (False, True):
    rng = np.random.default_rng(71)
    for case in range(56):
        c = stress_fold.draw_case(rng, case, ocs, ragged=ragged)
    print("ragged", ragged, "case 55: nS", c["nS"], "N", c["N"], "batch", c["batch"], "kind", c["kind"], "bounds", c["bounds"], "max h", np.max(np.diff(c["tspan"])))
    prob = ocs.LogisticProblem(c["m"], 1.5, 0.05, c["bounds"]); prob.set_batch_params([0], c["cs"][None, :])
    base = {"nERROR_PTS": c["N"] + 1, "nINTERP_PTS": 17, "nSWEEPS": 60, "cost_row": c["cost_row"]}
    for fuo in (0, 3, 1):
        r = ocs.fb_sweep_batch(prob, c["x0"], c["tspan"], dict(base, fused_update_off=fuo))
        sw = r["sweeps"]
        nanx = np.isnan(r["x"]).any(axis=(0, 1)); conv = sw > 0
        print("  fused_update_off", fuo, "converged", int(conv.sum()), "converged with NaN in x", int((conv & nanx).sum()), "first sweeps", sw[:8].tolist(), "J[0]", r["J"][0])
    ref = oracle.fb_sweep(oracle.LogisticProblem(c["m"], c["cs"][0], 0.05, c["bounds"]), c["x0"][:, 0], c["tspan"], base)
    print("  oracle instance 0: sweeps", ref["_sweeps"], "J", ref.get("J"))
