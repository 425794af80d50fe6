"""Randomised comparison of the folded fb_sweep kernels against the unfolded path (fused_update_off = 3) and, for a few
instances per case, the oracle (test infrastructure, not collected by pytest): python tests/stress_fold.py [ncases]"""
import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import __graft_entry__ as g
ocs = g.load_package()
from oracle import oracle
oracle.build()
P = {"c": 1.5, "m": 3.0, "r": 0.05}
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
worst = 0.0
for case in range(ncases):
    nS = int(rng.choice([1, 2, 4]))
    N = 8 * int(rng.integers(1, 40))
    batch = (64 // nS) * int(rng.integers(1, 5))
    T = float(rng.uniform(0.5, 6.0))
    kind = rng.integers(0, 3) if case % 4 == 3 else rng.integers(0, 2)   # (an irregular grid takes the unfolded path)
    if kind == 0:
        tspan = ocs.linspace(0, T, N + 1) if case % 3 else np.linspace(0, T, N + 1)   # MATLAB-style or numpy linspace
    elif kind == 1:
        tspan = np.arange(N + 1) * 2.0 ** -5
    else:
        tspan = np.sort(np.concatenate([[0.0, T], rng.uniform(0, T, N - 1)]))
        if np.min(np.diff(tspan)) < 1e-4 * T / N:
            tspan = np.linspace(0, T, N + 1)
    lbv = float(rng.choice([0.0, 0.1, -0.2])); ubv = lbv + float(rng.uniform(0.3, 1.2))
    if case % 2:   # every other case: an upper bound the control does not reach
        ubv = 6.0
    bounds = [[lbv, ubv]]
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    x0 = rng.uniform(0.8, 1.6, (nS, batch))
    cs = rng.uniform(1.0, 2.0, batch)
    prob = ocs.LogisticProblem(m, P["c"], P["r"], bounds)
    prob.set_batch_params([0], cs[None, :])
    base = {"nERROR_PTS": N + 1, "nINTERP_PTS": 17, "nSWEEPS": 60, "cost_row": int(rng.integers(0, 2))}
    ra = ocs.fb_sweep_batch(prob, x0, tspan, dict(base))
    rd = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, fused_update_off=3))
    ok = np.array_equal(ra["sweeps"], rd["sweeps"])
    err = 0.0
    for key in ("x", "lam", "u", "J"):
        a, b = np.asarray(ra[key], dtype=float), np.asarray(rd[key], dtype=float)
        fin = np.isfinite(b)
        if fin.any():
            err = max(err, float(np.max(np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin])))))
        ok = ok and np.array_equal(np.isfinite(a), fin)
    mca, mcd = ra["maxChange"], rd["maxChange"]
    ok = ok and np.array_equal(np.isnan(mca), np.isnan(mcd))
    mcerr = float(np.nanmax(np.abs(mca - mcd) / np.maximum(1.0, np.abs(mcd)))) if np.isfinite(mcd).any() else 0.0
    oerr = 0.0
    if oracle is not None and ra["sweeps"].min() > 0:
        for b_ in (0, batch - 1):
            ref = oracle.fb_sweep(oracle.LogisticProblem(m, cs[b_], P["r"], bounds), x0[:, b_], tspan, base)
            ok = ok and ref["_sweeps"] == ra["sweeps"][b_]
            if ref["_sweeps"] > 0:
                oerr = max(oerr, abs(ra["J"][b_] - ref["J"]) / abs(ref["J"]),
                           float(np.max(np.abs(ra["lam"][:, :, b_] - ref["lam"]) / np.maximum(1.0, np.abs(ref["lam"])))))
    worst = max(worst, err, oerr)
    print(f"case {case}: nS={nS} N={N} batch={batch} grid={kind} lb={lbv} sweeps {ra['sweeps'].min()}..{ra['sweeps'].max()} "
          f"fold-vs-plain {err:.2e} maxChange {mcerr:.2e} vs-oracle {oerr:.2e} {'ok' if ok and (ra['sweeps'].min() == 0 or (err < 1e-11 and mcerr < 1e-6 and oerr < 1e-10)) else 'FAIL'}", flush=True)
print("worst", worst)
