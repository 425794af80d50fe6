"""Randomised parity stress of the shooting objective and gradient, [J, dJdv] = nlpObjective(v) (functions/single_shooting.m:137-150
= compute_u, compute_states, compute_adjoints, compute_dJdv, free initial states):   python tests/stress_nlp.py [ncases]

Every case draws a control basis (PWLinear / PWConstant / Chebyshev), its size, nS in 1..4, a step count and a batch around the
block and tile sizes of the fused kernels, a uniform or non-uniform grid, free initial states or none, the fusion of the basis into the
RK4 kernels (automatic / off / forced), and compares J, dJdv (and the x0 written back) of sampled candidates with the CPU oracle at 1e-12.  A short
case list runs as tests/test_gpu_controls_shooting.py::test_randomised_objective_gradient_stress."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RTOL = 1e-12


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        e = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    return float("inf") if not np.all(np.isfinite(e)) else float(e.max(initial=0.0))


def draw(rng):
    kind = str(rng.choice(["PWLinearControl", "PWConstantControl", "ChebyshevControl"]))
    nS = int(rng.integers(1, 5))
    N = int(rng.choice([2, 3, 7, 8, 9, 16, 24, 31, 40, 64, 65, 100, 128, 250, 400, 1000]))
    nB = int(rng.choice([1, 2, 5, 8, 16, 17, 32]) if kind == "ChebyshevControl" else rng.choice([2, 3, 6, 11, 33, 101]))
    if kind != "ChebyshevControl":
        nB = max(2, min(nB, N))
    tile = 64 // nS if nS != 3 else 64
    batch = int(rng.choice([1, 3, tile, tile + 1, 2 * tile, 130, 256, 300, 1024]))
    nfree = int(rng.integers(0, nS + 1)) if rng.integers(0, 3) == 0 else 0
    return {"kind": kind, "nS": nS, "N": N, "nB": nB, "batch": batch, "h": float(rng.choice([0.002, 0.01, 0.03])),
            "uniform": bool(rng.integers(0, 2)), "free": sorted(rng.choice(np.arange(1, nS + 1), nfree, replace=False).tolist())[::-1],
            "fusion": str(rng.choice(["auto", "off", "on", "lane"])), "seed": int(rng.integers(1 << 30))}


def run_case(ocs, oracle, c):
    rng = np.random.default_rng(c["seed"])
    nS, N, nB, batch = c["nS"], c["N"], c["nB"], c["batch"]
    T = N * c["h"]
    if c["uniform"]:
        tspan = oracle.linspace(0.0, T, N + 1)
    else:
        w = rng.uniform(0.5, 1.5, N)
        tspan = np.concatenate([[0.0], np.cumsum(w) * (T / w.sum())])
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    pg, po = ocs.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]]), oracle.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]])
    g, go = ocs.RK4Integrator(tspan), oracle.RK4Integrator(tspan)
    cg, co = getattr(ocs, c["kind"])(g.t, nB, 1), getattr(oracle, c["kind"])(go.t, nB, 1)
    if c["fusion"] != "auto":
        cg.set_fusion(c["fusion"])
    if c["kind"] == "ChebyshevControl":
        V = 0.05 * rng.normal(size=(nB, batch)) / np.arange(1, nB + 1)[:, None]
        V[0] += 0.3
    else:
        V = rng.uniform(0.05, 0.45, (nB, batch))
    free = c["free"]
    if free:
        V = np.vstack([V, rng.uniform(0.9, 2.0, (len(free), batch))])
    x0 = rng.uniform(0.9, 2.0, (nS, batch))
    try:
        J, dJdv, x0n = ocs.nlp_objective(g, pg, cg, x0, V, FreeInitStates=free)
    except ocs.OcsError as e:
        if c["fusion"] in ("on", "lane") and e.code == -6:     # a forced fused mapping the shape does not admit: refused
            return "refused", 0.0
        raise
    worst = 0.0
    for b in sorted({0, batch // 2, batch - 1, int(rng.integers(batch))}):
        Jo, do, x0o = oracle.nlp_objective(go, po, co, x0[:, b], V[:, b], FreeInitStates=free)
        worst = max(worst, relerr(J[b], Jo), relerr(dJdv[:, b], do), 0.0 if np.array_equal(x0n[:, b], x0o) else float("inf"))
    return "ok" if worst < RTOL else "FAILED", worst


def run(ocs, oracle, ncases, seed=1, log=print):
    rng = np.random.default_rng(seed)
    failed, worst_all = 0, 0.0
    for k in range(ncases):
        c = draw(rng)
        verdict, worst = run_case(ocs, oracle, c)
        failed += verdict == "FAILED"
        worst_all = max(worst_all, worst)
        log(f"case {k}: {c['kind']} nBasis={c['nB']} nS={c['nS']} N={c['N']} batch={c['batch']} h={c['h']} "
            f"{'uniform' if c['uniform'] else 'non-uniform'} free={c['free']} fusion={c['fusion']}: {verdict} {worst:.2e}")
    log(f"failed cases: {failed} of {ncases}; worst {worst_all:.2e}")
    return failed, worst_all


if __name__ == "__main__":
    import __graft_entry__ as g
    from oracle import oracle as orc
    f, _ = run(g.load_package(), orc, int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(os.environ.get("SEED", "1")))
    sys.exit(1 if f else 0)
