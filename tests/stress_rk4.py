"""Randomised parity stress of the RK4 pass pair (compute_states / compute_adjoints, RK4Integrator.m:28-121) over shapes,
grids and mappings:  python tests/stress_rk4.py [ncases]   (SEED=... in the environment; run on an MI355X)

Every case draws nS in 1..4, a step count (whole blocks, remainders, fewer steps than a block), a batch (whole tiles and
ragged), a uniform or non-uniform grid, a mapping (automatic, lane, row-split, pipeline, scan), default or explicit lamT, and
compares x, J, lam, dJdu of a handful of trajectories with the CPU oracle (1e-12 relative to max(1, |ref|)); forced mappings
that the shape does not admit must be refused with OCS_ERR_UNSUPPORTED (-6), never mis-computed.  Also a pytest case
(tests/test_gpu_rk4_parity.py::test_randomised_pass_pair_stress) with a short case list."""
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
    nS = int(rng.integers(1, 5))
    N = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 12, 15, 16, 17, 24, 31, 33, 40, 63, 64, 65, 100, 128, 131, 250, 333, 1000]))
    tile = 64 // nS if nS != 3 else 64
    batch = int(rng.choice([1, 2, 5, tile, tile + 1, 2 * tile, 3 * tile - 1, 4 * tile, 130, 256, 257, 512]))
    # (steps inside RK4's stability region for these problems -- rates up to 3 --: the trajectories stay finite)
    return {"nS": nS, "N": N, "batch": batch, "T": round(N * float(rng.choice([0.002, 0.01, 0.03, 0.06])), 6),
            "uniform": bool(rng.integers(0, 2)), "mapping": str(rng.choice(["auto", "auto", "lane", "rowsplit", "pipeline", "scan"])),
            "lamT": bool(rng.integers(0, 2)), "seed": int(rng.integers(1 << 30))}


def run_case(ocs, oracle, c):
    rng = np.random.default_rng(c["seed"])
    nS, N, batch = c["nS"], c["N"], c["batch"]
    if c["uniform"]:
        tspan = oracle.linspace(0.0, c["T"], N + 1)
    else:
        w = rng.uniform(0.4, 1.6, N)
        tspan = np.concatenate([[0.0], np.cumsum(w) * (c["T"] / w.sum())])
    t = np.zeros(2 * N + 1)
    t[0::2], t[1::2] = tspan, (tspan[:-1] + tspan[1:]) / 2
    f, ph = rng.uniform(0, 1, batch), rng.uniform(0, 2 * np.pi, batch)
    u = np.asfortranarray(np.clip(0.25 + 0.2 * np.sin(2 * np.pi * f[None, :] * t[:, None] + ph[None, :]), 0, 1)[None, :, :])
    x0 = rng.uniform(0.8, 2.5, (nS, batch))
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    if nS == 1 and rng.integers(0, 2):
        pg, po = ocs.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]]), oracle.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]])
    else:
        pg, po = ocs.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]]), oracle.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]])
    g = ocs.RK4Integrator(tspan)
    if c["mapping"] != "auto":
        g.set_mapping(c["mapping"])
    lamT = rng.normal(size=(nS + 1, batch)) if c["lamT"] else None
    try:
        x, J = g.compute_states(pg, x0, u)
        lam, dJdu = g.compute_adjoints(pg, u, lamT) if lamT is not None else g.compute_adjoints(pg, u)
    except ocs.OcsError as e:
        if c["mapping"] != "auto" and e.code == -6:
            return "refused", 0.0
        raise
    worst = 0.0
    go = oracle.RK4Integrator(tspan)
    for b in sorted({0, batch // 2, batch - 1, int(rng.integers(batch))}):
        xo, Jo = go.compute_states(po, x0[:, b], u[:, :, b])
        if lamT is not None:
            lo, do = go.compute_adjoints(po, u[:, :, b], lamT[:, b])
        else:
            lo, do = go.compute_adjoints(po, u[:, :, b])
        worst = max(worst, relerr(x[:, :, b], xo), relerr(J[b], Jo), relerr(lam[:, :, b], lo), relerr(dJdu[:, :, b], do))
    return "ok" if worst < RTOL else "FAILED", worst


def run(ocs, oracle, ncases, seed=1, log=print):
    rng = np.random.default_rng(seed)
    failed, refused, worst_all = 0, 0, 0.0
    for k in range(ncases):
        c = draw(rng)
        verdict, worst = run_case(ocs, oracle, c)
        failed += verdict == "FAILED"
        refused += verdict == "refused"
        worst_all = max(worst_all, worst)
        log(f"case {k}: nS={c['nS']} N={c['N']} batch={c['batch']} T={c['T']} {'uniform' if c['uniform'] else 'non-uniform'} "
            f"mapping={c['mapping']} lamT={'given' if c['lamT'] else 'default'}: {verdict} {worst:.2e}")
    log(f"failed cases: {failed} of {ncases} ({refused} refused as unsupported for a forced mapping); worst {worst_all:.2e}")
    return failed, worst_all


if __name__ == "__main__":
    import __graft_entry__ as g
    from oracle import oracle as orc
    f, _ = run(g.load_package(), orc, int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(os.environ.get("SEED", "1")))
    sys.exit(1 if f else 0)
