"""Generates tests/golden/*.json from the CPU oracle (oracle/ocs_oracle.c) on fixed seeded inputs.

These are NOT reference outputs: the reference (MATLAB) cannot run in this pipeline and ships no
golden vectors; parity with MATLAB itself stays unpinned.  The fixtures freeze the
oracle's answers for BASELINE configs BL-1..BL-4 (SURVEY KAT 8) so that (a) an accidental change of
the oracle is caught on CPU and (b) the GPU path is checked against numbers that do not depend on
the oracle being rebuilt on the GPU box.  Small by design: first/last 3 time columns, J and dJdv of
a few sampled trajectories.  Regenerate with:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as o  # noqa: E402

P = {"c": 1.5, "m": 3.0, "r": 0.05}
BOUNDS = [[0.0, 1.0]]
US = 0.72336878009798256


def cols(a):
    a = np.atleast_2d(a)
    return {"first3": a[:, :3].tolist(), "last3": a[:, -3:].tolist()}


def bl2_inputs(batch, N=1000, seed=20260401):
    rng = np.random.default_rng(seed)
    tspan = o.linspace(0.0, 10.0, N + 1)
    t = o.RK4Integrator(tspan).t
    f, ph = rng.uniform(0, 1, batch), rng.uniform(0, 2 * np.pi, batch)
    u = np.clip(0.25 + 0.2 * np.sin(2 * np.pi * f[None, :] * t[:, None] + ph[None, :]), 0, 1)[None]
    return tspan, np.ones((4, batch)), np.asfortranarray(u)


def main():
    out = {}
    # BL-1: tests/solve_test_problem.m:5-18, v0 = u* and two perturbed iterates
    tspan = o.linspace(0, 10, 501)
    prob, g = o.TestOCProblem(P, BOUNDS), o.RK4Integrator(tspan)
    c = o.PWLinearControl(g.t, 101, 1)
    v0 = c.compute_initial_v([US])
    bl1 = []
    for k, dv in enumerate((0.0, 0.05, -0.05)):
        v = np.clip(v0 + dv * np.cos(np.arange(101) * 0.3), 0, 1)
        J, dJdv, _ = o.nlp_objective(g, prob, c, [1.0], v)
        u = c.compute_u(v)
        x, _ = g.compute_states(prob, [1.0], u)
        lam, dJdu = g.compute_adjoints(prob, u)
        bl1.append({"dv": dv, "J": J, "dJdv": dJdv.tolist(), "x": cols(x), "lam": cols(lam), "dJdu": cols(dJdu)})
    out["BL1"] = bl1
    # BL-2: Logistic4, N=1000, batch 4096, sampled trajectories
    tspan, x0, u = bl2_inputs(4096)
    idx = [0, 1, 63, 64, 2047, 4095]
    pl = o.LogisticProblem([3.0, 2.5, 2.0, 1.5], P["c"], P["r"], BOUNDS)
    r = o.batch_states_adjoints(pl, tspan, x0[:, idx], u[:, :, idx])
    out["BL2"] = {"idx": idx, "J": r["J"].tolist(),
                  "traj": [{"x": cols(r["x"][:, :, k]), "lam": cols(r["lam"][:, :, k]), "dJdu": cols(r["dJdu"][:, :, k])}
                           for k in range(len(idx))]}
    # BL-3: fb_sweep, batch 16384 instances, sampled
    rng = np.random.default_rng(20260402)
    x0s, cs = rng.uniform(0.5, 2.5, (1, 16384)), rng.uniform(1.0, 2.0, 16384)
    tspan = o.linspace(0, 10, 1001)
    bl3 = []
    for b in (0, 1, 8191, 16383):
        s = o.fb_sweep(o.TestOCProblem({"c": cs[b], "m": 3.0, "r": 0.05}, BOUNDS), x0s[:, b], tspan)
        bl3.append({"b": b, "x0": float(x0s[0, b]), "c": float(cs[b]), "sweeps": int(s["_sweeps"]), "J": s["J"],
                    "maxChange": s["_maxChange"][: s["_sweeps"]].tolist(), "u": s["u"][:, ::100].tolist(),
                    "x": s["x"][:, ::100].tolist(), "lam": s["lam"][:, ::100].tolist()})
    out["BL3"] = bl3
    # BL-4: Chebyshev-16 objective + gradient, batch 65536, sampled
    rng = np.random.default_rng(20260403)
    V = 0.05 * rng.normal(size=(16, 65536)) / np.arange(1, 17)[:, None]
    V[0] += 0.5
    g = o.RK4Integrator(tspan)
    cc = o.ChebyshevControl(g.t, 16, 1)
    bl4 = []
    for b in (0, 1, 32767, 65535):
        J, dJdv, _ = o.nlp_objective(g, prob, cc, [1.0], V[:, b])
        bl4.append({"b": b, "J": J, "dJdv": dJdv.tolist()})
    out["BL4"] = bl4
    with open(os.path.join(HERE, "oracle_goldens.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote", os.path.join(HERE, "oracle_goldens.json"), os.path.getsize(os.path.join(HERE, "oracle_goldens.json")), "bytes")


if __name__ == "__main__":
    main()
