"""Committed fixtures tests/golden/oracle_goldens.json (made by tests/golden/make_golden.py from the
CPU oracle; NOT reference outputs -- parity with MATLAB is unpinned).  CPU: the oracle still reproduces
them.  GPU: the HIP path, through the C-ABI, reproduces them at BASELINE's full sizes."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "oracle_goldens.json")))
P = {"c": 1.5, "m": 3.0, "r": 0.05}
BOUNDS = [[0.0, 1.0]]
US = 0.72336878009798256


def close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) < tol


def cols_ok(arr, ref, tol):
    arr = np.atleast_2d(arr)
    return close(arr[:, :3], ref["first3"], tol) and close(arr[:, -3:], ref["last3"], tol)


def _bl1(mod, tol):
    orc_lin = np.linspace  # noqa: F841
    from oracle import oracle as o
    tspan = o.linspace(0, 10, 501)
    prob, g = mod.TestOCProblem(P, BOUNDS), mod.RK4Integrator(tspan)
    c = mod.PWLinearControl(g.t, 101, 1)
    v0 = c.compute_initial_v([US])
    for ref in G["BL1"]:
        v = np.clip(v0 + ref["dv"] * np.cos(np.arange(101) * 0.3), 0, 1)
        u = c.compute_u(v)
        x, J = g.compute_states(prob, [1.0], u)
        lam, dJdu = g.compute_adjoints(prob, u)
        dJdv = c.compute_dJdv(dJdu)
        assert abs(J - ref["J"]) < tol * abs(ref["J"]) and close(dJdv, ref["dJdv"], tol)
        assert cols_ok(x, ref["x"], tol) and cols_ok(lam, ref["lam"], tol) and cols_ok(dJdu, ref["dJdu"], tol)


def test_oracle_reproduces_goldens_bl1(oracle):
    _bl1(oracle, 1e-15)


def test_oracle_reproduces_goldens_bl4(oracle):
    tspan = oracle.linspace(0, 10, 1001)
    rng = np.random.default_rng(20260403)
    V = 0.05 * rng.normal(size=(16, 65536)) / np.arange(1, 17)[:, None]
    V[0] += 0.5
    g = oracle.RK4Integrator(tspan)
    prob, cc = oracle.TestOCProblem(P, BOUNDS), oracle.ChebyshevControl(g.t, 16, 1)
    for ref in G["BL4"]:
        J, dJdv, _ = oracle.nlp_objective(g, prob, cc, [1.0], V[:, ref["b"]])
        assert J == ref["J"] and np.array_equal(dJdv, ref["dJdv"])


@pytest.fixture(scope="module")
def ocs():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    return g.load_package()


@pytest.mark.gpu
def test_gpu_reproduces_goldens_bl1(ocs):
    _bl1(ocs, 1e-12)


@pytest.mark.gpu
def test_gpu_reproduces_goldens_bl2_full_size(ocs, oracle):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden import bl2_inputs
    tspan, x0, u = bl2_inputs(4096)
    prob = ocs.LogisticProblem([3.0, 2.5, 2.0, 1.5], P["c"], P["r"], BOUNDS)
    g = ocs.RK4Integrator(tspan)
    x, J = g.compute_states(prob, x0, u)
    lam, dJdu = g.compute_adjoints(prob, u)
    ref = G["BL2"]
    for k, b in enumerate(ref["idx"]):
        assert abs(J[b] - ref["J"][k]) < 1e-12 * abs(ref["J"][k])
        t = ref["traj"][k]
        assert cols_ok(x[:, :, b], t["x"], 1e-12) and cols_ok(lam[:, :, b], t["lam"], 1e-12)
        assert cols_ok(dJdu[:, :, b], t["dJdu"], 1e-12)


@pytest.mark.gpu
def test_gpu_reproduces_goldens_bl3_full_size(ocs):
    import torch
    rng = np.random.default_rng(20260402)
    x0s, cs = rng.uniform(0.5, 2.5, (1, 16384)), rng.uniform(1.0, 2.0, 16384)
    prob = ocs.TestOCProblem(P, BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    integ = ocs.RK4Integrator(np.linspace(0, 10, 1001))
    r = ocs.fb_sweep_dev(prob, integ, torch.tensor(x0s, device="cuda:0"))
    torch.cuda.synchronize()
    sw, J = r["sweeps"].cpu().numpy(), r["J"].cpu().numpy()
    u, lam, x = r["u"].cpu().numpy(), r["lam"].cpu().numpy(), r["xaug"].cpu().numpy()
    mc = r["maxChange"].cpu().numpy()
    for ref in G["BL3"]:
        b = ref["b"]
        assert sw[b] == ref["sweeps"] and abs(J[b] - ref["J"]) < 1e-10 * abs(ref["J"])
        assert close(mc[: ref["sweeps"], b], ref["maxChange"], 1e-6)
        assert close(u[::100, 0, b], ref["u"][0], 1e-10) and close(lam[::100, 0, b], ref["lam"][0], 1e-10)
        assert close(x[::100, 0, b], ref["x"][0], 1e-10)


@pytest.mark.gpu
def test_gpu_reproduces_goldens_bl4_full_size(ocs):
    import torch
    rng = np.random.default_rng(20260403)
    V = 0.05 * rng.normal(size=(16, 65536)) / np.arange(1, 17)[:, None]
    V[0] += 0.5
    g = ocs.RK4Integrator(np.linspace(0, 10, 1001))
    prob, cc = ocs.TestOCProblem(P, BOUNDS), ocs.ChebyshevControl(g.t, 16, 1)
    dev = torch.device("cuda:0")
    Jd, gd = ocs.nlp_objective_dev(g, prob, cc, torch.ones((1, 65536), dtype=torch.float64, device=dev),
                                   torch.tensor(V, device=dev))
    torch.cuda.synchronize()
    J, dJdv = Jd.cpu().numpy(), gd.cpu().numpy()
    assert np.all(np.isfinite(J)) and np.all(np.isfinite(dJdv))
    for ref in G["BL4"]:
        b = ref["b"]
        assert abs(J[b] - ref["J"]) < 1e-12 * abs(ref["J"]) and close(dJdv[:, b], ref["dJdv"], 1e-12)
