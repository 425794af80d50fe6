"""The multi-device entry points of the C-ABI (include/ocs.h ocs_multi_*) on the one GPU a test box has:
one communicator over one device (ncclCommInitAll), every entry point against the one-device entry point of the same
name (bit-equal: the same kernels on the same block), the RCCL reductions against numpy.  N > 1 devices cannot be run
here; the block arithmetic that N > 1 adds is covered without a GPU in tests/test_host_logic.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = {"c": 1.5, "m": 3.0, "r": 0.05}
BOUNDS = [[0.0, 1.0]]


@pytest.fixture(scope="module")
def ocs():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    return g.load_package()


def test_multi_device_equals_single_device(ocs, oracle):
    md = ocs.MultiDevice([0])
    assert md.size == 1 and md.shard(1000, 0) == (0, 1000)
    N, B, nB = 64, 200, 12
    tspan = oracle.linspace(0, 4, N + 1)
    rng = np.random.default_rng(3)
    integs = md.replicate(lambda: ocs.RK4Integrator(tspan))
    probs = md.replicate(lambda: ocs.LogisticProblem([3.0, 2.5], P["c"], P["r"], BOUNDS))
    g1, p1 = ocs.RK4Integrator(tspan), ocs.LogisticProblem([3.0, 2.5], P["c"], P["r"], BOUNDS)
    x0 = rng.uniform(0.8, 1.6, (2, B))
    u = rng.uniform(0.0, 0.6, (1, 2 * N + 1, B))
    x, J, st, rc = md.compute_states(integs, probs, x0, u)
    lam, dJdu = md.compute_adjoints(integs, probs, u)
    xr, Jr = g1.compute_states(p1, x0, u)
    lamr, dr = g1.compute_adjoints(p1, u)
    assert rc == 0 and np.array_equal(x, xr) and np.array_equal(J, Jr) and np.array_equal(lam, lamr) and np.array_equal(dJdu, dr)
    assert st["count"] == B and st["argmin"] == int(np.argmin(Jr)) and st["min_J"] == Jr.min()
    assert abs(st["sum_J"] - Jr.sum()) < 1e-12 * abs(Jr.sum())
    # a non-finite objective is not counted and never the minimum
    u2 = u.copy()
    u2[:, :, 7] = np.nan
    _, J2, st2, rc2 = md.compute_states(integs, probs, x0, u2, want_x=False)
    assert rc2 == 1 and st2["count"] == B - 1 and not np.isfinite(J2[7]) and st2["argmin"] != 7
    # nlpObjective with a free initial state
    ctrls = md.replicate(lambda: ocs.ChebyshevControl(integs[0].t, nB, 1))
    c1 = ocs.ChebyshevControl(g1.t, nB, 1)
    V = 0.05 * rng.normal(size=(nB, B)) / np.arange(1, nB + 1)[:, None]
    V[0] += 0.4
    V = np.vstack([V, rng.uniform(0.8, 1.6, (1, B))])
    Jm, dm, x0m, stm, _ = md.nlp_objective(integs, probs, ctrls, x0.copy(), V, FreeInitStates=[2])
    Js, ds, x0s = ocs.nlp_objective(g1, p1, c1, x0.copy(), V, FreeInitStates=[2])
    assert np.array_equal(Jm, Js) and np.array_equal(dm, ds) and np.array_equal(x0m, x0s)
    assert stm["argmin"] == int(np.argmin(Js)) and abs(stm["sum_J"] - Js.sum()) < 1e-12 * abs(Js.sum())
    # fb_sweep: instances that converge next to instances that do not (lower bound -0.2, 30 sweeps)
    pb = md.replicate(lambda: ocs.LogisticProblem([3.0], P["c"], P["r"], [[-0.2, 6.0]]))
    ps = ocs.LogisticProblem([3.0], P["c"], P["r"], [[-0.2, 6.0]])
    ts2 = oracle.linspace(0, 4.5, 169)
    ig = md.replicate(lambda: ocs.RK4Integrator(ts2))
    x0b = rng.uniform(0.8, 1.6, (1, 128))
    opt = {"nERROR_PTS": 169, "nINTERP_PTS": 17, "nSWEEPS": 30}
    rm = md.fb_sweep(ig, pb, x0b, opt)
    rs = ocs.fb_sweep_batch(ps, x0b, ts2, opt)
    conv = rs["sweeps"] > 0
    assert np.array_equal(rm["sweeps"], rs["sweeps"]) and conv.any() and (~conv).any()
    for k in ("x", "lam", "u"):
        assert np.array_equal(rm[k][..., conv], rs[k][..., conv])
    assert np.array_equal(rm["J"][conv], rs["J"][conv])
    assert rm["stats"]["count"] == int(conv.sum())
    assert rm["stats"]["argmin"] == int(np.flatnonzero(conv)[np.argmin(rs["J"][conv])])
    # argument checks
    with pytest.raises(ocs.OcsError):
        ocs.MultiDevice([0, 0])
    with pytest.raises(ocs.OcsError):
        ocs.MultiDevice([99])
