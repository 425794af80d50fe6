"""GPU parity of the control bases, the single_shooting objective/gradient (A8) and
RK4InfiniteIntegrator (A5) against the CPU oracle, through the C-ABI.  Tolerance 1e-12 relative
(fp64, FMA contraction + device exp); dJdv sums ~2000 terms, checked against max(1,|ref|)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = {"c": 1.5, "m": 3.0, "r": 0.05}
BOUNDS = [[0.0, 1.0]]
RTOL = 1e-12


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


@pytest.fixture(scope="module")
def ocs():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    return g.load_package()


@pytest.mark.parametrize("kind,nB,nC", [("PWLinearControl", 21, 1), ("PWLinearControl", 5, 3), ("PWConstantControl", 10, 2),
                                        ("ChebyshevControl", 16, 1), ("ChebyshevControl", 3, 2)])
def test_compute_u_and_dJdv(ocs, oracle, kind, nB, nC):
    t = oracle.RK4Integrator(oracle.linspace(0, 10, 201)).t
    cg, co = getattr(ocs, kind)(t, nB, nC), getattr(oracle, kind)(t, nB, nC)
    rng = np.random.default_rng(1)
    batch = 70
    v = rng.normal(size=(nC * nB, batch))
    d = rng.normal(size=(nC, t.size, batch))
    u = cg.compute_u(v)
    dv = cg.compute_dJdv(d)
    for b in (0, 33, 69):
        assert relerr(u[:, :, b], co.compute_u(v[:, b])) < 1e-14
        assert relerr(dv[:, b], co.compute_dJdv(d[:, :, b])) < 1e-13
    # batch = 1 keeps the reference's shapes
    assert cg.compute_u(v[:, 0]).shape == (nC, t.size) and cg.compute_dJdv(d[:, :, 0]).shape == (nC * nB,)


def test_backprop_test_script(ocs, oracle):
    """tests/backprop_test.m:5-43, seeded: adjoint gradient vs forward differences (eps = 1e-4); the 21
    perturbed candidates + the base point run as ONE batch of 22."""
    N, nPts, eps = 500, 21, 1e-4
    tspan = oracle.linspace(0, 10, N + 1)
    prob, integ = ocs.TestOCProblem(P, BOUNDS), ocs.RK4Integrator(tspan)
    ctrl = ocs.PWLinearControl(integ.t, nPts, 1)
    v = 0.5 + 0.5 * np.random.default_rng(20260404).random(nPts)
    V = np.tile(v[:, None], (1, nPts + 1))
    V[np.arange(nPts), np.arange(1, nPts + 1)] += eps
    J, dJdv, _ = ocs.nlp_objective(integ, prob, ctrl, np.ones((1, nPts + 1)), V)
    fd = (J[1:] - J[0]) / eps
    assert np.max(np.abs(fd - dJdv[:, 0])) < 1e-3
    po, go = oracle.TestOCProblem(P, BOUNDS), oracle.RK4Integrator(tspan)
    Jo, do, _ = oracle.nlp_objective(go, po, oracle.PWLinearControl(go.t, nPts, 1), [1.0], v)
    assert abs(J[0] - Jo) < RTOL * abs(Jo) and relerr(dJdv[:, 0], do) < RTOL


@pytest.mark.parametrize("kind,nB", [("PWLinearControl", 101), ("PWConstantControl", 50), ("ChebyshevControl", 16)])
def test_nlp_objective_matches_oracle(ocs, oracle, kind, nB):
    nS, N, batch = 2, 300, 66
    m = [3.0, 2.5]
    tspan = oracle.linspace(0, 10, N + 1)
    pg, po = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS), oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    g, go = ocs.RK4Integrator(tspan), oracle.RK4Integrator(tspan)
    cg, co = getattr(ocs, kind)(g.t, nB, 1), getattr(oracle, kind)(go.t, nB, 1)
    rng = np.random.default_rng(3)
    if kind == "ChebyshevControl":  # SURVEY BL-4 candidates
        V = 0.05 * rng.normal(size=(nB, batch)) / np.arange(1, nB + 1)[:, None]
        V[0] += 0.4
    else:
        V = rng.uniform(0.05, 0.45, (nB, batch))
    x0 = rng.uniform(0.9, 2.0, (nS, batch))
    J, dJdv, _ = ocs.nlp_objective(g, pg, cg, x0, V)
    for b in (0, 1, 64, 65):
        Jo, do, _ = oracle.nlp_objective(go, po, co, x0[:, b], V[:, b])
        assert abs(J[b] - Jo) < RTOL * abs(Jo)
        assert relerr(dJdv[:, b], do) < RTOL


def test_free_initial_states(ocs, oracle):
    # single_shooting.m:144-149: v carries x0(FreeInitStates); dJdv gets lam(FreeInitStates,1)
    m = [3.0, 2.0, 2.5]
    tspan = oracle.linspace(0, 5, 101)
    pg, po = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS), oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    g, go = ocs.RK4Integrator(tspan), oracle.RK4Integrator(tspan)
    cg, co = ocs.PWLinearControl(g.t, 6, 1), oracle.PWLinearControl(go.t, 6, 1)
    rng = np.random.default_rng(4)
    batch = 5
    V = np.vstack([rng.uniform(0.1, 0.4, (6, batch)), rng.uniform(1.0, 2.0, (2, batch))])
    x0 = np.ones((3, batch))
    J, dJdv, x0n = ocs.nlp_objective(g, pg, cg, x0, V, FreeInitStates=[3, 1])
    for b in range(batch):
        Jo, do, x0o = oracle.nlp_objective(go, po, co, x0[:, b], V[:, b], FreeInitStates=[3, 1])
        assert abs(J[b] - Jo) < RTOL * abs(Jo) and relerr(dJdv[:, b], do) < RTOL
        assert np.array_equal(x0n[:, b], x0o) and x0n[2, b] == V[6, b] and x0n[0, b] == V[7, b]


def test_infinite_integrator(ocs, oracle):
    # RK4InfiniteIntegrator.m:12-30 with the tail under u* of the equilibrium (solve_test_problem.m:28-33)
    us = 0.72336878009798256
    tspan, tx = oracle.linspace(0, 10, 201), oracle.linspace(10, 20, 151)
    pg, po = ocs.TestOCProblem(P, BOUNDS), oracle.TestOCProblem(P, BOUNDS)
    gi, go = ocs.RK4InfiniteIntegrator(tspan, tx, [us]), oracle.RK4InfiniteIntegrator(tspan, tx, [us])
    assert np.array_equal(gi.t, go.t)
    rng = np.random.default_rng(6)
    batch = 67
    u = rng.uniform(0, 1, (1, 401, batch))
    x0 = rng.uniform(0.5, 2.5, (1, batch))
    x, J = gi.compute_states(pg, x0, u)
    lam, dJdu = gi.compute_adjoints(pg, u)
    for b in (0, 1, 65, 66):
        xo, Jo = go.compute_states(po, x0[:, b], u[:, :, b])
        lamo, do = go.compute_adjoints(po, u[:, :, b])
        assert relerr(x[:, :, b], xo) < RTOL and abs(J[b] - Jo) < RTOL * abs(Jo)
        assert relerr(lam[:, :, b], lamo) < RTOL and relerr(dJdu[:, :, b], do) < RTOL
    # as the Integrator of the shooting objective (solve_test_problem.m:38-39, commented out there)
    cg, co = ocs.PWLinearControl(gi.t, 11, 1), oracle.PWLinearControl(go.t, 11, 1)
    V = rng.uniform(0.2, 0.9, (11, 3))
    J, dJdv, _ = ocs.nlp_objective(gi, pg, cg, np.ones((1, 3)), V)
    for b in range(3):
        Jo, do, _ = oracle.nlp_objective(go, po, co, [1.0], V[:, b])
        assert abs(J[b] - Jo) < RTOL * abs(Jo) and relerr(dJdv[:, b], do) < RTOL


@pytest.mark.parametrize("nS,N,N2,batch", [(1, 200, 152, 128), (2, 64, 64, 96), (4, 48, 40, 32), (2, 64, 60, 96), (2, 64, 64, 70)])
def test_infinite_integrator_tail_leg_mappings(ocs, oracle, nS, N, N2, batch):
    """RK4InfiniteIntegrator.m:12-30 on batches where the tail leg (constant control uStar) runs on the wave-specialised kernels
    with the control as samples (whole blocks of 8 steps, whole tiles of 64/nS trajectories) and where it stays on the lane kernels
    (N2 = 60 is split, batch 70 is ragged): x, J = J1 + J2, lam (with lamT = lam2(:,1)) and dJdu against the oracle."""
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    us = 0.4
    tspan, tx = oracle.linspace(0, 2.0, N + 1), oracle.linspace(2.0, 4.0, N2 + 1)
    pg, po = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS), oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    gi, go = ocs.RK4InfiniteIntegrator(tspan, tx, [us]), oracle.RK4InfiniteIntegrator(tspan, tx, [us])
    rng = np.random.default_rng(N + N2 + batch)
    u = rng.uniform(0.05, 0.45, (1, 2 * N + 1, batch))
    x0 = rng.uniform(0.8, 2.0, (nS, batch))
    for rep in range(2):        # (the second call re-uses the sampled control kept with the handle)
        x, J = gi.compute_states(pg, x0, u)
        lam, dJdu = gi.compute_adjoints(pg, u)
        for b in sorted({0, batch // 2, batch - 1}):
            xo, Jo = go.compute_states(po, x0[:, b], u[:, :, b])
            lamo, do = go.compute_adjoints(po, u[:, :, b])
            assert relerr(x[:, :, b], xo) < RTOL and abs(J[b] - Jo) < RTOL * abs(Jo)
            assert relerr(lam[:, :, b], lamo) < RTOL and relerr(dJdu[:, :, b], do) < RTOL


def test_device_objective_bl4_shape(ocs, oracle):
    """BASELINE config 4 shape (Chebyshev-16 objective+gradient) on device buffers, reduced batch."""
    import torch
    N, nB, batch = 1000, 16, 512
    tspan = oracle.linspace(0, 10, N + 1)
    pg, po = ocs.TestOCProblem(P, BOUNDS), oracle.TestOCProblem(P, BOUNDS)
    g, go = ocs.RK4Integrator(tspan), oracle.RK4Integrator(tspan)
    cg, co = ocs.ChebyshevControl(g.t, nB, 1), oracle.ChebyshevControl(go.t, nB, 1)
    rng = np.random.default_rng(20260403)
    V = 0.05 * rng.normal(size=(nB, batch)) / np.arange(1, nB + 1)[:, None]
    V[0] += 0.5
    dev = torch.device("cuda:0")
    vd = torch.tensor(V, device=dev)                   # [nV][B]
    x0d = torch.ones((1, batch), dtype=torch.float64, device=dev)
    Jd, gd = ocs.nlp_objective_dev(g, pg, cg, x0d, vd)
    torch.cuda.synchronize()
    Jh, gh = Jd.cpu().numpy(), gd.cpu().numpy()
    for b in (0, 255, 511):
        Jo, do, _ = oracle.nlp_objective(go, po, co, [1.0], V[:, b])
        assert abs(Jh[b] - Jo) < RTOL * abs(Jo) and relerr(gh[:, b], do) < RTOL


def test_single_shooting_end_to_end(ocs, oracle):
    """tests/solve_test_problem.m:5-39 (BL-1 plumbing): T=10, 500 steps, 101 control points, u0 = u*.
    fmincon is not available; the SLSQP stand-in must reach the same optimum as the same optimiser
    driven by the CPU oracle, and the optimal control must sit on the turnpike u* in mid-horizon."""
    from scipy.optimize import minimize
    us = 0.72336878009798256
    tspan = oracle.linspace(0, 10, 501)
    prob = ocs.TestOCProblem(P, BOUNDS)
    soln = ocs.single_shooting(prob, [1.0], tspan, 101, u0=us)
    assert set(("J", "u", "x", "lam")) <= set(soln)
    po, go = oracle.TestOCProblem(P, BOUNDS), oracle.RK4Integrator(tspan)
    co = oracle.PWLinearControl(go.t, 101, 1)
    res = minimize(lambda v: oracle.nlp_objective(go, po, co, [1.0], v)[:2], np.full(101, us), jac=True,
                   method="SLSQP", bounds=[(0.0, 1.0)] * 101, options={"ftol": 3e-7, "maxiter": 400})
    assert abs(soln["J"] - res.fun) < 1e-6 * abs(res.fun)
    tq = np.array([4.0, 5.0, 6.0])
    assert np.max(np.abs(soln["u"](tq) - us)) < 2e-2
    assert np.max(np.abs(soln["x"](tq) - 2.7355691886341361)) < 2e-2
    assert soln["x"](np.array([0.0]))[0, 0] == 1.0 and soln["lam"](tq).shape == (1, 3)


def test_single_shooting_constraint_hooks_and_tolx(ocs, oracle):
    """single_shooting.m:99-115: compute_lincon / compute_nonlcon of the control object reach the optimiser, TolX ends
    the iteration.  ChebyshevControl has no bounds of its own (its compute_lincon is empty in the reference,
    ChebyshevControl.m:51-53); with the linear constraints at the grid points the optimal control of the test problem --
    which without them sits on the turnpike u* = 0.72 above the upper bound 0.5 -- stays inside ControlBounds."""
    tspan = oracle.linspace(0, 10, 201)
    prob = ocs.TestOCProblem(P, [[0.0, 0.5]])
    integ = ocs.RK4Integrator(tspan)

    free_c = ocs.ChebyshevControl(integ.t, 12, 1)
    s_con = ocs.single_shooting(prob, [2.9], tspan, 12, u0=0.4, Control=free_c, Integrator=integ, TolX=0)
    A, b = free_c.compute_lincon(prob.ControlBounds)
    assert A.shape == (2 * integ.t.size, 12) and s_con["_constraints"] == 1
    assert np.all(A @ s_con["_v"] <= b + 1e-7)
    tq = integ.t
    ucon = s_con["u"](tq)
    assert ucon.max() <= 0.5 + 1e-6 and ucon.min() >= -1e-6
    s_unc = ocs.single_shooting(prob, [2.9], tspan, 12, u0=0.4, Control=_NoLincon(ocs, integ.t, 12, 1), Integrator=integ, TolX=0)
    assert s_unc["_constraints"] == 0 and s_unc["u"](tq).max() > 0.5 + 1e-3 and s_unc["J"] < s_con["J"]
    # a nonlinear constraint with its gradient (GradConstr 'on', :101): the mean square of the coefficients above the first
    nl = _NonlconControl(ocs, integ.t, 12, 1, 1e-3)
    s_nl = ocs.single_shooting(prob, [2.9], tspan, 12, u0=0.4, Control=nl, Integrator=integ, TolX=0)
    assert s_nl["_constraints"] == 1 and np.sum(s_nl["_v"][1:] ** 2) <= 1e-3 * (1 + 1e-6) < np.sum(s_unc["_v"][1:] ** 2)
    # TolX: a loose step tolerance ends the iteration early (fewer evaluations, objective within first order of the step)
    s_tx = ocs.single_shooting(prob, [2.9], tspan, 12, u0=0.4, Control=_NoLincon(ocs, integ.t, 12, 1), Integrator=integ, TolX=1e-2)
    assert s_tx["_stopped_on_TolX"] and s_tx["_nfev"] < s_unc["_nfev"] and s_tx["J"] >= s_unc["J"] - 1e-9
    # the batched projected-gradient driver takes box bounds only and says so
    with pytest.warns(RuntimeWarning):
        ocs.single_shooting_batch(prob, np.full((1, 4), 2.9), tspan, 12, Control=free_c, Integrator=integ, u0=0.5, MaxIter=3)
    with pytest.raises(ValueError):
        ocs.single_shooting_batch(prob, np.full((1, 4), 2.9), tspan, 12, Control=free_c, Integrator=integ, constraints="error")


def _NoLincon(ocs, *a):
    class C(ocs.ChebyshevControl):
        def __getattribute__(self, name):   # the reference's ChebyshevControl as shipped: no usable constraint hook
            if name == "compute_lincon":
                raise AttributeError(name)
            return super().__getattribute__(name)
    return C(*a)


def _NonlconControl(ocs, t, nB, nC, cap):
    class C(ocs.ChebyshevControl):
        def __getattribute__(self, name):
            if name == "compute_lincon":
                raise AttributeError(name)
            return super().__getattribute__(name)

        def compute_nonlcon(self, v):       # [c, ceq, gradc, gradceq] = compute_nonlcon(obj, v)   Control.m:12
            g = 2 * v
            g[0] = 0.0
            return np.array([np.sum(v[1:] ** 2) - cap]), np.zeros(0), g[:, None], np.zeros((v.size, 0))
    return C(t, nB, nC)


def test_compute_equilibrium_and_solve_test_problem_script(ocs, oracle):
    """tests/solve_test_problem.m:21-39 end to end: equilibrium (analytic KAT 1) -> RK4InfiniteIntegrator with
    uStar -> single_shooting with that integrator (the line the reference leaves commented out, :38-39)."""
    prob = ocs.TestOCProblem(P, BOUNDS)
    lb, ub = [0.0, -np.inf, 0.0], [np.inf, np.inf, 1.0]                          # :25-26
    xs, ls, us, resnorm, _, flag = ocs.compute_equilibrium(prob, 2.7, 2.2, 0.7, lb, ub, P["r"])  # :27-29
    assert flag > 0 and resnorm < 1e-20
    assert abs(xs[0] - 2.7355691886341361) < 1e-9 and abs(ls[0] - 2.1701063402939477) < 1e-9
    assert abs(us[0] - 0.72336878009798256) < 1e-9
    tspan, tx = oracle.linspace(0, 10, 201), oracle.linspace(10, 20, 201)
    integ = ocs.RK4InfiniteIntegrator(tspan, tx, us)                              # :33
    soln = ocs.single_shooting(prob, [1.0], tspan, 41, u0=us, Integrator=integ)   # :37-39
    # with the tail leg the optimal control stays on the turnpike up to the end of the horizon
    assert abs(soln["u"](np.array([9.5]))[0, 0] - us[0]) < 2e-2
    plain = ocs.single_shooting(prob, [1.0], tspan, 41, u0=us)
    assert plain["u"](np.array([10.0]))[0, 0] < 0.2  # finite horizon: harvest drops at the end (lam(T) = 0)


def _testoc_equilibrium(c, m, r):
    """interior root of the optimality system of TestOCProblem (analytic): with lam = 2 c u (dFdu' [lam;1] = 0 at
    t = 0), x' = 0: u = x (m - x), costate: (r - m + 2 x) lam = 2 x  ->  one equation in x, solved by bisection on the
    branch x in (m/2, m) (the root compute_equilibrium converges to from the reference's guess)."""
    # x != 0: (r - m + 2 x) c (m - x) = 1, a downward parabola minus 1; the feasible root (u <= 1) is the upper one,
    # between the vertex and m
    f = lambda x: (r - m + 2 * x) * c * (m - x) - 1.0
    lo, hi = (3 * m - r) / 4, m
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if f(mid) > 0 else (lo, mid)
    x = 0.5 * (lo + hi)
    u = x * (m - x)
    return x, 2 * c * u, u


def test_compute_equilibrium_undefined_values_and_fixed_variables(ocs):
    """compute_equilibrium.m:23-27: lsqnonlin raises an error on undefined values and never reports such a point as a
    solution, and it accepts lb == ub.  Here: a guess with a NaN (or one that overflows the residual) gives exitflag -1
    for that instance only; a single instance raises; a variable fixed by lb == ub stays put while the others solve the
    remaining equations in the least-squares sense."""
    prob = ocs.TestOCProblem(P, BOUNDS)
    lb, ub = [0.0, -np.inf, 0.0], [np.inf, np.inf, 1.0]
    xG = np.array([[2.7, np.nan, 2.7, 1e200]])
    lG = np.array([[2.2, 2.2, np.nan, 2.2]])
    uG = np.full((1, 4), 0.7)
    xs, ls, us, resnorm, res, flag = ocs.compute_equilibrium(prob, xG, lG, uG, lb, ub, P["r"])
    assert flag.tolist() == [1, -1, -1, -1] and resnorm[0] < 1e-24 and not np.isfinite(resnorm[1:]).any()
    assert abs(xs[0, 0] - 2.7355691886341361) < 1e-12
    with pytest.raises(ocs.OcsError):
        ocs.compute_equilibrium(prob, np.nan, 2.2, 0.7, lb, ub, P["r"])
    # u fixed at 0.5 by lb == ub: same point as with the active upper bound 0.5
    xa, la, ua, rna, _, fa = ocs.compute_equilibrium(prob, 2.7, 2.2, 0.4, lb, [np.inf, np.inf, 0.5], P["r"])
    xf, lf, uf, rnf, _, ff = ocs.compute_equilibrium(prob, 2.7, 2.2, 0.5, [0.0, -np.inf, 0.5], [np.inf, np.inf, 0.5], P["r"])
    assert ff == 1 and uf[0] == 0.5 and np.isfinite(rnf)
    assert abs(xf[0] - xa[0]) < 1e-9 and abs(lf[0] - la[0]) < 1e-9 and abs(rnf - rna) < 1e-12
    # x fixed at its equilibrium value: the other two unknowns reach the root
    xe = 2.7355691886341361
    x1, l1, u1, rn1, _, f1 = ocs.compute_equilibrium(prob, xe, 2.0, 0.6, [xe, -np.inf, 0.0], [xe, np.inf, 1.0], P["r"])
    assert f1 == 1 and x1[0] == xe and abs(l1[0] - 2.1701063402939477) < 1e-9 and abs(u1[0] - 0.72336878009798256) < 1e-9


def test_compute_equilibrium_batched_on_device(ocs, oracle):
    """compute_equilibrium.m:10-27 as a batch: 4096 instances with per-instance c, every one against the analytic root
    and against the oracle-side residual; the device entry point; the reference's known answer from its guess."""
    import torch
    B = 4096
    rng = np.random.default_rng(21)
    cs = rng.uniform(1.2, 2.0, B)
    prob = ocs.TestOCProblem(P, BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    lb, ub = [0.0, -np.inf, 0.0], [np.inf, np.inf, 1.0]                          # solve_test_problem.m:25-26
    xG, lG, uG = np.full((1, B), 2.7), np.full((1, B), 2.2), np.full((1, B), 0.7)
    xs, ls, us, resnorm, res, flag = ocs.compute_equilibrium(prob, xG, lG, uG, lb, ub, P["r"])
    assert np.all(flag == 1) and np.max(resnorm) < 1e-24
    ref = np.array([_testoc_equilibrium(c, P["m"], P["r"]) for c in cs]).T
    assert np.max(np.abs(xs[0] - ref[0])) < 1e-11 and np.max(np.abs(ls[0] - ref[1])) < 1e-11
    assert np.max(np.abs(us[0] - ref[2])) < 1e-11
    # oracle-side residual of the reference's system (compute_equilibrium.m:13-21) at the device's answer
    for b in (0, 1, 777, B - 1):
        po = oracle.TestOCProblem({"c": cs[b], "m": P["m"], "r": P["r"]}, BOUNDS)
        y = np.array([xs[0, b], 0.0])
        F = po.F([0.0], y, [us[0, b]]).ravel()
        g = po.dFdx_times_vec([0.0], y, [us[0, b]], np.array([ls[0, b], 1.0])).ravel()
        gu = po.dFdu_times_vec([0.0], y, [us[0, b]], np.array([ls[0, b], 1.0])).ravel()
        assert abs(F[0]) < 1e-12 and abs(P["r"] * ls[0, b] - g[0]) < 1e-12 and abs(gu[0]) < 1e-12
    # SURVEY KAT 1 (c = 1.5) through the single-instance call of the reference's signature
    p1 = ocs.TestOCProblem(P, BOUNDS)
    x1, l1, u1, rn, _, fl = ocs.compute_equilibrium(p1, 2.7, 2.2, 0.7, lb, ub, P["r"])
    assert fl == 1 and rn < 1e-24
    assert abs(x1[0] - 2.7355691886341361) < 1e-12 and abs(l1[0] - 2.1701063402939477) < 1e-12
    assert abs(u1[0] - 0.72336878009798256) < 1e-12
    # device entry point, asynchronous, same answers
    dev = torch.device("cuda:0")
    yG = torch.tensor(np.vstack([xG, lG, uG]), device=dev)
    lbd = torch.tensor([0.0, -1e300, 0.0], dtype=torch.float64, device=dev)
    ubd = torch.tensor([1e300, 1e300, 1.0], dtype=torch.float64, device=dev)
    yd, rnd, resd, fld = ocs.compute_equilibrium_dev(prob, yG, lbd, ubd, P["r"])
    torch.cuda.synchronize()
    assert np.max(np.abs(yd.cpu().numpy() - np.vstack([xs, ls, us]))) < 1e-13 and bool((fld == 1).all())
    # an active bound: u <= 0.5 -> the constrained least-squares point sits on the bound, exitflag still 1
    xb, lbm, ub_, rnb, _, flb = ocs.compute_equilibrium(p1, 2.7, 2.2, 0.4, lb, [np.inf, np.inf, 0.5], P["r"])
    assert flb == 1 and abs(ub_[0] - 0.5) < 1e-15 and rnb > 1e-6
    # a two-state registry problem (five unknowns): its optimality system has NO steady state (the two logistic rows
    # cannot both balance one harvest rate with positive costates); lsqnonlin then returns the least-squares point, and
    # so must this: compare with scipy's trust-region-reflective solver on the oracle's plugin methods
    from scipy.optimize import least_squares
    m2 = [3.0, 2.5]
    p2, po2 = ocs.LogisticProblem(m2, P["c"], P["r"], BOUNDS), oracle.LogisticProblem(m2, P["c"], P["r"], BOUNDS)
    lb2, ub2 = [0, 0, -np.inf, -np.inf, 0], [np.inf] * 5

    def system(y):                                                         # compute_equilibrium.m:13-21
        ya, va = np.array([y[0], y[1], 0.0]), np.array([y[2], y[3], 1.0])
        return np.concatenate([po2.F([0.0], ya, y[4:]).ravel()[:2],
                               P["r"] * y[2:4] - po2.dFdx_times_vec([0.0], ya, y[4:], va).ravel()[:2],
                               po2.dFdu_times_vec([0.0], ya, y[4:], va).ravel()])
    ref2 = least_squares(system, [2.6, 2.1, 1.2, 1.2, 0.6], bounds=(lb2, ub2), xtol=1e-15, ftol=1e-15, gtol=1e-15)
    x2, l2, u2, rn2, res2, fl2 = ocs.compute_equilibrium(p2, [2.6, 2.1], [1.2, 1.2], [0.6], lb2, ub2, P["r"])
    assert fl2 == 1 and abs(rn2 - 2 * ref2.cost) < 1e-10 and rn2 > 0.1
    assert np.max(np.abs(np.concatenate([x2, l2, u2]) - ref2.x)) < 1e-6


def test_single_shooting_batch_on_device(ocs, oracle):
    """a batch of independent shooting NLPs (different x0 and c) solved together by the
    batched projected-gradient driver; every instance must reach the optimum the oracle-driven SLSQP finds."""
    from scipy.optimize import minimize
    rng = np.random.default_rng(12)
    B, N, nPts = 24, 200, 21
    tspan = oracle.linspace(0, 10, N + 1)
    x0 = rng.uniform(0.6, 2.4, (1, B))
    cs = rng.uniform(1.0, 2.0, B)
    prob = ocs.TestOCProblem(P, BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    r = ocs.single_shooting_batch(prob, x0, tspan, nPts, u0=0.7, TolFun=1e-6, MaxIter=400)
    J = r["J"].cpu().numpy()
    assert bool(r["converged"].all())
    go = oracle.RK4Integrator(tspan)
    co = oracle.PWLinearControl(go.t, nPts, 1)
    for b in (0, 7, 23):
        po = oracle.TestOCProblem({"c": cs[b], "m": P["m"], "r": P["r"]}, BOUNDS)
        res = minimize(lambda v: oracle.nlp_objective(go, po, co, x0[:, b], v)[:2], np.full(nPts, 0.7), jac=True,
                       method="SLSQP", bounds=[(0.0, 1.0)] * nPts, options={"ftol": 1e-12, "maxiter": 500})
        assert abs(J[b] - res.fun) < 2e-6 * abs(res.fun)
        s = r["soln_of"](b)
        assert np.max(np.abs(s["v"] - res.x)) < 5e-3


def test_single_shooting_batch_free_initial_states(ocs, oracle):
    """The batched driver with FreeInitStates inside FreeStateBounds (single_shooting.m:81-94, :144-149): each
    instance against SLSQP on the oracle's nlpObjective with the same bounds.  (The free state ends on its lower
    bound, the control coefficients partly inside their bounds.)"""
    from scipy.optimize import minimize
    rng = np.random.default_rng(3)
    B, N, nPts = 10, 100, 11
    tspan = oracle.linspace(0, 5, N + 1)
    x0 = rng.uniform(0.95, 1.5, (1, B))
    cs = rng.uniform(1.0, 2.0, B)
    prob = ocs.TestOCProblem(P, BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    fsb = [[0.9, 2.0]]
    r = ocs.single_shooting_batch(prob, x0, tspan, nPts, u0=0.3, TolFun=1e-7, TolX=1e-12, MaxIter=600,
                                  FreeInitStates=[1], FreeStateBounds=fsb)
    assert bool(r["converged"].all()) and int(r["iterations"].max()) < 200
    J, V, x0n = r["J"].cpu().numpy(), r["v"].cpu().numpy(), r["x0"].cpu().numpy()
    assert np.array_equal(x0n[0], V[nPts])                                           # :121-124
    assert np.all(V[nPts] >= 0.9) and np.all(V[nPts] <= 2.0) and np.all(V[:nPts] >= 0) and np.all(V[:nPts] <= 1)
    go = oracle.RK4Integrator(tspan)
    co = oracle.PWLinearControl(go.t, nPts, 1)
    for b in (0, 4, 9):
        po = oracle.TestOCProblem({"c": cs[b], "m": P["m"], "r": P["r"]}, BOUNDS)
        f = lambda v: oracle.nlp_objective(go, po, co, x0[:, b], v, FreeInitStates=[1])[:2]
        res = minimize(f, np.concatenate([np.full(nPts, 0.3), [x0[0, b]]]), jac=True, method="SLSQP",
                       bounds=[(0.0, 1.0)] * nPts + [tuple(fsb[0])], options={"ftol": 1e-13, "maxiter": 500})
        assert abs(J[b] - res.fun) < 2e-6 * abs(res.fun)
        assert np.max(np.abs(V[:, b] - res.x)) < 5e-3


def test_dense_basis_kernels_large_batch(ocs, oracle):
    # Chebyshev is a dense basis: at batch >= 16384 the register-resident kernels take over
    import torch
    t = oracle.RK4Integrator(oracle.linspace(0, 10, 101)).t
    nB, nC, batch = 12, 2, 16384
    cg, co = ocs.ChebyshevControl(t, nB, nC), oracle.ChebyshevControl(t, nB, nC)
    rng = np.random.default_rng(5)
    v = rng.normal(size=(nB, nC, batch))
    d = rng.normal(size=(t.size, nC, batch))
    dev = torch.device("cuda:0")
    u = cg.compute_u_dev(torch.tensor(v, device=dev)).cpu().numpy()
    dv = cg.compute_dJdv_dev(torch.tensor(d, device=dev)).cpu().numpy()
    for b in (0, 8191, 16383):
        vb = v[:, :, b].reshape(-1)                       # control index fastest within a basis function
        assert relerr(u[:, :, b].T, co.compute_u(vb)) < 1e-13
        assert relerr(dv[:, :, b].reshape(-1), co.compute_dJdv(d[:, :, b].T)) < 1e-12


@pytest.mark.parametrize("nS,nB,N,batch", [(1, 16, 1000, 70), (1, 5, 7, 3), (2, 20, 50, 130), (4, 32, 64, 64),
                                           (3, 1, 9, 65), (2, 12, 48, 130), (1, 9, 16, 200), (2, 16, 8, 1),
                                           (1, 3, 24, 64)])
def test_fused_control_objective_gradient(ocs, oracle, nS, nB, N, batch):
    """single_shooting.m:137-150 with ChebyshevControl.m:35-43 applied inside the RK4 kernels (u and dJdu never in
    memory) against the oracle's unfused composition and against this library's own unfused path; free initial
    states (:144-149) and a per-trajectory parameter included.  Shapes with N % 8 == 0, nB <= 16, nS <= 2 run the
    two-role kernels (integrator wave + basis wave, k_forward_fc2 / k_backward_fc2: one, two, three and 125 blocks,
    odd and even block counts, partial last wave), the others the one-wave lane kernels; (1, 3, 24, 64) is forced
    onto the lane kernels below ("lane"), its shape would otherwise take the wave-specialised path."""
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    T = 10.0 if N >= 50 else 1.0
    tspan = oracle.linspace(0, T, N + 1)
    g, go = ocs.RK4Integrator(tspan), oracle.RK4Integrator(tspan)
    cg, co = ocs.ChebyshevControl(g.t, nB, 1), oracle.ChebyshevControl(go.t, nB, 1)
    rng = np.random.default_rng(nB * 100 + N)
    free = [nS, 1] if nS > 1 else [1]
    V = 0.05 * rng.normal(size=(nB, batch)) / np.arange(1, nB + 1)[:, None]
    V[0] += 0.4
    V = np.vstack([V, rng.uniform(0.8, 1.6, (len(free), batch))])
    cs = rng.uniform(1.0, 2.0, batch)
    x0 = rng.uniform(0.8, 1.5, (nS, batch))
    pg = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    pg.set_batch_params([0], cs[None, :])   # LogisticK parameter block [c r m_1..m_nS]: c per trajectory
    out = {}
    for mode in ("lane", "off"):
        cg.set_fusion(mode)
        out[mode] = ocs.nlp_objective(g, pg, cg, x0.copy(), V, FreeInitStates=free)
    Jf, df, x0f = out["lane"]
    Ju, du, x0u = out["off"]
    assert relerr(Jf, Ju) < 1e-13 and relerr(df, du) < RTOL and np.array_equal(x0f, x0u)
    for b in sorted({0, batch // 2, batch - 1}):
        po = oracle.LogisticProblem(m, cs[b], P["r"], BOUNDS)
        Jo, do, x0o = oracle.nlp_objective(go, po, co, x0[:, b], V[:, b], FreeInitStates=free)
        assert abs(Jf[b] - Jo) < RTOL * max(1.0, abs(Jo)) and relerr(df[:, b], do) < RTOL
        assert np.array_equal(x0f[:, b], x0o)
    # the fused pass leaves no forward pass behind that compute_adjoints could pair with a control array
    cg.set_fusion("on")
    ocs.nlp_objective(g, pg, cg, x0.copy(), V, FreeInitStates=free)
    with pytest.raises(Exception):
        g.compute_adjoints(pg, np.zeros((1, 2 * N + 1, batch)))
    cg.set_fusion("auto")


@pytest.mark.parametrize("nB,N,batch,grid", [(16, 1000, 128, "lin"), (5, 8, 64, "lin"), (32, 64, 192, "rand"), (20, 200, 64, "lin"),
                                             (1, 72, 64, "lin"), (13, 136, 320, "rand"), (4, 1000, 64, "np")])
def test_fused_control_wave_kernels(ocs, oracle, nB, N, batch, grid):
    """single_shooting.m:137-150 + ChebyshevControl.m:35-43 on the wave-specialised state pass and the adjoint scan with
    u = v B and dJdv = dJdu B' on the matrix cores (csrc/ocs_fused_wave_kernels.hip): against the oracle's unfused
    composition, the lane-per-trajectory fused kernels and the unfused path.  Shapes: one to three superblocks of the scan
    with dead chunks (N = 8, 72, 136), 1..32 basis functions (one to eight k-steps, two gradient tiles), several
    workgroups, a non-uniform grid, a numpy linspace (not bitwise uniform), a free initial state, a per-trajectory
    parameter."""
    rng = np.random.default_rng(nB * 1000 + N)
    T = 10.0 if N >= 50 else 1.0
    tspan = {"lin": oracle.linspace(0, T, N + 1), "np": np.linspace(0, T, N + 1),
             "rand": np.concatenate([[0.0], np.sort(rng.uniform(0, T, N - 1)), [T]])}[grid]
    g, go = ocs.RK4Integrator(tspan), oracle.RK4Integrator(tspan)
    cg, co = ocs.ChebyshevControl(g.t, nB, 1), oracle.ChebyshevControl(go.t, nB, 1)
    V = 0.05 * rng.normal(size=(nB, batch)) / np.arange(1, nB + 1)[:, None]
    V[0] += 0.4
    V = np.vstack([V, rng.uniform(0.8, 1.6, (1, batch))])
    cs = rng.uniform(1.0, 2.0, batch)
    x0 = rng.uniform(0.8, 1.5, (1, batch))
    pg = ocs.LogisticProblem([3.0], P["c"], P["r"], BOUNDS)
    pg.set_batch_params([0], cs[None, :])
    out = {}
    for mode in ("on", "lane", "off"):
        cg.set_fusion(mode)
        out[mode] = ocs.nlp_objective(g, pg, cg, x0.copy(), V, FreeInitStates=[1])
    Jw, dw, x0w = out["on"]
    for other in ("lane", "off"):
        Jl, dl, x0l = out[other]
        assert relerr(Jw, Jl) < 1e-13 and relerr(dw, dl) < RTOL and np.array_equal(x0w, x0l), other
    for b in sorted({0, 15, 16, batch // 2 + 1, batch - 1}):
        po = oracle.LogisticProblem([3.0], cs[b], P["r"], BOUNDS)
        Jo, do, _ = oracle.nlp_objective(go, po, co, x0[:, b], V[:, b], FreeInitStates=[1])
        assert abs(Jw[b] - Jo) < RTOL * max(1.0, abs(Jo)) and relerr(dw[:, b], do) < RTOL
    # without free initial states (no lam0 output)
    cg.set_fusion("on")
    J2, d2, _ = ocs.nlp_objective(g, pg, cg, x0.copy(), V[:nB])
    cg.set_fusion("lane")
    J3, d3, _ = ocs.nlp_objective(g, pg, cg, x0.copy(), V[:nB])
    assert relerr(J2, J3) < 1e-13 and relerr(d2, d3) < RTOL
    cg.set_fusion("auto")


@pytest.mark.parametrize("kind,nS,nB,N,batch", [("lin", 1, 101, 500, 70), ("lin", 1, 2, 7, 3), ("lin", 2, 11, 50, 130),
                                                ("lin", 4, 33, 64, 64), ("lin", 1, 51, 50, 5), ("lin", 3, 300, 40, 9),
                                                ("const", 1, 50, 500, 70), ("const", 2, 1, 9, 65), ("const", 4, 7, 50, 33),
                                                ("const", 1, 100, 50, 4), ("lin", 1, 6, 1, 2)])
def test_fused_banded_control_objective_gradient(ocs, oracle, kind, nS, nB, N, batch):
    """single_shooting.m:137-150 with the reference's default bases (PWLinearControl.m:31-62, PWConstantControl.m:30-50)
    applied inside the RK4 kernels: two live coefficient rows per trajectory, u and dJdu never in memory.  Against
    the oracle's unfused composition and this library's unfused path, including a non-uniform grid, more control
    points than grid samples (falls back to the unfused kernels), free initial states, a per-trajectory parameter."""
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    T = 10.0 if N >= 50 else 1.0
    rng = np.random.default_rng(nB * 100 + N)
    tspan = oracle.linspace(0, T, N + 1) if N != 64 else np.concatenate([[0.0], np.sort(rng.uniform(0, T, N - 1)), [T]])
    g, go = ocs.RK4Integrator(tspan), oracle.RK4Integrator(tspan)
    Cg, Co = (ocs.PWLinearControl, oracle.PWLinearControl) if kind == "lin" else (ocs.PWConstantControl, oracle.PWConstantControl)
    cg, co = Cg(g.t, nB, 1), Co(go.t, nB, 1)
    free = [nS, 1] if nS > 1 else [1]
    V = np.vstack([rng.uniform(0.05, 0.45, (nB, batch)), rng.uniform(0.8, 1.6, (len(free), batch))])
    cs = rng.uniform(1.0, 2.0, batch)
    x0 = rng.uniform(0.8, 1.5, (nS, batch))
    pg = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    pg.set_batch_params([0], cs[None, :])
    out = {}
    for mode in ("lane", "off"):
        cg.set_fusion(mode)
        out[mode] = ocs.nlp_objective(g, pg, cg, x0.copy(), V, FreeInitStates=free)
    Jf, df, x0f = out["lane"]
    Ju, du, x0u = out["off"]
    assert relerr(Jf, Ju) < 1e-13 and relerr(df, du) < RTOL and np.array_equal(x0f, x0u)
    for b in sorted({0, batch // 2, batch - 1}):
        po = oracle.LogisticProblem(m, cs[b], P["r"], BOUNDS)
        Jo, do, x0o = oracle.nlp_objective(go, po, co, x0[:, b], V[:, b], FreeInitStates=free)
        assert abs(Jf[b] - Jo) < RTOL * max(1.0, abs(Jo)) and relerr(df[:, b], do) < RTOL
    cg.set_fusion("auto")


@pytest.mark.parametrize("basis,nB,mode,batch", [("cheb", 16, "on", 64), ("cheb", 16, "lane", 64), ("cheb", 11, "lane", 70),
                                                 ("cheb", 16, "off", 64), ("pwl", 21, "auto", 64), ("pwc", 10, "auto", 64)])
def test_shooting_gradient_is_the_gradient_of_the_objective_without_the_oracle(ocs, basis, nB, mode, batch):
    """single_shooting.m:137-150 held against itself on the GPU (no oracle): dJdv of nlpObjective is the exact gradient of
    its J, so dJdv . d equals the directional derivative of J along d, taken by a fourth-order central difference whose four
    shifted coefficient vectors ride in the batch.  Every fusion mode of the Chebyshev basis (wave-specialised + scan with
    the products on the matrix cores, lane kernels in one wave / two roles, unfused), piecewise-linear and
    piecewise-constant bases; free initial states included."""
    rng = np.random.default_rng(nB * 7 + batch)
    N, eps, T = 64, 1e-3, 4.0
    g = ocs.RK4Integrator(np.linspace(0.0, T, N + 1))
    ctrl = {"cheb": ocs.ChebyshevControl, "pwl": ocs.PWLinearControl, "pwc": ocs.PWConstantControl}[basis](g.t, nB, 1)
    if basis == "cheb":
        ctrl.set_fusion(mode)
    prob = ocs.LogisticProblem([3.0], P["c"], P["r"], BOUNDS)
    nV = ctrl.nBasis + 1                                  # + one free initial state
    v = np.vstack([0.3 + 0.05 * rng.normal(size=(ctrl.nBasis, batch)) / (1 + np.arange(ctrl.nBasis))[:, None] ** (basis == "cheb"),
                   rng.uniform(0.9, 1.4, (1, batch))])
    x0 = rng.uniform(0.9, 1.4, (1, batch))
    d = rng.normal(size=(nV, batch))
    J0, dJdv, _ = ocs.nlp_objective(g, prob, ctrl, x0.copy(), v, FreeInitStates=[1])
    vb = np.concatenate([v + s * eps * d for s in (1.0, -1.0, 2.0, -2.0)], axis=1)
    Jb, _, _ = ocs.nlp_objective(g, prob, ctrl, np.tile(x0, (1, 4)), vb, FreeInitStates=[1])
    Jp, Jm, Jpp, Jmm = (Jb[k * batch:(k + 1) * batch] for k in range(4))
    fd = (8.0 * (Jp - Jm) - (Jpp - Jmm)) / (12.0 * eps)
    an = np.sum(dJdv * d, axis=0)
    scale = np.maximum(np.abs(an), np.abs(J0) * 1e-3 + 1e-6)
    print(basis, nB, mode, "max relative difference", float(np.max(np.abs(an - fd) / scale)))
    assert np.max(np.abs(an - fd) / scale) < 5e-9
    if basis == "cheb":
        ctrl.set_fusion("auto")


def test_randomised_objective_gradient_stress(ocs, oracle):
    """tests/stress_nlp.py, 40 random cases of [J, dJdv] = nlpObjective(v) (single_shooting.m:137-150): the three control bases,
    basis sizes, nS 1..4, step counts and batches around the kernels' block and tile sizes, uniform and non-uniform grids, free
    initial states, every fusion mode -- sampled candidates against the oracle at 1e-12 (the long run: python tests/stress_nlp.py)."""
    from tests.stress_nlp import run
    lines = []
    failed, worst = run(ocs, oracle, 40, seed=5, log=lines.append)
    assert failed == 0, "\n".join(l for l in lines if "FAILED" in l)
