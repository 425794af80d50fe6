"""Known-answer tests that pin the CPU oracle (SURVEY 8(c) KATs 1-7).

The reference ships no golden vectors (parity with MATLAB itself is unpinned);
these analytic / self-consistency checks are what anchors oracle/ocs_oracle.c.
"""
import numpy as np
import pytest
from scipy.integrate import solve_ivp
from scipy.interpolate import PchipInterpolator

from oracle import np_twin as tw

P = {"c": 1.5, "m": 3.0, "r": 0.05}
BOUNDS = [[0.0, 1.0]]


def _setup(oracle, N=500, T=10.0):
    prob = oracle.TestOCProblem(P, BOUNDS)
    tspan = oracle.linspace(0.0, T, N + 1)
    return prob, tspan, oracle.RK4Integrator(tspan)


def test_linspace_matches_matlab_formula(oracle):
    a = oracle.linspace(0.0, 10.0, 501)
    k = np.arange(501.0)
    assert np.array_equal(a[1:-1], (0.0 + (k * 10.0) / 500.0)[1:-1])
    assert a[0] == 0.0 and a[-1] == 10.0
    assert np.array_equal(a, tw.matlab_linspace(0.0, 10.0, 501))


def test_integrator_grid(oracle):
    # RK4Integrator.m:16-25
    tspan = np.array([0.0, 0.3, 1.0, 1.25])
    g = oracle.RK4Integrator(tspan)
    assert g.nSTEPS == 3
    assert np.array_equal(g.h, np.diff(tspan))
    assert np.array_equal(g.t, [0.0, 0.15, 0.3, 0.65, 1.0, 1.125, 1.25])


def test_equilibrium_is_a_root_of_the_optimality_system(oracle):
    # KAT 1: compute_equilibrium.m:10-21 residual at the analytic equilibrium
    prob = oracle.TestOCProblem(P, BOUNDS)
    xs, ls, us = 2.7355691886341361, 2.1701063402939477, 0.72336878009798256
    F = prob.F(0.0, [xs, 0.0], [us])
    g = prob.dFdx_times_vec(0.0, [xs, 0.0], [us], [ls, 1.0])
    hu = prob.dFdu_times_vec(0.0, [xs, 0.0], [us], [ls, 1.0])
    assert abs(F[0, 0]) < 1e-14
    assert abs(P["r"] * ls - g[0, 0]) < 1e-14
    assert abs(hu[0, 0]) < 1e-14
    # and ControlChar of the adapter returns u* there
    assert abs(prob.ControlChar(0.0, [xs], [ls])[0, 0] - us) < 1e-15


def test_closed_form_state_and_rk4_order(oracle):
    # KAT 2 + 5: u = 0, x0 = 1 -> x(t) = 3 / (1 + 2 exp(-3 t)); RK4 is 4th order
    errs = []
    for N in (250, 500, 1000):
        prob, tspan, g = _setup(oracle, N)
        x, J = g.compute_states(prob, [1.0], np.zeros((1, 2 * N + 1)))
        exact = 3.0 / (1.0 + 2.0 * np.exp(-3.0 * tspan))
        errs.append(np.max(np.abs(x[0] - exact)))
    assert errs[1] < 2e-8
    assert 12 < errs[0] / errs[1] < 20 and 12 < errs[1] / errs[2] < 20
    assert abs(x[0, -1] - 2.9999999999994387) < 1e-9


def test_adjoint_invariants(oracle):
    # KAT 3: lam(end,:) == 1 exactly, lam(1:nS, N+1) == 0, xK slots 2:4 of the last column NaN
    prob, tspan, g = _setup(oracle, 200)
    rng = np.random.default_rng(1)
    u = rng.uniform(0, 1, (1, 401))
    g.compute_states(prob, [1.0], u)
    lam, dJdu = g.compute_adjoints(prob, u)
    assert np.all(lam[-1] == 1.0)
    assert lam[0, -1] == 0.0
    xK = g.xK
    assert np.all(np.isnan(xK[:, -1, 1:])) and not np.any(np.isnan(xK[:, :-1, :]))


def test_c_oracle_equals_numpy_twin(oracle):
    # two independently written restatements must agree to rounding
    for m in ([3.0], [3.0, 2.5, 2.0, 1.5]):
        nS = len(m)
        probC = oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
        probN = tw.TestOCProblemNP(P["c"], m, P["r"])
        tspan = np.sort(np.concatenate([[0.0, 4.0], np.random.default_rng(5).uniform(0, 4, 58)]))
        gC, gN = oracle.RK4Integrator(tspan), tw.RK4IntegratorNP(tspan)
        u = np.random.default_rng(2).uniform(0, 1, (1, 2 * gC.nSTEPS + 1))
        x0 = np.linspace(0.8, 1.3, nS)
        xC, JC = gC.compute_states(probC, x0, u)
        xN, JN = gN.compute_states(probN, x0, u)
        lamC, dC = gC.compute_adjoints(probC, u)
        lamN, dN = gN.compute_adjoints(probN, u)
        np.testing.assert_allclose(xC, xN, rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(lamC, lamN, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(dC, dN, rtol=1e-12, atol=1e-16)
        assert abs(JC - JN) <= 1e-14 * abs(JN)
        lamT = np.random.default_rng(3).normal(size=nS + 1)
        np.testing.assert_allclose(gC.compute_adjoints(probC, u, lamT)[0],
                                   gN.compute_adjoints(probN, u, lamT)[0], rtol=1e-13, atol=1e-15)


def test_test_problem_is_logistic_with_one_state(oracle):
    pT = oracle.TestOCProblem(P, BOUNDS)
    pL = oracle.LogisticProblem([P["m"]], P["c"], P["r"], BOUNDS)
    rng = np.random.default_rng(7)
    t, y, u, v = rng.uniform(0, 10, 6), rng.normal(size=(2, 6)), rng.uniform(0, 1, (1, 6)), rng.normal(size=(2, 6))
    assert np.array_equal(pT.F(t, y, u), pL.F(t, y, u))
    assert np.array_equal(pT.dFdx_times_vec(t, y, u, v), pL.dFdx_times_vec(t, y, u, v))
    assert np.array_equal(pT.dFdu_times_vec(t, y, u, v), pL.dFdu_times_vec(t, y, u, v))


def test_gradient_exactness_complex_step(oracle):
    # KAT 4: the discrete adjoint is the exact gradient of the discrete J.  The complex-step
    # derivative of the NumPy twin (h = 1e-30) is exact to rounding.
    N, nPts = 500, 21
    prob, tspan, g = _setup(oracle, N)
    ctrl = oracle.PWLinearControl(g.t, nPts, 1)
    v = 0.5 + 0.5 * np.random.default_rng(20260404).random(nPts)  # backprop_test.m:19, seeded
    J, dJdv, _ = oracle.nlp_objective(g, prob, ctrl, [1.0], v)
    probN, gN = tw.TestOCProblemNP(P["c"], [P["m"]], P["r"]), tw.RK4IntegratorNP(tspan)
    cs = np.empty(nPts)
    for i in range(nPts):
        vc = v.astype(complex)
        vc[i] += 1e-30j
        _, Jc = gN.compute_states(probN, np.array([1.0]), vc[None, :] @ ctrl.B)
        cs[i] = Jc.imag / 1e-30
    np.testing.assert_allclose(dJdv, cs, rtol=5e-13, atol=1e-15)


def test_backprop_test_script(oracle):
    # tests/backprop_test.m:5-43 with its forward differences, eps = 1e-4
    N, nPts, eps = 500, 21, 1e-4
    prob, tspan, g = _setup(oracle, N)
    ctrl = oracle.PWLinearControl(g.t, nPts, 1)
    v = 0.5 + 0.5 * np.random.default_rng(20260404).random(nPts)
    u = ctrl.compute_u(v)
    _, J = g.compute_states(prob, [1.0], u)
    _, dJdu = g.compute_adjoints(prob, u)
    dJdv = ctrl.compute_dJdv(dJdu)
    fd = np.empty(nPts)
    for i in range(nPts):
        v[i] += eps
        _, Jfd = g.compute_states(prob, [1.0], ctrl.compute_u(v))
        fd[i] = (Jfd - J) / eps
        v[i] -= eps
    assert np.max(np.abs(fd - dJdv)) < 1e-3
    # central differences tighten it (truncation eps^2, round-off |J| ulp / eps)
    cd = np.empty(nPts)
    for i in range(nPts):
        e = 1e-5
        v[i] += e
        _, Jp = g.compute_states(prob, [1.0], ctrl.compute_u(v))
        v[i] -= 2 * e
        _, Jm = g.compute_states(prob, [1.0], ctrl.compute_u(v))
        v[i] += e
        cd[i] = (Jp - Jm) / (2 * e)
    assert np.max(np.abs(cd - dJdv)) < 5e-8


def test_continuum_limit_against_dop853(oracle):
    # KAT 6: J_RK4(N) vs an independent adaptive solver with the same piecewise-linear u
    N, nPts = 1000, 11
    prob, tspan, g = _setup(oracle, N)
    ctrl = oracle.PWLinearControl(g.t, nPts, 1)
    v = 0.3 + 0.4 * np.random.default_rng(11).random(nPts)
    J, _, _ = oracle.nlp_objective(g, prob, ctrl, [1.0], v)
    cp = ctrl.controlPts

    def rhs(t, y):
        u = np.interp(t, cp, v)
        return [y[0] * (P["m"] - y[0]) - u, np.exp(-P["r"] * t) * (y[0] ** 2 + P["c"] * u * u)]

    Jref, y = 0.0, [1.0, 0.0]
    for a, b in zip(cp[:-1], cp[1:]):  # integrate kink to kink
        s = solve_ivp(rhs, (a, b), y, method="DOP853", rtol=1e-13, atol=1e-14)
        y = s.y[:, -1]
    assert abs(J - y[1]) < 2e-9 * abs(y[1])


def test_basis_matrices(oracle):
    # KAT 7
    g = oracle.RK4Integrator(oracle.linspace(0, 10, 501))
    t = g.t
    Bl = oracle.PWLinearControl(t, 101, 1).B
    np.testing.assert_allclose(Bl.sum(axis=0), 1.0, rtol=0, atol=2e-15)
    assert (np.count_nonzero(Bl, axis=0) <= 2).all() and Bl.min() >= 0
    np.testing.assert_allclose(Bl, tw.pwlinear_basis(t, 101)[0], rtol=0, atol=1e-14)
    Bc = oracle.PWConstantControl(t, 50, 1).B
    assert np.array_equal(Bc.sum(axis=0), np.ones(t.size)) and (np.count_nonzero(Bc, axis=0) == 1).all()
    assert np.array_equal(Bc, tw.pwconstant_basis(t, 50)[0])
    Bch = oracle.ChebyshevControl(t, 16, 1).B
    tau = 2 * (t - t[0]) / (t[-1] - t[0]) - 1
    k = np.arange(16)[:, None]
    np.testing.assert_allclose(Bch, np.cos(k * np.arccos(np.clip(tau, -1, 1))), atol=2e-13)
    assert np.array_equal(Bch, tw.chebyshev_basis(t, 16))


def test_controls_u_dJdv_initial_v_bounds_ufunc(oracle):
    g = oracle.RK4Integrator(oracle.linspace(0, 4, 41))
    rng = np.random.default_rng(3)
    for cls, nB in ((oracle.PWLinearControl, 9), (oracle.PWConstantControl, 8), (oracle.ChebyshevControl, 6)):
        c = cls(g.t, nB, 2)
        v = rng.normal(size=2 * nB)
        u = c.compute_u(v)
        np.testing.assert_allclose(u, v.reshape(2, nB, order="F") @ c.B, rtol=1e-14, atol=1e-15)
        d = rng.normal(size=u.shape)
        np.testing.assert_allclose(c.compute_dJdv(d), (d @ c.B.T).reshape(-1, order="F"), rtol=1e-13, atol=1e-14)
        # uFunc reproduces u on the integrator grid
        np.testing.assert_allclose(c.compute_uFunc(v)(g.t), u, rtol=1e-12, atol=1e-12)
    c = oracle.PWLinearControl(g.t, 9, 2)
    assert np.array_equal(c.compute_initial_v([0.25, 0.5]), np.tile([0.25, 0.5], 9))
    Lb, Ub = c.compute_nlp_bounds([[0.0, 1.0], [-2.0, 3.0]])
    assert np.array_equal(Lb, np.tile([0.0, -2.0], 9)) and np.array_equal(Ub, np.tile([1.0, 3.0], 9))
    ch = oracle.ChebyshevControl(g.t, 6, 1)
    assert np.array_equal(ch.compute_initial_v([0.7]), [0.7, 0, 0, 0, 0, 0])
    with pytest.raises(ValueError):
        c.compute_initial_v([1.0, 2.0, 3.0])


def test_pchip_matches_scipy(oracle):
    rng = np.random.default_rng(9)
    x = np.sort(rng.uniform(0, 5, 40))
    for y in (np.sin(3 * x), np.cumsum(rng.normal(size=40)), np.where(x > 2, 1.0, 0.0)):
        np.testing.assert_allclose(oracle.pchip_slopes(x, y), PchipInterpolator(x, y).derivative()(x),
                                   rtol=1e-12, atol=1e-13)
        q = np.concatenate([x, rng.uniform(x[0], x[-1], 300)])
        np.testing.assert_allclose(oracle.interp1(x, y, oracle.INTERP_PCHIP, q), PchipInterpolator(x, y)(q),
                                   rtol=1e-12, atol=1e-13)
    # node values come back exactly
    y = np.sin(3 * x)
    assert np.array_equal(oracle.interp1(x, y, oracle.INTERP_PCHIP, x), y)


def test_infinite_integrator(oracle):
    # RK4InfiniteIntegrator.m:20-30: J = J1 + J2, terminal adjoint of leg 1 = lam2(:,1)
    prob = oracle.TestOCProblem(P, BOUNDS)
    tspan, tx = oracle.linspace(0, 10, 101), oracle.linspace(10, 20, 101)
    us = 0.72336878009798256
    gi = oracle.RK4InfiniteIntegrator(tspan, tx, [us])
    g1, g2 = oracle.RK4Integrator(tspan), oracle.RK4Integrator(tx)
    u = np.random.default_rng(4).uniform(0, 1, (1, 201))
    x, J = gi.compute_states(prob, [1.0], u)
    lam, dJdu = gi.compute_adjoints(prob, u)
    x1, J1 = g1.compute_states(prob, [1.0], u)
    us_row = us * np.ones((1, 201))
    _, J2 = g2.compute_states(prob, x1[:-1, -1], us_row)
    lam2 = g2.compute_adjoints(prob, us_row, want_dJdu=False)
    lam1, dJdu1 = g1.compute_adjoints(prob, u, lam2[:, 0])
    assert J == J1 + J2 and np.array_equal(x, x1)
    assert np.array_equal(lam, lam1) and np.array_equal(dJdu, dJdu1)
    # gradient of the two-leg objective by central differences
    j = 57
    e = 1e-5
    up, um = u.copy(), u.copy()
    up[0, j] += e
    um[0, j] -= e
    fd = (gi.compute_states(prob, [1.0], up)[1] - gi.compute_states(prob, [1.0], um)[1]) / (2 * e)
    assert abs(fd - dJdu[0, j]) < 2e-7


def test_free_initial_states_gradient(oracle):
    # single_shooting.m:144-149: dJdv = [dJdv ; lam(FreeInitStates,1)]
    prob = oracle.LogisticProblem([3.0, 2.0], P["c"], P["r"], BOUNDS)
    g = oracle.RK4Integrator(oracle.linspace(0, 5, 101))
    ctrl = oracle.PWLinearControl(g.t, 6, 1)
    v = np.concatenate([np.full(6, 0.4), [1.7]])
    J, dJdv, x0 = oracle.nlp_objective(g, prob, ctrl, [1.0, 1.0], v, FreeInitStates=[2])
    assert x0[1] == 1.7
    e = 1e-5
    vp, vm = v.copy(), v.copy()
    vp[-1] += e
    vm[-1] -= e
    fd = (oracle.nlp_objective(g, prob, ctrl, [1.0, 1.0], vp, [2])[0]
          - oracle.nlp_objective(g, prob, ctrl, [1.0, 1.0], vm, [2])[0]) / (2 * e)
    assert abs(fd - dJdv[-1]) < 2e-7


def test_fb_sweep_is_grid_dependent_and_converges_with_the_grid(oracle):
    """The reference's fb_sweep integrates with adaptive odevr7 (RelTol = AbsTol = 5e-14, fb_sweep.m:18-19); the build's
    scheme is RK4 on tspan with pchip coupling, so its answers depend on the grid (README / INTEGRATION say so).  What
    holds instead: the converged J approaches a limit as the grid is refined, with differences shrinking at the
    scheme's order (>= 3: pchip midpoints are the lowest-order piece), and the limit is the continuum optimum that an
    independent adaptive solver reproduces from the converged control."""
    prob = oracle.TestOCProblem(P, [[0.0, 1.0]])
    Js, sol = [], None
    for N in (125, 250, 500, 1000):
        sol = oracle.fb_sweep(prob, [1.0], oracle.linspace(0, 10, N + 1), {"nERROR_PTS": N + 1, "nINTERP_PTS": N + 1})
        assert sol["_sweeps"] > 0
        Js.append(sol["J"])
    d = np.abs(np.diff(Js))
    assert d[0] > d[1] > d[2] and d[2] < 1e-7
    assert np.log2(d[0] / d[1]) > 2.5 and np.log2(d[1] / d[2]) > 2.5
    # continuum check: integrate state + objective with DOP853 under the converged control (pchip samples on the
    # finest grid); agrees with the grid J to the truncation level measured above
    from scipy.interpolate import PchipInterpolator
    tq = sol["_interpPts"]
    uf = PchipInterpolator(tq, sol["u"][0])

    def rhs(t, y):
        u = float(uf(t))
        return [y[0] * (P["m"] - y[0]) - u, np.exp(-P["r"] * t) * (y[0] ** 2 + P["c"] * u * u)]
    s = solve_ivp(rhs, (0.0, 10.0), [1.0, 0.0], method="DOP853", rtol=1e-12, atol=1e-13, max_step=0.05)
    assert abs(s.y[1, -1] - Js[-1]) < 1e-6 * abs(Js[-1])


def test_tuned_cpu_baseline_agrees_with_the_restatement(oracle):
    """oracle/ocs_cpu_fast.c (bench.py's cpu_baseline: the same arithmetic in vector loops over blocks of 64 trajectories,
    stage states recomputed, FMA contraction on) against the literal restatement of RK4Integrator.m:28-121: round-off."""
    rng = np.random.default_rng(12)
    for nS, N, B in ((4, 120, 150), (1, 33, 64), (2, 7, 3)):
        m = [3.0, 2.5, 2.0, 1.5][:nS]
        tspan = np.concatenate([[0.0], np.sort(rng.uniform(0, 6, N - 1)), [6.0]])
        x0, u = rng.uniform(0.8, 1.8, (nS, B)), rng.uniform(0.0, 0.5, (2 * N + 1, B))
        prob = oracle.LogisticProblem(m, 1.5, 0.05, [[0.0, 1.0]])
        ref = oracle.batch_states_adjoints(prob, tspan, x0, np.asfortranarray(u[None]), nthreads=2)
        f = oracle.fast_logistic_pair(m, 1.5, 0.05, tspan, x0, u, nthreads=2)
        rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))
        assert rel(f["x"].transpose(1, 0, 2), ref["x"]) < 1e-13 and rel(f["J"], ref["J"]) < 1e-13
        assert rel(f["lam"].transpose(1, 0, 2), ref["lam"]) < 1e-13 and rel(f["dJdu"][None], ref["dJdu"]) < 1e-13
