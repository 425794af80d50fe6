"""GPU parity of user-written OCProblem plugins compiled with hipRTC -- the open
plugin surface of OCProblem/OCProblem.m -- including a coupled problem that is not in the registry, and
the build-defined LQ problem of BASELINE config 5 (nC > 1, RK4InfiniteIntegrator) at a reduced size."""
import numpy as np
import pytest

from oracle import np_twin as tw
from tests.user_problems import (LOGISTIC2_SRC, LOGISTIC_ROWS_SRC, PREDPREY_PARAMS, PREDPREY_SRC, PROPHARVEST_ROWS_SRC,
                                 PredPreyNP, PropHarvestNP, lq_matrices, lq_source)

pytestmark = pytest.mark.gpu
RTOL = 1e-12
BOUNDS = [[0.0, 1.0]]


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


@pytest.fixture(scope="module")
def ocs():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    return g.load_package()


def test_hand_written_logistic2_equals_builtin_and_oracle(ocs, oracle):
    c, r, m = 1.5, 0.05, [3.0, 2.5]
    pu = ocs.UserProblem(LOGISTIC2_SRC, 2, 1, [c, r] + m, BOUNDS, has_control_char=True)
    pb, po = ocs.LogisticProblem(m, c, r, BOUNDS), oracle.LogisticProblem(m, c, r, BOUNDS)
    rng = np.random.default_rng(1)
    t, y = rng.uniform(0, 10, 9), rng.normal(1.5, 0.5, (3, 9))
    u, v = rng.uniform(0, 1, (1, 9)), rng.normal(size=(3, 9))
    assert relerr(pu.F(t, y, u), po.F(t, y, u)) < 1e-14
    assert relerr(pu.dFdx_times_vec(t, y, u, v), po.dFdx_times_vec(t, y, u, v)) < 1e-14
    assert relerr(pu.dFdu_times_vec(t, y, u, v), po.dFdu_times_vec(t, y, u, v)) < 1e-14
    N, batch = 203, 70
    tspan = oracle.linspace(0, 10, N + 1)
    uu = rng.uniform(0.05, 0.45, (1, 2 * N + 1, batch))
    x0 = rng.uniform(0.9, 2.0, (2, batch))
    g = ocs.RK4Integrator(tspan)
    x, J = g.compute_states(pu, x0, uu)
    lam, dJdu = g.compute_adjoints(pu, uu)
    ref = oracle.batch_states_adjoints(po, tspan, x0, uu)
    assert relerr(x, ref["x"]) < RTOL and relerr(J, ref["J"]) < RTOL
    assert relerr(lam, ref["lam"]) < RTOL and relerr(dJdu, ref["dJdu"]) < RTOL
    xb, Jb = g.compute_states(pb, x0, uu)
    assert relerr(x, xb) < 1e-13
    # per-trajectory parameter override works for user problems with <= 16 parameters
    cs = rng.uniform(1, 2, batch)
    pu.set_batch_params([0], cs[None, :])
    _, Jc = g.compute_states(pu, x0, uu)
    go = oracle.RK4Integrator(tspan)
    _, Jo = go.compute_states(oracle.LogisticProblem(m, cs[5], r, BOUNDS), x0[:, 5], uu[:, :, 5])
    assert abs(Jc[5] - Jo) < RTOL * abs(Jo)
    pu.set_batch_params([], None)
    # compute_equilibrium on the user's plugin methods (hipRTC instance of k_equilibrium) == the registry problem's
    big = [np.inf] * 2
    lb, ub = [0, 0, -np.inf, -np.inf, 0], big + big + [np.inf]
    eu = ocs.compute_equilibrium(pu, [2.6, 2.1], [1.2, 1.2], [0.6], lb, ub, r)
    eb = ocs.compute_equilibrium(pb, [2.6, 2.1], [1.2, 1.2], [0.6], lb, ub, r)
    # (this system has no root: both return the same least-squares point, see test_compute_equilibrium_batched_on_device)
    assert eu[5] == 1 and abs(eu[3] - eb[3]) < 1e-12 and relerr(np.concatenate(eu[:3]), np.concatenate(eb[:3])) < 1e-9
    # fb_sweep through the user's ocs_ControlChar
    s1 = ocs.fb_sweep_batch(pu, np.array([[1.0], [1.5]]), oracle.linspace(0, 8, 161), {"nERROR_PTS": 161, "nINTERP_PTS": 81})
    s2 = oracle.fb_sweep(po, [1.0, 1.5], oracle.linspace(0, 8, 161), {"nERROR_PTS": 161, "nINTERP_PTS": 81})
    assert s1["sweeps"][0] == s2["_sweeps"] > 0 and abs(s1["J"][0] - s2["J"]) < 1e-10 * abs(s2["J"])
    assert relerr(s1["u"][:, :, 0], s2["u"]) < 1e-10


@pytest.mark.parametrize("nS,N,batch", [(4, 1000, 64), (2, 203, 70), (1, 96, 128), (4, 37, 17)])
def test_row_separable_user_problem_on_the_fast_mappings(ocs, oracle, nS, N, batch):
    """A user problem given as ROW functions (hipRTC) runs on the mappings of the registry problems -- the
    wave-specialised state pass k_forward_p2 and the scan adjoint pass k_backward_scan, instantiated for it -- and on
    the lane kernels through the derived full-vector methods: all against the oracle (the hand-written LogisticK rows
    must reproduce the built-in problem), incl. remainders of the 8- / 4-step blockings, ragged tiles, explicit lamT."""
    import torch
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    c, r = 1.5, 0.05
    pu = ocs.UserProblem(LOGISTIC_ROWS_SRC, nS, 1, [c, r] + m, BOUNDS, row_separable=True)
    pb, po = ocs.LogisticProblem(m, c, r, BOUNDS), oracle.LogisticProblem(m, c, r, BOUNDS)
    rng = np.random.default_rng(10 + nS)
    t, y = rng.uniform(0, 10, 9), rng.normal(1.5, 0.5, (nS + 1, 9))
    u, v = rng.uniform(0, 1, (1, 9)), rng.normal(size=(nS + 1, 9))
    assert relerr(pu.F(t, y, u), po.F(t, y, u)) < 1e-14                      # derived full-vector methods
    assert relerr(pu.dFdx_times_vec(t, y, u, v), po.dFdx_times_vec(t, y, u, v)) < 1e-14
    assert relerr(pu.dFdu_times_vec(t, y, u, v), po.dFdu_times_vec(t, y, u, v)) < 1e-14
    tspan = oracle.linspace(0, 10, N + 1)
    uu = rng.uniform(0.05, 0.45, (1, 2 * N + 1, batch))
    x0 = rng.uniform(0.9, 2.0, (nS, batch))
    ref = oracle.batch_states_adjoints(po, tspan, x0, uu)
    for mapping in ("auto", "lane"):
        g = ocs.RK4Integrator(tspan).set_mapping(mapping)
        x, J = g.compute_states(pu, x0, uu)
        lam, dJdu = g.compute_adjoints(pu, uu)
        assert relerr(x, ref["x"]) < RTOL and relerr(J, ref["J"]) < RTOL
        assert relerr(lam, ref["lam"]) < RTOL and relerr(dJdu, ref["dJdu"]) < RTOL
        assert np.all(lam[-1] == 1.0)
        lamT = np.random.default_rng(3).normal(size=(nS + 1, batch))
        lam2, d2 = g.compute_adjoints(pu, uu, lamT)
        go = oracle.RK4Integrator(tspan)
        for b in (0, batch - 1):
            go.compute_states(po, x0[:, b], uu[:, :, b])
            lo, do = go.compute_adjoints(po, uu[:, :, b], lamT[:, b])
            assert relerr(lam2[:, :, b], lo) < RTOL and relerr(d2[:, :, b], do) < RTOL
    if (nS, N) != (4, 1000):
        return
    # BASELINE batch: the row-function instance against the registry problem (same kernels, generic row functions
    # instead of the shifted logistic form), timing reported and bounded
    B = 4096
    dev = torch.device("cuda:0")
    tspan = np.linspace(0, 10, N + 1)
    x0d = torch.ones((nS, B), dtype=torch.float64, device=dev)
    ud = 0.05 + 0.4 * torch.rand((2 * N + 1, 1, B), dtype=torch.float64, device=dev)
    outs, times = {}, {}
    for name, prob in (("registry", pb), ("rows", pu)):
        g = ocs.RK4Integrator(tspan)
        xd = torch.empty((N + 1, nS + 1, B), dtype=torch.float64, device=dev)
        lamd, dd = torch.empty_like(xd), torch.empty_like(ud)
        for _ in range(3):
            _, Jd = g.compute_states_dev(prob, x0d, ud, xd)
            g.compute_adjoints_dev(prob, ud, None, lamd, dd)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(20):
            g.compute_states_dev(prob, x0d, ud, xd)
            g.compute_adjoints_dev(prob, ud, None, lamd, dd)
        torch.cuda.synchronize()
        times[name] = (time.perf_counter() - t0) / 20
        outs[name] = (xd.cpu().numpy(), lamd.cpu().numpy(), dd.cpu().numpy())
    for a_, b_ in zip(outs["rows"], outs["registry"]):
        assert relerr(a_, b_) < 1e-12
    # (reported, not asserted: a wall-clock bound on a shared GPU pool fails without a code defect; the steady-state
    #  comparison is scripts/user_rows_time.py: 151 against 140 us)
    print(f"pass pair at batch 4096: registry {times['registry']*1e6:.1f} us, row functions {times['rows']*1e6:.1f} us")


def test_row_separable_problem_whose_control_gradient_reads_the_state(ocs, oracle):
    """x_r' = x_r (m_r - x_r) - u x_r / (1 + r): dF/du depends on y, so the scan kernel recomputes the stage state below
    each chunk for the chunk's lowest dJdu column; against the NumPy twin of the reference's integrator."""
    m, c, r = [3.0, 2.5], 1.5, 0.05
    pu = ocs.UserProblem(PROPHARVEST_ROWS_SRC, 2, 1, [c, r] + m, BOUNDS, row_separable=True)
    pn = PropHarvestNP(m, c, r)
    N, batch = 100, 40
    tspan = oracle.linspace(0, 6, N + 1)
    rng = np.random.default_rng(5)
    uu = rng.uniform(0.0, 1.0, (1, 2 * N + 1, batch))
    x0 = rng.uniform(1.0, 2.5, (2, batch))
    gn = tw.RK4IntegratorNP(tspan)
    for mapping in ("auto", "lane"):
        g = ocs.RK4Integrator(tspan).set_mapping(mapping)
        x, J = g.compute_states(pu, x0, uu)
        lam, dJdu = g.compute_adjoints(pu, uu)
        for b in (0, 1, 39):
            xn, Jn = gn.compute_states(pn, x0[:, b], uu[:, :, b])
            lamn, dn = gn.compute_adjoints(pn, uu[:, :, b])
            assert relerr(x[:, :, b], xn) < RTOL and abs(J[b] - Jn) < RTOL * max(1, abs(Jn))
            assert relerr(lam[:, :, b], lamn) < RTOL and relerr(dJdu[:, :, b], dn) < RTOL


def test_coupled_problem_outside_the_registry(ocs, oracle):
    pu, pn = ocs.UserProblem(PREDPREY_SRC, 2, 1, PREDPREY_PARAMS, BOUNDS), PredPreyNP()
    N, batch = 120, 66
    tspan = oracle.linspace(0, 6, N + 1)
    rng = np.random.default_rng(2)
    uu = rng.uniform(0.0, 1.0, (1, 2 * N + 1, batch))
    x0 = rng.uniform(1.0, 2.5, (2, batch))
    g = ocs.RK4Integrator(tspan)
    x, J = g.compute_states(pu, x0, uu)
    lam, dJdu = g.compute_adjoints(pu, uu)
    gn = tw.RK4IntegratorNP(tspan)
    for b in (0, 1, 64, 65):
        xn, Jn = gn.compute_states(pn, x0[:, b], uu[:, :, b])
        lamn, dn = gn.compute_adjoints(pn, uu[:, :, b])
        assert relerr(x[:, :, b], xn) < RTOL and abs(J[b] - Jn) < RTOL * max(1, abs(Jn))
        assert relerr(lam[:, :, b], lamn) < RTOL and relerr(dJdu[:, :, b], dn) < RTOL
    # objective + gradient through a control basis; gradient exactness by complex step on the NumPy twin
    ctrl = ocs.PWLinearControl(g.t, 7, 1)
    v = rng.uniform(0.1, 0.9, 7)
    Jg, dJdv, _ = ocs.nlp_objective(g, pu, ctrl, x0[:, 0], v)
    cs = np.empty(7)
    for i in range(7):
        vc = v.astype(complex)
        vc[i] += 1e-30j
        cs[i] = gn.compute_states(pn, x0[:, 0].astype(complex), vc[None, :] @ ctrl.B)[1].imag / 1e-30
    assert relerr(dJdv, cs) < 1e-11
    with pytest.raises(ocs.OcsError):
        ocs.fb_sweep_batch(pu, x0[:, :1], tspan)  # no ocs_ControlChar in this plugin
    # the same problem with its discount factor hoisted into the step records (OCS_USER_TCOEF: the methods receive
    # ocs_tcoef(t, p) = e^{-rt} in the place of t): plugin evaluation, both passes on the lane kernels and on the vector mappings
    from tests.user_problems import PREDPREY_TC_SRC
    pt = ocs.UserProblem(PREDPREY_TC_SRC, 2, 1, PREDPREY_PARAMS, BOUNDS)
    tt, yy = rng.uniform(0, 6, 9), rng.uniform(0.5, 2.5, (3, 9))
    ue, ve = rng.uniform(0, 1, (1, 9)), rng.normal(size=(3, 9))
    assert relerr(pt.F(tt, yy, ue), pn.F(tt, yy, ue)) < 1e-14
    assert relerr(pt.dFdx_times_vec(tt, yy, ue, ve), pn.dFdx_times_vec(tt, yy, ue, ve)) < 1e-14
    assert relerr(pt.dFdu_times_vec(tt, yy, ue, ve), pn.dFdu_times_vec(tt, yy, ue, ve)) < 1e-14
    for nb, mapping in ((batch, "auto"), (64, "auto"), (64, "lane")):
        gt = ocs.RK4Integrator(tspan).set_mapping(mapping)
        xt, Jt = gt.compute_states(pt, x0[:, :nb], uu[:, :, :nb])
        lamt, dt = gt.compute_adjoints(pt, uu[:, :, :nb])
        assert relerr(xt, x[:, :, :nb]) < 1e-13 and relerr(Jt, J[:nb]) < 1e-13
        assert relerr(lamt, lam[:, :, :nb]) < 1e-12 and relerr(dt, dJdu[:, :, :nb]) < 1e-12


def test_lq_problem_bl5_style(ocs, oracle):
    """BASELINE config 5 at reduced size: LQ with a dense shared A, nC = 2, RK4InfiniteIntegrator with uStar = 0."""
    nS, nC, N, batch = 6, 2, 160, 68
    A, Bu, q, rdiag = lq_matrices(nS, nC)
    r = 0.05
    par = np.concatenate([[r], A.ravel(order="F"), Bu.ravel(order="F"), q, rdiag])
    bounds = [[-1.0, 1.0]] * nC
    pu = ocs.UserProblem(lq_source(nS, nC), nS, nC, par, bounds)
    po = oracle.LQProblem(A, Bu, q, rdiag, r, bounds)
    tspan, tx = oracle.linspace(0, 2, N + 1), oracle.linspace(2, 4, N + 1)
    rng = np.random.default_rng(3)
    u = rng.uniform(-1, 1, (nC, 2 * N + 1, batch))
    x0 = rng.normal(size=(nS, batch))
    gi, go = ocs.RK4InfiniteIntegrator(tspan, tx, np.zeros(nC)), oracle.RK4InfiniteIntegrator(tspan, tx, np.zeros(nC))
    x, J = gi.compute_states(pu, x0, u)
    lam, dJdu = gi.compute_adjoints(pu, u)
    for b in (0, 1, 66, 67):
        xo, Jo = go.compute_states(po, x0[:, b], u[:, :, b])
        lamo, do = go.compute_adjoints(po, u[:, :, b])
        assert relerr(x[:, :, b], xo) < RTOL and abs(J[b] - Jo) < RTOL * max(1, abs(Jo))
        assert relerr(lam[:, :, b], lamo) < RTOL and relerr(dJdu[:, :, b], do) < RTOL
    # nC = 2 through the control bases and the shooting objective
    cg, co = ocs.PWConstantControl(gi.t, 5, nC), oracle.PWConstantControl(go.t, 5, nC)
    V = rng.uniform(-1, 1, (nC * 5, 3))
    Jn, dJdv, _ = ocs.nlp_objective(gi, pu, cg, x0[:, :3], V)
    for b in range(3):
        Jo, do, _ = oracle.nlp_objective(go, po, co, x0[:, b], V[:, b])
        assert abs(Jn[b] - Jo) < RTOL * max(1, abs(Jo)) and relerr(dJdv[:, b], do) < RTOL


@pytest.mark.parametrize("N,batch", [(120, 128), (64, 64), (203, 192), (8, 64), (1000, 256)])
def test_vector_mappings_for_coupled_problems(ocs, oracle, N, batch):
    """OCProblem/OCProblem.m:8-21 allows any coupled F: the predator-prey plugin (full-vector methods through hipRTC) on the
    vector-lane state pass (csrc/ocs_pipelinev_kernel.hpp: whole tiles of 64 trajectories, whole blocks of 8 steps, the
    rest of the steps on the lane kernel) and the scan adjoint with dense 2 x 2 step maps (csrc/ocs_vscan_kernel.hpp)
    against the NumPy twin of RK4Integrator.m and against the lane kernels; explicit lamT, lam only, dJdu only."""
    pu, pn = ocs.UserProblem(PREDPREY_SRC, 2, 1, PREDPREY_PARAMS, BOUNDS), PredPreyNP()
    tspan = oracle.linspace(0, 6 if N > 8 else 0.5, N + 1)
    rng = np.random.default_rng(N)
    uu = rng.uniform(0.0, 1.0, (1, 2 * N + 1, batch))
    x0 = rng.uniform(1.0, 2.5, (2, batch))
    out = {}
    for mapping in ("auto", "lane"):
        g = ocs.RK4Integrator(tspan).set_mapping(mapping)
        x, J = g.compute_states(pu, x0, uu)
        lam, dJdu = g.compute_adjoints(pu, uu)
        out[mapping] = (x, J, lam, dJdu)
    for a_, b_ in zip(out["auto"], out["lane"]):
        assert relerr(a_, b_) < RTOL
    assert np.array_equal(out["auto"][0][:2], out["lane"][0][:2])     # the state rows: the same operations in the same order
    x, J, lam, dJdu = out["auto"]
    assert np.array_equal(J, x[2, -1, :])                              # J == x(end, end)   RK4Integrator.m:55
    gn = tw.RK4IntegratorNP(tspan)
    for b in sorted({0, 1, 63, batch // 2, batch - 1}):
        xn, Jn = gn.compute_states(pn, x0[:, b], uu[:, :, b])
        lamn, dn = gn.compute_adjoints(pn, uu[:, :, b])
        assert relerr(x[:, :, b], xn) < RTOL and abs(J[b] - Jn) < RTOL * max(1, abs(Jn))
        assert relerr(lam[:, :, b], lamn) < RTOL and relerr(dJdu[:, :, b], dn) < RTOL
    # explicit lamT (RK4Integrator.m:63-66), and the outputs on their own
    g = ocs.RK4Integrator(tspan)
    lamT = rng.normal(size=(3, batch))
    g.compute_states(pu, x0, uu)
    lamL, dL = g.compute_adjoints(pu, uu, lamT)
    gl = ocs.RK4Integrator(tspan).set_mapping("lane")
    gl.compute_states(pu, x0, uu)
    lamR, dR = gl.compute_adjoints(pu, uu, lamT)
    assert relerr(lamL, lamR) < RTOL and relerr(dL, dR) < RTOL
    import torch
    dev = torch.device("cuda:0")
    ud, x0d = torch.tensor(np.ascontiguousarray(uu.transpose(1, 0, 2)), device=dev), torch.tensor(x0, device=dev)
    g.compute_states_dev(pu, x0d, ud)
    lam_only, _ = g.compute_adjoints_dev(pu, ud, None, torch.empty((N + 1, 3, batch), dtype=torch.float64, device=dev), None)
    _, d_only = g.compute_adjoints_dev(pu, ud, None, None, torch.empty_like(ud))
    torch.cuda.synchronize()
    assert np.array_equal(lam_only.cpu().numpy().transpose(1, 0, 2), lam)
    assert np.array_equal(d_only.cpu().numpy().transpose(1, 0, 2), dJdu)


def test_vector_mappings_two_controls_and_three_states(ocs, oracle):
    """nC = 2 and nS = 3 (neither has a row-split mapping): the LQ plugin source with three states and two controls on the
    vector-lane state pass and the dense-map scan, against the oracle's LQ problem; and the registry's three-state logistic
    problem (LogisticK<3>) against the oracle."""
    nS, nC, N, batch = 3, 2, 96, 128
    A, Bu, q, rdiag = lq_matrices(nS, nC)
    r = 0.05
    par = np.concatenate([[r], A.ravel(order="F"), Bu.ravel(order="F"), q, rdiag])
    bounds = [[-1.0, 1.0]] * nC
    pu = ocs.UserProblem(lq_source(nS, nC), nS, nC, par, bounds)
    po = oracle.LQProblem(A, Bu, q, rdiag, r, bounds)
    tspan = oracle.linspace(0, 2, N + 1)
    rng = np.random.default_rng(8)
    uu = rng.uniform(-1, 1, (nC, 2 * N + 1, batch))
    x0 = rng.normal(size=(nS, batch))
    g = ocs.RK4Integrator(tspan)
    x, J = g.compute_states(pu, x0, uu)
    lam, dJdu = g.compute_adjoints(pu, uu)
    ref = oracle.batch_states_adjoints(po, tspan, x0, uu)
    assert relerr(x, ref["x"]) < RTOL and relerr(J, ref["J"]) < RTOL
    assert relerr(lam, ref["lam"]) < RTOL and relerr(dJdu, ref["dJdu"]) < RTOL
    gl = ocs.RK4Integrator(tspan).set_mapping("lane")
    xl, Jl = gl.compute_states(pu, x0, uu)
    assert relerr(x, xl) < 1e-13 and relerr(J, Jl) < 1e-13
    # registry, three states
    m = [3.0, 2.5, 2.0]
    pb, pob = ocs.LogisticProblem(m, 1.5, 0.05, BOUNDS), oracle.LogisticProblem(m, 1.5, 0.05, BOUNDS)
    N2 = 200
    ts2 = oracle.linspace(0, 10, N2 + 1)
    u2 = rng.uniform(0.05, 0.45, (1, 2 * N2 + 1, 64))
    x02 = rng.uniform(0.9, 2.0, (3, 64))
    g2 = ocs.RK4Integrator(ts2)
    x2, J2 = g2.compute_states(pb, x02, u2)
    lam2, d2 = g2.compute_adjoints(pb, u2)
    ref2 = oracle.batch_states_adjoints(pob, ts2, x02, u2)
    assert relerr(x2, ref2["x"]) < RTOL and relerr(J2, ref2["J"]) < RTOL
    assert relerr(lam2, ref2["lam"]) < RTOL and relerr(d2, ref2["dJdu"]) < RTOL


@pytest.mark.parametrize("nS,N,batch,grid,hoist", [(2, 1000, 64, "lin", False), (1, 160, 128, "lin", True),
                                                   (4, 160, 32, "rand", False), (2, 96, 96, "np", True),
                                                   (1, 1000, 192, "lin", False), (4, 200, 48, "np", True)])
def test_fb_sweep_two_kernel_sweep_for_user_row_functions(ocs, oracle, nS, N, batch, grid, hoist):
    """functions/fb_sweep.m:79-115 for a hipRTC problem given as row functions whose ocs_ControlChar reads the costate alone
    (flag bit 2, control_from_costate): the two-kernel sweep of the registry problems (k_forward_cc: state pass forming its
    control from the costate of the sweep before; k_costate_scan: costate pass as a scan with check_convergence inside),
    instantiated with hipRTC.  Against the oracle per instance, against the same source without the declaration (the
    kernel-by-kernel sequence) and against the registry problem of the same equations; uniform, numpy-linspace and
    non-uniform grids, nS = 1, 2, 4, one to three workgroups; `hoist`: the source tabulates ControlChar's exponential
    through ocs_cc_tcoef (OCS_USER_CC_TCOEF)."""
    from tests.user_problems import LOGISTIC_ROWS_CC_SRC, LOGISTIC_ROWS_CCT_SRC
    SRC = LOGISTIC_ROWS_CCT_SRC if hoist else LOGISTIC_ROWS_CC_SRC
    c, r = 1.5, 0.05
    m = [3.0, 2.5, 2.0, 3.5][:nS]
    rng = np.random.default_rng(100 * nS + N)
    T = 8.0
    tspan = {"lin": oracle.linspace(0, T, N + 1), "np": np.linspace(0, T, N + 1),
             "rand": np.concatenate([[0.0], np.sort(rng.uniform(0, T, N - 1)), [T]])}[grid]
    if grid == "rand":   # keep the steps within a factor of the mean (the sweep has to converge on it)
        tspan = 0.5 * (tspan + oracle.linspace(0, T, N + 1))
    x0 = rng.uniform(0.8, 1.6, (nS, batch))
    kw = dict(has_control_char=True, row_separable=True)
    fold = ocs.UserProblem(SRC, nS, 1, [c, r] + m, BOUNDS, control_from_costate=True, **kw)
    plain = ocs.UserProblem(SRC, nS, 1, [c, r] + m, BOUNDS, **kw)
    reg = ocs.LogisticProblem(m, c, r, BOUNDS)
    opts = {"nERROR_PTS": N + 1, "nINTERP_PTS": 81}
    gf, gp, gr = (ocs.RK4Integrator(tspan) for _ in range(3))
    sf = ocs.fb_sweep_batch(fold, x0, tspan, opts, integrator=gf)
    sp = ocs.fb_sweep_batch(plain, x0, tspan, opts, integrator=gp)
    sr = ocs.fb_sweep_batch(reg, x0, tspan, opts, integrator=gr)
    on_nodes = grid != "rand"   # (error points = linspace: grid nodes only on an evenly spaced tspan)
    # (without the declaration: state pass, costate scan that reads the control samples, ControlChar on the grid, bookkeeping,
    #  enqueued one sweep ahead)
    assert ocs.fb_sweep_path(gf) == (4 if on_nodes else 5) and ocs.fb_sweep_path(gp) == (2 if on_nodes else 5)   # (5: error points off the nodes, sweeps enqueued ahead)
    assert np.array_equal(sf["sweeps"], sp["sweeps"]) and np.array_equal(sf["sweeps"], sr["sweeps"])
    ok = sf["sweeps"] > 0   # (with four states a few instances do not converge in 20 sweeps, on every path alike)
    assert ok.mean() > 0.75 and (nS == 4 or ok.all())
    for k, tol in (("J", 1e-11), ("x", 1e-10), ("lam", 1e-10), ("u", 1e-10)):
        assert relerr(sf[k][..., ok], sp[k][..., ok]) < tol and relerr(sf[k][..., ok], sr[k][..., ok]) < tol, k
    po = oracle.LogisticProblem(m, c, r, BOUNDS)
    for b in [int(i) for i in np.flatnonzero(ok)[[0, ok.sum() // 3, -1]]]:
        so = oracle.fb_sweep(po, x0[:, b], tspan, opts)
        assert sf["sweeps"][b] == so["_sweeps"] and abs(sf["J"][b] - so["J"]) < 1e-10 * abs(so["J"])
        assert relerr(sf["u"][:, :, b], so["u"]) < 1e-10 and relerr(sf["x"][:, :, b], so["x"]) < 1e-10
        assert relerr(sf["lam"][:, :, b], so["lam"]) < 1e-10


@pytest.mark.parametrize("nS,N,batch", [(2, 400, 64), (1, 96, 192), (4, 160, 48)])
def test_fb_sweep_row_functions_whose_adjoint_reads_the_control(ocs, oracle, nS, N, batch):
    """fb_sweep.m:79-115 for a problem given as row functions whose ControlChar reads x and whose dF/dy reads u (proportional
    harvest): the costate pass runs as the scan that reads the control samples (k_costate_scan<.., UR>), the sweeps are
    enqueued one ahead.  Against the kernel-by-kernel sequence of the same library on the lane kernels (fused_update_off = 1:
    lane-per-instance costate kernel, pchip midpoints and ControlChar in their own kernels, one host round trip per sweep)."""
    from tests.user_problems import PROPHARVEST_ROWS_CC_SRC
    c, r = 3.0, 0.05
    m = [3.0, 2.5, 2.0, 3.5][:nS]
    rng = np.random.default_rng(7 * nS + N)
    tspan = oracle.linspace(0, 5.0, N + 1)   # (grid helper only)
    x0 = rng.uniform(0.8, 1.6, (nS, batch))
    pu = ocs.UserProblem(PROPHARVEST_ROWS_CC_SRC, nS, 1, [c, r] + m, [[0.0, 1.0]], has_control_char=True, row_separable=True)
    # (the plain iteration u <- ControlChar oscillates on this problem: damped update, ocs.h uRelax)
    opts = {"nERROR_PTS": N + 1, "nINTERP_PTS": 61, "uRelax": 0.5, "nSWEEPS": 100}
    ga, gb = ocs.RK4Integrator(tspan), ocs.RK4Integrator(tspan)
    sa = ocs.fb_sweep_batch(pu, x0, tspan, opts, integrator=ga)
    sb = ocs.fb_sweep_batch(pu, x0, tspan, dict(opts, fused_update_off=1), integrator=gb)
    assert ocs.fb_sweep_path(ga) == 2 and ocs.fb_sweep_path(gb) == 1
    print("sweeps", np.unique(sa["sweeps"], return_counts=True), "max u", float(sa["u"].max()))
    assert np.array_equal(sa["sweeps"], sb["sweeps"]) and (sa["sweeps"] > 0).mean() > 0.9
    ok = sa["sweeps"] > 0
    for k, tol in (("J", 1e-11), ("x", 1e-10), ("lam", 1e-10), ("u", 1e-10)):
        assert relerr(sa[k][..., ok], sb[k][..., ok]) < tol, k
    assert np.all(sa["u"][..., ok].max(axis=(0, 1)) > 0.02)   # the control is not stuck at its lower bound


@pytest.mark.parametrize("N,batch,grid", [(200, 128, "lin"), (1000, 64, "lin"), (96, 70, "np"), (104, 192, "np")])
def test_fb_sweep_full_vector_plugin_on_the_vector_mappings(ocs, oracle, N, batch, grid):
    """fb_sweep.m:79-115 for a plugin given as the three full-vector methods + ocs_ControlChar (no row structure declared):
    vector-lane state pass with the frozen / gate arguments (k_forward_pv), costate pass as a scan with dense step maps
    (k_costate_vscan: pchip midpoints of x formed inside, control samples read), ControlChar on the grid, bookkeeping; sweeps
    enqueued ahead (path 2; a ragged batch runs the lane state pass and the lane costate pass inside the same loop).
    Per instance against the oracle and the registry problem of the same equations."""
    c, r, m = 1.5, 0.05, [3.0, 2.5]
    rng = np.random.default_rng(N + batch)
    tspan = oracle.linspace(0, 8.0, N + 1) if grid == "lin" else np.linspace(0, 8.0, N + 1)
    x0 = rng.uniform(0.8, 1.6, (2, batch))
    pu = ocs.UserProblem(LOGISTIC2_SRC, 2, 1, [c, r] + m, BOUNDS, has_control_char=True)
    reg = ocs.LogisticProblem(m, c, r, BOUNDS)
    opts = {"nERROR_PTS": N + 1, "nINTERP_PTS": 81}
    gu, gr = ocs.RK4Integrator(tspan), ocs.RK4Integrator(tspan)
    su = ocs.fb_sweep_batch(pu, x0, tspan, opts, integrator=gu)
    sr = ocs.fb_sweep_batch(reg, x0, tspan, opts, integrator=gr)
    assert ocs.fb_sweep_path(gu) == 2   # (ragged batches too since round 4: the lane kernels take the gate of a sweep enqueued ahead)
    assert np.array_equal(su["sweeps"], sr["sweeps"]) and np.all(su["sweeps"] > 0)
    for k, tol in (("J", 1e-11), ("x", 1e-10), ("lam", 1e-10), ("u", 1e-10)):
        assert relerr(su[k], sr[k]) < tol, k
    po = oracle.LogisticProblem(m, c, r, BOUNDS)
    for b in sorted({0, batch // 2, batch - 1}):
        so = oracle.fb_sweep(po, x0[:, b], tspan, opts)
        assert su["sweeps"][b] == so["_sweeps"] and abs(su["J"][b] - so["J"]) < 1e-10 * abs(so["J"])
        assert relerr(su["lam"][:, :, b], so["lam"]) < 1e-10 and relerr(su["u"][:, :, b], so["u"]) < 1e-10


# ---- plugins generated from symbols (optimal-control-solvers_amd/symbolic.py; functions/make_from_symbolic.m) ----------
def _sym_logistic(ocs, nS):
    import importlib

    import sympy as sp
    sym = importlib.import_module("ocs_amd.symbolic")
    names = ["c", "r"] + [f"m{k + 1}" for k in range(nS)]
    t, x, lam, u, p = sym.symbols(nS, 1, names)
    g = sp.exp(-p["r"] * t) * (sum(xi ** 2 for xi in x) + p["c"] * u[0] ** 2)
    f = [x[k] * (p[f"m{k + 1}"] - x[k]) - u[0] for k in range(nS)]
    vals = {"c": 1.5, "r": 0.05, **{f"m{k + 1}": [3.0, 2.5, 2.0, 1.5][k] for k in range(nS)}}
    return g, f, vals


@pytest.mark.parametrize("nS,rows", [(1, True), (2, True), (4, True), (2, False)])
def test_problem_generated_from_symbols_equals_registry_problem(ocs, oracle, nS, rows):
    """TestOCProblem / LogisticK from their two symbolic expressions (make_from_symbolic.m:1-38): the generated plugin
    -- as row functions with the costate-only declaration, and kept as full-vector methods -- against the registry
    problem to 1e-14 on the integrator passes, and through fb_sweep (the generated ControlChar with the clamp of :111)
    with the oracle's sweep counts; the row form runs the two-kernel sweep."""
    g, f, vals = _sym_logistic(ocs, nS)
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    ps = ocs.make_from_symbolic(g, f, nS, 1, vals, BOUNDS, allow_rows=rows)
    assert ps.generated["form"] == ("rows" if rows else "vector") and ps.generated["has_control_char"]
    pb, po = ocs.LogisticProblem(m, 1.5, 0.05, BOUNDS), oracle.LogisticProblem(m, 1.5, 0.05, BOUNDS)
    rng = np.random.default_rng(nS)
    k = 9
    t, y = rng.uniform(0, 10, k), rng.normal(1.5, 0.5, (nS + 1, k))
    u, v = rng.uniform(0, 1, (1, k)), rng.normal(size=(nS + 1, k))
    assert relerr(ps.F(t, y, u), po.F(t, y, u)) < 1e-14
    assert relerr(ps.dFdx_times_vec(t, y, u, v), po.dFdx_times_vec(t, y, u, v)) < 1e-14
    assert relerr(ps.dFdu_times_vec(t, y, u, v), po.dFdu_times_vec(t, y, u, v)) < 1e-14
    N, batch = 200, 128
    tspan = oracle.linspace(0, 10, N + 1)
    uu, x0 = rng.uniform(0.05, 0.45, (1, 2 * N + 1, batch)), rng.uniform(0.9, 2.0, (nS, batch))
    gs, gb = ocs.RK4Integrator(tspan), ocs.RK4Integrator(tspan)
    xs, Js = gs.compute_states(ps, x0, uu)
    ls, ds = gs.compute_adjoints(ps, uu)
    xb, Jb = gb.compute_states(pb, x0, uu)
    lb, db = gb.compute_adjoints(pb, uu)
    assert relerr(xs, xb) < 1e-13 and relerr(Js, Jb) < 1e-13 and relerr(ls, lb) < 1e-13 and relerr(ds, db) < 1e-13
    ts = oracle.linspace(0, 8, 161)
    opt = {"nERROR_PTS": 161, "nINTERP_PTS": 81}
    X0 = rng.uniform(0.9, 2.0, (nS, 64))
    s1 = ocs.fb_sweep_batch(ps, X0, ts, opt, integrator=(gi := ocs.RK4Integrator(ts)))
    s2 = ocs.fb_sweep_batch(pb, X0, ts, opt)
    ok = s1["sweeps"] > 0   # (with four states the undamped sweep of fb_sweep.m:79-87 does not converge: on every path alike)
    assert np.array_equal(s1["sweeps"], s2["sweeps"]) and (nS == 4 or ok.all())
    for k in ("J", "u", "lam"):
        assert not ok.any() or relerr(s1[k][..., ok], s2[k][..., ok]) < 1e-10
    so = oracle.fb_sweep(po, X0[:, 3], ts, opt)
    assert s1["sweeps"][3] == max(so["_sweeps"], 0) and (not ok[3] or abs(s1["J"][3] - so["J"]) < 1e-10 * abs(so["J"]))
    if rows:
        assert ocs.fb_sweep_path(gi) == 4   # row functions + ControlChar of the costate alone: the two-kernel sweep


def test_predator_prey_generated_from_symbols_equals_hand_written_plugin(ocs, oracle):
    import importlib

    import sympy as sp
    sym = importlib.import_module("ocs_amd.symbolic")
    names = ["al", "be", "de", "ga", "c", "q", "xb", "r"]
    t, x, lam, u, p = sym.symbols(2, 1, names)
    g = sp.exp(-p["r"] * t) * (p["c"] * u[0] ** 2 + p["q"] * (x[0] - p["xb"]) ** 2)
    f = [x[0] * (p["al"] - p["be"] * x[1]), x[1] * (p["de"] * x[0] - p["ga"]) - u[0] * x[1]]
    ps = ocs.make_from_symbolic(g, f, 2, 1, dict(zip(names, PREDPREY_PARAMS)), BOUNDS)
    ph = ocs.UserProblem(PREDPREY_SRC, 2, 1, PREDPREY_PARAMS, BOUNDS)
    assert ps.generated["form"] == "vector" and ps.generated["tcoef"] is not None
    rng = np.random.default_rng(3)
    N, batch = 120, 128
    tspan = oracle.linspace(0, 6, N + 1)
    uu, x0 = rng.uniform(0.0, 0.6, (1, 2 * N + 1, batch)), rng.uniform(0.8, 1.6, (2, batch))
    gs, gh = ocs.RK4Integrator(tspan), ocs.RK4Integrator(tspan)
    xs, Js = gs.compute_states(ps, x0, uu)
    ls, ds = gs.compute_adjoints(ps, uu)
    xh, Jh = gh.compute_states(ph, x0, uu)
    lh, dh = gh.compute_adjoints(ph, uu)
    assert relerr(xs, xh) < 1e-13 and relerr(Js, Jh) < 1e-13 and relerr(ls, lh) < 1e-13 and relerr(ds, dh) < 1e-13
    # and against the NumPy twin of the hand-written plugin on one trajectory
    tw_i = tw.RK4IntegratorNP(tspan)
    xo, Jo = tw_i.compute_states(PredPreyNP(), x0[:, 5], uu[:, :, 5])
    assert relerr(xs[:, :, 5], xo) < RTOL and abs(Js[5] - Jo) < RTOL * max(1, abs(Jo))
    # fb_sweep: the generated ControlChar (reads x: kernel-by-kernel / vector sweep) converges to a control that satisfies
    # its own definition
    ts = oracle.linspace(0, 4, 201)
    r = ocs.fb_sweep_batch(ps, rng.uniform(0.8, 1.6, (2, 64)), ts, {"nERROR_PTS": 201, "nINTERP_PTS": 201, "nSWEEPS": 100})
    ok = r["sweeps"] > 0
    assert ok.any()
    g1 = ps.gen1
    b = int(np.flatnonzero(ok)[0])
    uc = g1.ControlChar(ts, r["x"][:, :, b], r["lam"][:, :, b])
    assert relerr(uc, r["u"][:, :, b]) < 1e-9


def test_coupled_six_state_three_control_plugin(ocs, oracle):
    """OCProblem.m:8-21 puts no limit on the shapes: a coupled plugin with nS = 6, nC = 3 (generated from symbols,
    tests/user_problems.ring6_symbolic) runs the lane kernels -- integrator passes against the NumPy twin to 1e-12 -- and
    fb_sweep runs it with the fused control update and sweeps enqueued ahead (path 2: every state pass takes the gate now),
    equal to the kernel-by-kernel sequence with a host round trip per sweep (path 1) in sweep counts and to round-off."""
    import importlib
    from tests.user_problems import ring6_symbolic
    sym = importlib.import_module("ocs_amd.symbolic")
    g, f, vals = ring6_symbolic(sym)
    bounds = [[0.0, 1.0]] * 3
    prob = ocs.make_from_symbolic(g, f, 6, 3, vals, bounds)
    assert prob.generated["form"] == "vector" and prob.generated["has_control_char"] and not prob.generated["control_from_costate"]
    rng = np.random.default_rng(66)
    N, batch = 160, 130
    tspan = oracle.linspace(0, 4, N + 1)
    u, x0 = rng.uniform(0.0, 0.8, (3, 2 * N + 1, batch)), rng.uniform(0.6, 1.8, (6, batch))
    gi = ocs.RK4Integrator(tspan)
    x, J = gi.compute_states(prob, x0, u)
    lam, dJdu = gi.compute_adjoints(prob, u)
    twin, ti = prob.numpy_twin, tw.RK4IntegratorNP(tspan)
    for b in (0, 64, batch - 1):
        xo, Jo = ti.compute_states(twin, x0[:, b], u[:, :, b])
        lamo, do = ti.compute_adjoints(twin, u[:, :, b])
        assert relerr(x[:, :, b], xo) < RTOL and abs(J[b] - Jo) < RTOL * max(1, abs(Jo))
        assert relerr(lam[:, :, b], lamo) < RTOL and relerr(dJdu[:, :, b], do) < RTOL
    # the undamped iteration of fb_sweep.m:79-87 does not converge on this problem over [0, 4] (on neither path); the damped
    # update (uRelax, an extension that is off by default) does.  Both loops: the same iteration, sweep by sweep.
    opt = {"nERROR_PTS": N + 1, "nINTERP_PTS": 41, "nSWEEPS": 120, "uRelax": 0.35}
    X0 = rng.uniform(0.6, 1.8, (6, 96))
    g2, g1 = ocs.RK4Integrator(tspan), ocs.RK4Integrator(tspan)
    s2 = ocs.fb_sweep_batch(prob, X0, tspan, opt, integrator=g2)
    s1 = ocs.fb_sweep_batch(prob, X0, tspan, dict(opt, fused_update_off=1), integrator=g1)
    assert ocs.fb_sweep_path(g2) == 2 and ocs.fb_sweep_path(g1) == 1
    ok = s2["sweeps"] > 0
    assert np.array_equal(s2["sweeps"], s1["sweeps"])
    mc2, mc1 = s2["maxChange"][:8], s1["maxChange"][:8]     # the "Normalized change in u" of :109, first sweeps, every instance
    assert np.all(np.isfinite(mc2)) and np.allclose(mc2, mc1, rtol=1e-7, atol=0)
    assert ok.mean() > 0.5, ok.mean()
    for k in ("J", "x", "lam", "u"):
        assert relerr(s2[k][..., ok], s1[k][..., ok]) < 1e-9, k
    # the converged control satisfies its own definition (ControlChar of the returned x, lam with the clamp of :111)
    b = int(np.flatnonzero(ok)[0])
    tq = oracle.linspace(0, 4, 41)
    from scipy.interpolate import PchipInterpolator
    xq = np.vstack([PchipInterpolator(tspan, s2["x"][r, :, b])(tq) for r in range(6)])
    lq = np.vstack([PchipInterpolator(tspan, s2["lam"][r, :, b])(tq) for r in range(6)])
    assert relerr(prob.gen1.ControlChar(tq, xq, lq), s2["u"][:, :, b]) < 1e-5   # (to the sweep's tolerance: the damped fixed point)


def test_six_state_plugin_control_at_points(ocs, oracle):
    """fb_sweep.m:123 on the six-state plugin: uOpt = ControlChar(interpPts, x(interpPts), lam(interpPts)) with pchip x, lam.  State
    vectors beyond four rows walk their points with one sliding window per row where the points sit in consecutive intervals
    (a grid whose nodes are shifted to the right of the uniform ones, nINTERP_PTS = N + 1) and evaluate them one by one
    otherwise (41 points); error points off the nodes (nERROR_PTS = N + 1 on that grid) take the same kernel inside the
    loop.  Both against SciPy's pchip of the returned x, lam and the generated NumPy ControlChar."""
    import importlib
    from scipy.interpolate import PchipInterpolator
    from tests.user_problems import ring6_symbolic
    sym = importlib.import_module("ocs_amd.symbolic")
    g, f, vals = ring6_symbolic(sym)
    prob = ocs.make_from_symbolic(g, f, 6, 3, vals, [[0.0, 1.0]] * 3)
    N, T = 96, 2.0
    i = np.arange(N + 1)
    tspan = i * (T / N) + 0.4 * (T / N) * np.sin(np.pi * i / N) ** 2     # t_i in (i h, (i + 1) h) inside, ends on the ends
    X0 = np.random.default_rng(67).uniform(0.6, 1.8, (6, 70))
    for nI in (N + 1, 41):
        s = ocs.fb_sweep_batch(prob, X0, tspan, {"nERROR_PTS": N + 1, "nINTERP_PTS": nI, "nSWEEPS": 4, "uRelax": 0.35})
        tq = oracle.linspace(0, T, nI)
        for b in (0, 69):
            xq = np.vstack([PchipInterpolator(tspan, s["x"][r, :, b])(tq) for r in range(6)])
            lq = np.vstack([PchipInterpolator(tspan, s["lam"][r, :, b])(tq) for r in range(6)])
            assert relerr(prob.gen1.ControlChar(tq, xq, lq), s["u"][:, :, b]) < 1e-10, (nI, b)


def test_control_char_method(ocs, oracle):
    """ControlChar(t, x, lam) as a problem method (make_from_symbolic.m:33-38 with the clamp of :111; fb_sweep.m:96,123 call
    it): registry problems against the oracle, the LQ problem through its plugin twin, a generated plugin against the NumPy
    function SymPy writes, a plugin without ocs_ControlChar refused."""
    import importlib
    from tests.user_problems import ring6_symbolic
    rng = np.random.default_rng(71)
    k = 37
    t = rng.uniform(0, 8, k)
    cases = [(ocs.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]]), oracle.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]])),
             (ocs.LogisticProblem([3.0, 2.5, 2.0, 2.8], 1.5, 0.05, [[0.0, 1.0]]), oracle.LogisticProblem([3.0, 2.5, 2.0, 2.8], 1.5, 0.05, [[0.0, 1.0]]))]
    nS, nC = 5, 2
    A = -np.eye(nS) + 0.1 * rng.normal(size=(nS, nS))
    Bu, q, rd = rng.normal(size=(nS, nC)), rng.uniform(0.5, 1.5, nS), rng.uniform(1, 2, nC)
    cases.append((ocs.LQProblem(A, Bu, q, rd, 0.05, [[-1.0, 1.0]] * nC), oracle.LQProblem(A, Bu, q, rd, 0.05, [[-1.0, 1.0]] * nC)))
    for gp, op in cases:
        x, lam = rng.uniform(0.2, 2.0, (gp.nS, k)), rng.normal(size=(gp.nS, k)) * 0.8
        got, want = gp.ControlChar(t, x, lam), op.ControlChar(t, x, lam)
        assert got.shape == (gp.nC, k) and relerr(got, want) < 1e-13
        lo = -1.0 if gp.nC == 2 else 0.0
        assert (got == lo).any() or (got == 1.0).any()     # the clamp is active somewhere
    sym = importlib.import_module("ocs_amd.symbolic")
    g, f, vals = ring6_symbolic(sym)
    prob = ocs.make_from_symbolic(g, f, 6, 3, vals, [[0.0, 1.0]] * 3)
    x, lam = rng.uniform(0.2, 2.0, (6, k)), rng.normal(size=(6, k))
    assert relerr(prob.ControlChar(t, x, lam), prob.gen1.ControlChar(t, x, lam)) < 1e-13
    with pytest.raises(ocs.OcsError, match="no ocs_ControlChar"):
        ocs.UserProblem(PREDPREY_SRC, 2, 1, PREDPREY_PARAMS, BOUNDS).ControlChar(t, x[:2], lam[:2])


def test_false_declarations_are_refused(ocs):
    """ocs_problem_create_from_source probes the dFdy half of flag bit 2 (control from the costate alone): a plugin whose
    (dF/dy)'v reads u cannot claim it.  Per-trajectory parameters are refused for plugins that tabulate a time coefficient
    from the shared parameter block (ocs_tcoef / ocs_cc_tcoef)."""
    from tests.user_problems import PREDPREY_TC_SRC, PROPHARVEST_ROWS_CC_SRC
    kw = dict(has_control_char=True, row_separable=True)
    ocs.UserProblem(PROPHARVEST_ROWS_CC_SRC, 2, 1, [1.5, 0.05, 3.0, 2.5], BOUNDS, **kw)           # without the claim: fine
    with pytest.raises(ocs.OcsError, match="dFdy does not read u"):
        ocs.UserProblem(PROPHARVEST_ROWS_CC_SRC, 2, 1, [1.5, 0.05, 3.0, 2.5], BOUNDS, control_from_costate=True, **kw)
    # ... and the ControlChar half: a ControlChar that reads x (here: scaled by x(1)) cannot claim it either
    from tests.user_problems import LOGISTIC_ROWS_CC_SRC
    cc_reads_x = LOGISTIC_ROWS_CC_SRC.replace("s * exp(p[1] * t) / (2 * p[0])", "x[0] * s * exp(p[1] * t) / (2 * p[0])")
    assert cc_reads_x != LOGISTIC_ROWS_CC_SRC
    ocs.UserProblem(cc_reads_x, 2, 1, [1.5, 0.05, 3.0, 2.5], BOUNDS, **kw)                         # without the claim: fine
    with pytest.raises(ocs.OcsError, match="ControlChar does not read x"):
        ocs.UserProblem(cc_reads_x, 2, 1, [1.5, 0.05, 3.0, 2.5], BOUNDS, control_from_costate=True, **kw)
    ocs.UserProblem(LOGISTIC_ROWS_CC_SRC, 2, 1, [1.5, 0.05, 3.0, 2.5], BOUNDS, control_from_costate=True, **kw)   # a true claim
    pt = ocs.UserProblem(PREDPREY_TC_SRC, 2, 1, PREDPREY_PARAMS, BOUNDS)
    with pytest.raises(ocs.OcsError, match="tabulated"):
        pt.set_batch_params([4], np.full((1, 8), 2.0))
    pn = ocs.UserProblem(PREDPREY_SRC, 2, 1, PREDPREY_PARAMS, BOUNDS)
    pn.set_batch_params([4], np.full((1, 8), 2.0))                                                 # no tabulated coefficient: allowed


def test_reference_symbolic_scripts_on_the_device(ocs, oracle):
    """The problems of the two symbolic scripts the reference ships (tests/symbolic_test2.m: make_from_symbolic(x^2 + c u^2,
    x (m - x) - u, 1, 1, {m .5, c 4}, [0 1]), with its commented solve on [0, 5] from x0 = .1; tests/symbolic_test.m: two states,
    two controls, f = [x1 x2 - u1; u2 x2 + 3]), generated from the same expressions.  The first is TestOCProblem with r = 0:
    its sweep equals the registry problem's and the oracle's.  The second runs the vector mappings: passes against the NumPy
    twin, and the returned control satisfies its own definition u1 = lam1 / 2, u2 = -lam2 x2 / 2."""
    import importlib
    from scipy.interpolate import PchipInterpolator
    from tests.test_symbolic import reference_symbolic_test, reference_symbolic_test2
    sym = importlib.import_module("ocs_amd.symbolic")
    g, f, vals, bounds = reference_symbolic_test2(sym)
    ps = ocs.make_from_symbolic(g, f, 1, 1, vals, bounds)
    pr, po = ocs.TestOCProblem({"c": 4.0, "m": 0.5, "r": 0.0}, bounds), oracle.TestOCProblem({"c": 4.0, "m": 0.5, "r": 0.0}, bounds)
    X0 = np.array([[0.1, 0.3, 0.2, 0.45] * 16])
    opt = {"nERROR_PTS": 201, "nINTERP_PTS": 101}
    # symbolic_test2.m:12 solves [0, 5] from x0 = .1 with bvp_solver; fb_sweep.m:79-87 does not settle on that horizon (the
    # iterates drive x below 0, where x (m - x) runs away): generated plugin, registry problem and oracle report that alike.
    # On [0, 2] it converges: same sweep counts, same solution.
    for T, conv in ((5.0, False), (2.0, True)):
        ts = oracle.linspace(0, T, 201)
        s1, s2 = ocs.fb_sweep_batch(ps, X0, ts, opt, integrator=(gi := ocs.RK4Integrator(ts))), ocs.fb_sweep_batch(pr, X0, ts, opt)
        assert ocs.fb_sweep_path(gi) == 4 and np.array_equal(s1["sweeps"], s2["sweeps"])
        ok = s1["sweeps"] > 0
        assert ok.all() == conv and (conv or not ok[0])
        for b in (0, 1):
            so = oracle.fb_sweep(po, X0[:, b], ts, opt)
            assert s1["sweeps"][b] == max(so["_sweeps"], 0)
            if ok[b]:
                assert abs(s1["J"][b] - so["J"]) < 1e-11 * max(1.0, abs(so["J"]))
                assert relerr(s1["u"][:, :, b], so["u"]) < 1e-9 and relerr(s1["x"][:, :, b], so["x"]) < 1e-11
        for k in ("J", "x", "lam", "u"):
            assert not ok.any() or relerr(s1[k][..., ok], s2[k][..., ok]) < 1e-11, k

    g, f, vals, bounds = reference_symbolic_test(sym)
    p2 = ocs.make_from_symbolic(g, f, 2, 2, vals, bounds)
    rng = np.random.default_rng(72)
    N, batch = 160, 192
    tspan = oracle.linspace(0, 0.8, N + 1)
    u, x0 = rng.uniform(-0.5, 0.5, (2, 2 * N + 1, batch)), rng.uniform(0.2, 0.8, (2, batch))
    gi = ocs.RK4Integrator(tspan)
    x, J = gi.compute_states(p2, x0, u)
    lam, dJdu = gi.compute_adjoints(p2, u)
    twin, ti = p2.numpy_twin, tw.RK4IntegratorNP(tspan)
    for b in (0, 100, batch - 1):
        xo, Jo = ti.compute_states(twin, x0[:, b], u[:, :, b])
        lamo, do = ti.compute_adjoints(twin, u[:, :, b])
        assert relerr(x[:, :, b], xo) < RTOL and abs(J[b] - Jo) < RTOL * max(1, abs(Jo))
        assert relerr(lam[:, :, b], lamo) < RTOL and relerr(dJdu[:, :, b], do) < RTOL
    s = ocs.fb_sweep_batch(p2, rng.uniform(0.2, 0.6, (2, 64)), tspan, {"nERROR_PTS": N + 1, "nINTERP_PTS": 41, "uRelax": 0.5, "nSWEEPS": 200})
    ok = s["sweeps"] > 0
    assert ok.mean() > 0.5, ok.mean()
    b = int(np.flatnonzero(ok)[0])
    tq = oracle.linspace(0, 0.8, 41)
    xq = np.vstack([PchipInterpolator(tspan, s["x"][r, :, b])(tq) for r in range(2)])
    lq = np.vstack([PchipInterpolator(tspan, s["lam"][r, :, b])(tq) for r in range(2)])
    assert relerr(s["u"][:, :, b], np.vstack([lq[0] / 2, -lq[1] * xq[1] / 2])) < 1e-10     # symbolic_test.m:26-29 uOpt
