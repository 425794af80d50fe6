"""GPU parity of the built-in linear-quadratic problem (BASELINE config 5) on the matrix-core kernels
(csrc/ocs_lq_kernels.hip) against the CPU oracle: RK4Integrator.m:28-121 and RK4InfiniteIntegrator.m:12-30
semantics with F = [A x + Bu u ; e^{-rt}(x'Qx + u'Ru)].  The oracle is the build's restatement (parity
unpinned against MATLAB, see oracle/README); tolerance 1e-12 relative (fp64, summation order of the
16x16x4 matrix instruction differs from the oracle's row loops)."""
import numpy as np
import pytest

from tests.user_problems import lq_matrices

pytestmark = pytest.mark.gpu
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


def make(ocs, oracle, nS, nC, seed=20260405):
    A, Bu, q, rdiag = lq_matrices(nS, nC, seed)
    bounds = [[-1.0, 1.0]] * nC
    return (ocs.LQProblem(A, Bu, q, rdiag, 0.05, bounds), oracle.LQProblem(A, Bu, q, rdiag, 0.05, bounds))


@pytest.mark.parametrize("nS,nC", [(1, 1), (5, 2), (16, 4), (20, 3), (32, 4), (32, 1)])
def test_plugin_methods(ocs, oracle, nS, nC):
    pg, po = make(ocs, oracle, nS, nC)
    rng = np.random.default_rng(nS * 10 + nC)
    k = 11
    t, y = rng.uniform(0, 3, k), rng.normal(size=(nS + 1, k))
    u, v = rng.uniform(-1, 1, (nC, k)), rng.normal(size=(nS + 1, k))
    assert relerr(pg.F(t, y, u), po.F(t, y, u)) < 1e-13
    assert relerr(pg.dFdx_times_vec(t, y, u, v), po.dFdx_times_vec(t, y, u, v)) < 1e-13
    assert relerr(pg.dFdu_times_vec(t, y, u, v), po.dFdu_times_vec(t, y, u, v)) < 1e-13


@pytest.mark.parametrize("mapping", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("nS,nC,N,batch", [(1, 1, 7, 3), (5, 2, 33, 37), (16, 4, 64, 16), (17, 2, 20, 19), (20, 3, 50, 50),
                                           (32, 4, 96, 68), (32, 1, 3, 1)])
def test_states_adjoints_vs_oracle(ocs, oracle, nS, nC, N, batch, mapping):
    pg, po = make(ocs, oracle, nS, nC)
    rng = np.random.default_rng(N)
    tspan = np.concatenate([[0.0], np.sort(rng.uniform(0.0, 1.5, N - 1)), [1.5]]) if N > 3 else oracle.linspace(0, 0.1, N + 1)
    u = rng.uniform(-1, 1, (nC, 2 * N + 1, batch))
    x0 = rng.normal(size=(nS, batch))
    lamT = rng.normal(size=(nS + 1, batch))
    g, go = ocs.RK4Integrator(tspan), oracle.RK4Integrator(tspan)
    g.set_mapping(mapping)   # 0: automatic, 1: one wave per 16 trajectories, 2: two, 3: four, 4: time-parallel chunks
    x, J = g.compute_states(pg, x0, u)
    lam, dJdu = g.compute_adjoints(pg, u)
    for b in sorted({0, batch // 2, batch - 1}):
        xo, Jo = go.compute_states(po, x0[:, b], u[:, :, b])
        lamo, do = go.compute_adjoints(po, u[:, :, b])
        assert relerr(x[:, :, b], xo) < RTOL and abs(J[b] - Jo) < RTOL * max(1, abs(Jo))
        assert relerr(lam[:, :, b], lamo) < RTOL and relerr(dJdu[:, :, b], do) < RTOL
        assert np.all(lam[-1, :, b] == 1.0)
    # user-supplied terminal multiplier (RK4Integrator.m:63-69)
    g.compute_states(pg, x0, u)
    lam2, d2 = g.compute_adjoints(pg, u, lamT)
    for b in sorted({0, batch - 1}):
        go.compute_states(po, x0[:, b], u[:, :, b])
        lamo, do = go.compute_adjoints(po, u[:, :, b], lamT[:, b])
        assert relerr(lam2[:, :, b], lamo) < RTOL and relerr(d2[:, :, b], do) < RTOL


@pytest.mark.parametrize("mapping", [0, 1, 2, 3, 4])
def test_infinite_horizon_and_shooting_objective(ocs, oracle, mapping):
    """BL-5 shape at reduced size: nS = 32, nC = 4, RK4InfiniteIntegrator with uStar = 0, then the shooting objective
    (single_shooting.m:137-150) through a PWLinear basis with nC = 4 and a free initial state."""
    nS, nC, N, batch = 32, 4, 80, 40
    pg, po = make(ocs, oracle, nS, nC)
    tspan, tx = oracle.linspace(0, 1, N + 1), oracle.linspace(1, 2, N // 2 + 1)
    rng = np.random.default_rng(5)
    u = rng.uniform(-1, 1, (nC, 2 * N + 1, batch))
    x0 = rng.normal(size=(nS, batch))
    ustar = np.array([0.1, -0.2, 0.0, 0.3])
    gi, go = ocs.RK4InfiniteIntegrator(tspan, tx, ustar), oracle.RK4InfiniteIntegrator(tspan, tx, ustar)
    gi.set_mapping(mapping)
    x, J = gi.compute_states(pg, x0, u)
    lam, dJdu = gi.compute_adjoints(pg, u)
    for b in (0, 17, 39):
        xo, Jo = go.compute_states(po, x0[:, b], u[:, :, b])
        lamo, do = go.compute_adjoints(po, u[:, :, b])
        assert relerr(x[:, :, b], xo) < RTOL and abs(J[b] - Jo) < RTOL * max(1, abs(Jo))
        assert relerr(lam[:, :, b], lamo) < RTOL and relerr(dJdu[:, :, b], do) < RTOL
    cg, co = ocs.PWLinearControl(gi.t, 9, nC), oracle.PWLinearControl(go.t, 9, nC)
    V = rng.uniform(-1, 1, (nC * 9 + 2, 5))
    free = [3, 30]
    Jn, dJdv, _ = ocs.nlp_objective(gi, pg, cg, x0[:, :5], V, free)
    for b in range(5):
        Jo, do, _ = oracle.nlp_objective(go, po, co, x0[:, b], V[:, b], free)
        assert abs(Jn[b] - Jo) < RTOL * max(1, abs(Jo)) and relerr(dJdv[:, b], do) < RTOL


@pytest.mark.parametrize("nS,nC,N,N2,batch", [(32, 4, 517, 300, 40), (12, 2, 256, 130, 100)])
def test_time_parallel_chunks_selected_automatically(ocs, oracle, nS, nC, N, N2, batch):
    """A small batch on a long horizon runs the chunked passes without being asked (csrc/ocs_lq_kernels.hip, "chunked
    passes": 8 - 16 chunks here, the last one ragged, non-uniform grid): RK4InfiniteIntegrator against the oracle, and
    against the serial one-wave mapping -- equal to round-off but not bit for bit (other kernels ran), J == x(end,end)
    of the leg-1 sum as in every mapping, lam(end,:) == 1, lam-only / dJdu-only / J-only calls."""
    pg, po = make(ocs, oracle, nS, nC)
    rng = np.random.default_rng(N)
    tspan = np.concatenate([[0.0], np.sort(rng.uniform(0.0, 2.0, N - 1)), [2.0]])
    tx = oracle.linspace(2, 3, N2 + 1)
    u = rng.uniform(-1, 1, (nC, 2 * N + 1, batch))
    x0 = rng.normal(size=(nS, batch))
    ustar = rng.uniform(-0.3, 0.3, nC)
    gi, gs, go = (ocs.RK4InfiniteIntegrator(tspan, tx, ustar), ocs.RK4InfiniteIntegrator(tspan, tx, ustar).set_mapping(1),
                  oracle.RK4InfiniteIntegrator(tspan, tx, ustar))
    x, J = gi.compute_states(pg, x0, u)
    lam, dJdu = gi.compute_adjoints(pg, u)
    xs, Js = gs.compute_states(pg, x0, u)
    lams, ds = gs.compute_adjoints(pg, u)
    assert relerr(x, xs) < RTOL and relerr(J, Js) < RTOL and relerr(lam, lams) < RTOL and relerr(dJdu, ds) < RTOL
    assert not np.array_equal(x, xs)   # the chunked kernels ran, not the serial ones
    assert np.all(lam[-1] == 1.0) and np.all(np.isfinite(dJdu))
    for b in (0, batch // 2, batch - 1):
        xo, Jo = go.compute_states(po, x0[:, b], u[:, :, b])
        lamo, do = go.compute_adjoints(po, u[:, :, b])
        assert relerr(x[:, :, b], xo) < RTOL and abs(J[b] - Jo) < RTOL * max(1, abs(Jo))
        assert relerr(lam[:, :, b], lamo) < RTOL and relerr(dJdu[:, :, b], do) < RTOL
    # plain RK4Integrator, explicit lamT, partial outputs
    g1, go1 = ocs.RK4Integrator(tspan), oracle.RK4Integrator(tspan)
    lamT = rng.normal(size=(nS + 1, batch))
    x1, J1 = g1.compute_states(pg, x0, u)
    assert np.array_equal(J1, x1[-1, -1, :])
    l1, d1 = g1.compute_adjoints(pg, u, lamT)
    go1.compute_states(po, x0[:, 1], u[:, :, 1])
    lamo, do = go1.compute_adjoints(po, u[:, :, 1], lamT[:, 1])
    assert relerr(l1[:, :, 1], lamo) < RTOL and relerr(d1[:, :, 1], do) < RTOL
    l0, d0 = g1.compute_adjoints(pg, u)
    import torch
    dev = torch.device("cuda:0")
    ud = torch.tensor(np.ascontiguousarray(u.transpose(1, 0, 2)), device=dev)
    x0d = torch.tensor(x0, device=dev)
    _, Jd = g1.compute_states_dev(pg, x0d, ud)           # J only (checkpoints in the handle)
    dd = torch.empty_like(ud)
    g1.compute_adjoints_dev(pg, ud, None, None, dd)      # dJdu only
    ld = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev)
    g1.compute_adjoints_dev(pg, ud, None, ld, None)      # lam only
    torch.cuda.synchronize()
    assert relerr(Jd.cpu().numpy(), J1) < RTOL
    assert relerr(dd.cpu().numpy().transpose(1, 0, 2), d0) < RTOL and relerr(ld.cpu().numpy().transpose(1, 0, 2), l0) < RTOL


def test_matches_user_plugin_path(ocs):
    """The same problem given as plugin source (hipRTC, lane-per-trajectory VALU kernels) and through the registry
    (matrix cores) agree to round-off: two independent device implementations of the same recursion."""
    from tests.user_problems import lq_source
    nS, nC, N, batch = 12, 2, 40, 33
    A, Bu, q, rdiag = lq_matrices(nS, nC)
    bounds = [[-1.0, 1.0]] * nC
    par = np.concatenate([[0.05], A.ravel(order="F"), Bu.ravel(order="F"), q, rdiag])
    pu = ocs.UserProblem(lq_source(nS, nC), nS, nC, par, bounds)
    pg = ocs.LQProblem(A, Bu, q, rdiag, 0.05, bounds)
    rng = np.random.default_rng(9)
    tspan = np.linspace(0, 1, N + 1)
    u, x0 = rng.uniform(-1, 1, (nC, 2 * N + 1, batch)), rng.normal(size=(nS, batch))
    g = ocs.RK4Integrator(tspan)
    xa, Ja = g.compute_states(pu, x0, u)
    la, da = g.compute_adjoints(pu, u)
    xb, Jb = g.compute_states(pg, x0, u)
    lb, db = g.compute_adjoints(pg, u)
    assert relerr(xa, xb) < RTOL and relerr(Ja, Jb) < RTOL and relerr(la, lb) < RTOL and relerr(da, db) < RTOL


def test_fb_sweep_on_the_lq_problem(ocs, oracle):
    """fb_sweep.m:79-125 on OCS_PROBLEM_LQ (the registry problem is handed to the sweep kernels as generated plugin
    source, with the Gen-1 ControlChar u = clamp(-Bu' lam e^{rt} / (2 R))): single instance against the oracle, and a
    small batch of initial states; compute_x_lam on the same path."""
    nS, nC, N = 5, 2, 400
    pg, po = make(ocs, oracle, nS, nC)
    tspan = oracle.linspace(0, 2.0, N + 1)
    x0 = np.linspace(0.5, 1.5, nS)
    opt = {"nERROR_PTS": 401, "nINTERP_PTS": 101, "nSWEEPS": 200}
    ref = oracle.fb_sweep(po, x0, tspan, opt)
    assert ref["_sweeps"] > 0
    soln = ocs.fb_sweep(pg, x0, tspan, opt)
    assert set(soln) == {"x", "lam", "u", "J"}
    assert abs(soln["J"] - ref["J"]) < 1e-10 * abs(ref["J"])
    assert relerr(soln["u"](ref["_interpPts"]), ref["u"]) < 1e-10
    assert relerr(soln["x"](tspan), ref["x"]) < 1e-10 and relerr(soln["lam"](tspan), ref["lam"]) < 1e-10
    # a batch of instances differing in x0: instance b equals the single-instance solve
    X0 = np.stack([x0, 0.5 * x0, -x0], axis=1)
    sb = ocs.fb_sweep_batch(pg, X0, tspan, opt)
    assert sb["sweeps"][0] == ref["_sweeps"] and abs(sb["J"][0] - ref["J"]) < 1e-10 * abs(ref["J"])
    r2 = oracle.fb_sweep(po, -x0, tspan, opt)
    assert sb["sweeps"][2] == r2["_sweeps"] and abs(sb["J"][2] - r2["J"]) < 1e-10 * abs(r2["J"])
    # the quadratic objective is even in x0 (u -> -u, bounds symmetric): J(x0) == J(-x0) to round-off
    assert abs(sb["J"][0] - sb["J"][2]) < 1e-10 * abs(ref["J"])


def test_unsupported_shapes_fail_loudly(ocs):
    A, Bu, q, rdiag = lq_matrices(33, 2)
    with pytest.raises(Exception):
        ocs.LQProblem(A, Bu, q, rdiag, 0.05, [[-1, 1]] * 2)   # nS > 32: no kernel instantiated
    A, Bu, q, rdiag = lq_matrices(4, 5)
    with pytest.raises(Exception):
        ocs.LQProblem(A, Bu, q, rdiag, 0.05, [[-1, 1]] * 5)   # nC > 4


@pytest.mark.parametrize("batch", [8192, 1024])
def test_bl5_full_size_properties(ocs, batch):
    """BASELINE config 5 at full size (nS = 32, nC = 4, N = 4000 + 4000 tail steps, batch 8192; and its 8-GPU shard of
    1024 trajectories, which runs the time-parallel chunked passes) through properties
    that need no oracle: the state rows are linear in (x0, u); J is quadratic in u, so a central difference of J
    along a direction d equals <dJdu, d> exactly (up to round-off) -- which ties the adjoint pass, the tail leg's
    lamT hand-off (RK4InfiniteIntegrator.m:27-30) and compute_dJdu to the forward pass."""
    import torch
    dev = torch.device("cuda:0")
    nS, nC, N, T = 32, 4, 4000, 10.0
    rng = np.random.default_rng(20260405)
    A = -np.diag(np.logspace(0, 3, nS)) + 0.1 * rng.normal(size=(nS, nS))
    Bu = rng.normal(size=(nS, nC))
    prob = ocs.LQProblem(A, Bu, rng.uniform(0.5, 1.5, nS), rng.uniform(1, 2, nC), 0.05, [[-1.0, 1.0]] * nC)
    gi = ocs.RK4InfiniteIntegrator(np.linspace(0, T, N + 1), np.linspace(T, 2 * T, N + 1), np.zeros(nC))
    third = batch // 3
    gen = torch.Generator(device=dev).manual_seed(7)
    ua = torch.rand((2 * N + 1, nC, third), dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    d = torch.rand((2 * N + 1, nC, third), dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    xa = torch.randn((nS, third), dtype=torch.float64, device=dev, generator=gen)
    eps = 1e-3
    pad = batch - 3 * third
    # groups: [u, u + eps d, u - eps d, padding]; all with the same x0
    u = torch.cat([ua, ua + eps * d, ua - eps * d, ua[:, :, :pad]], dim=2).contiguous()
    x0 = torch.cat([xa, xa, xa, xa[:, :pad]], dim=1).contiguous()
    x = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev)
    lam = torch.empty_like(x)
    dJdu = torch.empty_like(u)
    _, J = gi.compute_states_dev(prob, x0, u, x)
    gi.compute_adjoints_dev(prob, u, None, lam, dJdu)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(lam).all()) and bool(torch.isfinite(dJdu).all())
    Jc, Jp, Jm = J[:third], J[third:2 * third], J[2 * third:3 * third]
    lhs = (Jp - Jm) / (2 * eps)
    rhs = (dJdu[:, :, :third] * d).sum(dim=(0, 1))
    scale = torch.maximum(rhs.abs(), (dJdu[:, :, :third].abs() * d.abs()).sum(dim=(0, 1)) * 1e-3)
    assert float(((lhs - rhs).abs() / scale).max()) < 1e-8, float(((lhs - rhs).abs() / scale).max())
    # second difference of a quadratic is direction-only: J(u+e d) + J(u-e d) - 2 J(u) = e^2 d'Hd >= 0 (R > 0, Q > 0)
    assert bool(((Jp + Jm - 2 * Jc) > 0).all())
    # linearity of the state rows: x(u + e d) - x(u) == x(u) - x(u - e d)
    xs = x[:, :nS, :]
    dev1 = xs[:, :, third:2 * third] - xs[:, :, :third]
    dev2 = xs[:, :, :third] - xs[:, :, 2 * third:3 * third]
    assert float((dev1 - dev2).abs().max()) < 1e-12 * max(1.0, float(xs.abs().max()))
    # invariants of the adjoint (SURVEY 8(c) KAT 3): lam(end,:) == 1
    assert bool((lam[:, nS, :] == 1.0).all())
    # padding trajectories repeat the first ones bit for bit (lane-placement independence)
    assert bool((x[:, :, 3 * third:] == x[:, :, :pad]).all()) and bool((dJdu[:, :, 3 * third:] == dJdu[:, :, :pad]).all())


@pytest.mark.parametrize("mapping", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("nS", [7, 32])
def test_known_answer_decoupled_linear_system(ocs, mapping, nS):
    """A known answer that owes nothing to the oracle: for x' = diag(lambda) x + Bu u with constant u and q = 0 the RK4
    recursion (RK4Integrator.m:37-51) has the closed form x_{i+1} = R(h lambda) x_i + h phi(h lambda) Bu u with the
    stability polynomial R(z) = 1 + z + z^2/2 + z^3/6 + z^4/24 and phi(z) = 1 + z/2 + z^2/6 + z^3/24; with the objective
    integrand e^{-rt} u'Ru alone the discrete adjoint (:72-88) is lam_i = R(h lambda) lam_{i+1} and dJ/du at the nodes follows
    from the quadrature weights.  Checked in extended precision (numpy longdouble) on every LQ mapping."""
    rng = np.random.default_rng(nS)
    nC, N, batch, T, r = 2, 48, 37, 1.2, 0.05
    lam_k = -rng.uniform(0.2, 6.0, nS)
    A = np.diag(lam_k)
    Bu = rng.normal(size=(nS, nC))
    q, rd = np.zeros(nS), rng.uniform(0.5, 2.0, nC)
    pg = ocs.LQProblem(A, Bu, q, rd, r, [[-1.0, 1.0]] * nC)
    tspan = np.linspace(0.0, T, N + 1)
    g = ocs.RK4Integrator(tspan).set_mapping(mapping)
    uc = rng.uniform(-1, 1, (nC, batch))
    u = np.repeat(uc[:, None, :], 2 * N + 1, axis=1)
    x0 = rng.normal(size=(nS, batch))
    lamT = np.vstack([rng.normal(size=(nS, batch)), np.ones((1, batch))])
    x, J = g.compute_states(pg, x0, u)
    lam, dJdu = g.compute_adjoints(pg, u, lamT)
    L = np.longdouble
    h = np.diff(tspan.astype(L))
    xr = x0.astype(L).copy()
    Bu_u = Bu.astype(L) @ uc.astype(L)
    for i in range(N):
        z = (h[i] * lam_k.astype(L))[:, None]
        R = 1 + z + z**2 / 2 + z**3 / 6 + z**4 / 24
        phi = 1 + z / 2 + z**2 / 6 + z**3 / 24
        xr = R * xr + h[i] * phi * Bu_u
    assert relerr(x[:nS, -1, :], xr.astype(np.float64)) < 1e-13
    # objective: the integrand does not depend on x: J = sum_i h_i/6 (e_A + 4 e_M + e_B) u'Ru
    tg = g.t.astype(L)
    e = np.exp(-L(r) * tg)
    w = np.sum(h / 6 * (e[0:-1:2] + 4 * e[1::2] + e[2::2]))
    Jr = w * np.sum(rd.astype(L)[:, None] * uc.astype(L) ** 2, axis=0)
    assert relerr(J, Jr.astype(np.float64)) < 1e-13
    # adjoint: lam(1:nS, 0) = prod_i R(h_i lambda) lamT(1:nS)   (q = 0: no coupling to the cost row)
    lr = lamT[:nS].astype(L).copy()
    for i in range(N - 1, -1, -1):
        z = (h[i] * lam_k.astype(L))[:, None]
        lr = (1 + z + z**2 / 2 + z**3 / 6 + z**4 / 24) * lr
    assert relerr(lam[:nS, 0, :], lr.astype(np.float64)) < 1e-13 and np.all(lam[nS] == 1.0)
    assert np.all(np.isfinite(dJdu)) and dJdu.shape == u.shape


def test_chunked_passes_stress_short(ocs):
    """tests/stress_lq_chunks.py, 12 random shapes (ragged groups, N down to 2, tail legs, explicit lamT): the requested chunked
    mapping against the serial one-wave mapping."""
    from tests.stress_lq_chunks import run
    failed, worst = run(ocs, 12, seed=7, verbose=False)
    assert failed == 0 and worst < 1e-11
