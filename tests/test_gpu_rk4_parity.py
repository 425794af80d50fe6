"""GPU parity: the HIP path, called through the C-ABI, against the CPU oracle.

Tolerance: fp64, GPU kernels contract a*b+c into FMA and hoist exp(-r t) into a table built
with the device exp; the oracle rounds every operation separately with libm's exp.  The
stated bar is 1e-12 relative, measured against max(1, |ref|).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = {"c": 1.5, "m": 3.0, "r": 0.05}
BOUNDS = [[0.0, 1.0]]
RTOL = 1e-12


def relerr(a, b):
    """max |a-b| / max(1,|b|); entries that are non-finite in the reference must match exactly."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    bad = ~np.isfinite(b)
    if bad.any():
        same = (np.isnan(a[bad]) & np.isnan(b[bad])) | (a[bad] == b[bad])
        if not same.all():
            return float("inf")
    ok = ~bad
    if not ok.any():
        return 0.0
    with np.errstate(invalid="ignore"):
        e = np.abs(a[ok] - b[ok]) / np.maximum(1.0, np.abs(b[ok]))
    return float("inf") if np.isnan(e).any() else float(np.max(e))


@pytest.fixture(scope="module")
def ocs():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import __graft_entry__ as g
    return g.load_package()


def _inputs(oracle, nS, N, batch, seed, T=10.0):
    rng = np.random.default_rng(seed)
    tspan = oracle.linspace(0.0, T, N + 1)
    t = np.zeros(2 * N + 1)
    t[0::2] = tspan
    t[1::2] = (tspan[:-1] + tspan[1:]) / 2
    f, ph = rng.uniform(0, 1, batch), rng.uniform(0, 2 * np.pi, batch)
    # SURVEY BL-2 candidates, amplitude reduced to [0.05, 0.45]: with the survey's 0.5 + 0.4 sin the
    # m = 1.5 state falls below its unstable equilibrium and runs off to -inf.
    u = np.clip(0.25 + 0.2 * np.sin(2 * np.pi * f[None, :] * t[:, None] + ph[None, :]), 0, 1)
    u = np.asfortranarray(u[None, :, :])
    x0 = rng.uniform(0.8, 2.5, (nS, batch))
    return tspan, x0, u


def test_plugin_methods_match_oracle(ocs, oracle):
    rng = np.random.default_rng(3)
    for m in ([3.0], [3.0, 2.5, 2.0, 1.5]):
        nS = len(m)
        pg = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS)
        po = oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
        k = 37
        t, y = rng.uniform(0, 10, k), rng.normal(1.5, 1, (nS + 1, k))
        u, v = rng.uniform(0, 1, (1, k)), rng.normal(size=(nS + 1, k))
        assert relerr(pg.F(t, y, u), po.F(t, y, u)) < 1e-14
        assert relerr(pg.dFdx_times_vec(t, y, u, v), po.dFdx_times_vec(t, y, u, v)) < 1e-14
        assert relerr(pg.dFdu_times_vec(t, y, u, v), po.dFdu_times_vec(t, y, u, v)) < 1e-14
    pt, ot = ocs.TestOCProblem(P, BOUNDS), oracle.TestOCProblem(P, BOUNDS)
    assert relerr(pt.F([1.0], [2.0, 0.0], [0.3]), ot.F([1.0], [2.0, 0.0], [0.3])) < 1e-15


@pytest.mark.parametrize("nS,N,batch,T", [(1, 500, 1, 10.0), (1, 37, 70, 2.0), (4, 200, 130, 10.0), (2, 3, 64, 0.2),
                                          (3, 1, 5, 0.05), (4, 1000, 64, 10.0), (2, 9, 129, 0.5)])
def test_states_and_adjoints_match_oracle(ocs, oracle, nS, N, batch, T):
    # N = 1, 3, 9, 37 exercise the remainder / partial-chunk paths of the prefetch pipeline,
    # batch = 1, 5, 70, 129, 130 the partially filled last wave
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    tspan, x0, u = _inputs(oracle, nS, N, batch, seed=100 + nS + N, T=T)
    pg, po = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS), oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    g = ocs.RK4Integrator(tspan)
    x, J = g.compute_states(pg, x0, u)
    lam, dJdu = g.compute_adjoints(pg, u)
    ref = oracle.batch_states_adjoints(po, tspan, x0, u)
    assert relerr(x, ref["x"]) < RTOL
    assert relerr(J, ref["J"]) < RTOL
    assert relerr(lam, ref["lam"]) < RTOL
    assert relerr(dJdu, ref["dJdu"]) < RTOL
    assert np.all(lam[-1] == 1.0)  # SURVEY KAT 3: exact


@pytest.mark.parametrize("nS,N,batch,T", [(4, 200, 130, 10.0), (2, 3, 64, 0.2), (2, 9, 129, 0.5), (4, 1, 5, 0.05),
                                          (4, 37, 1, 2.0), (2, 1000, 40, 10.0), (4, 8, 17, 0.4), (4, 16, 64, 0.8), (4, 48, 32, 2.0), (2, 8, 32, 0.4), (2, 40, 96, 2.0), (4, 1008, 16, 10.0),
                                          (2, 17, 33, 0.8), (4, 24, 3, 1.0), (4, 1003, 16, 10.0), (2, 17, 32, 0.8),
                                          (4, 12, 32, 0.6), (2, 15, 64, 0.7), (4, 7, 16, 0.3)])
@pytest.mark.parametrize("mapping", ["lane", "rowsplit", "pipeline", "scan"])
def test_both_mappings_match_oracle(ocs, oracle, nS, N, batch, T, mapping):
    # the row-split kernels (one state row per lane) must give the same answers as lane-per-trajectory,
    # including explicit lamT, lam-only / dJdu-only variants and partially filled last waves
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    tspan, x0, u = _inputs(oracle, nS, N, batch, seed=300 + nS + N, T=T)
    pg, po = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS), oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    g = ocs.RK4Integrator(tspan).set_mapping(mapping)
    D, tile = 8, 64 // nS
    if mapping == "pipeline" and (N < D or (batch % tile != 0 and (batch < tile or batch % 2 != 0))):
        # the wave-specialised kernels hand off in blocks of D steps over tiles of 64/nS trajectories; a step
        # count that is not a multiple of D is split (whole blocks on the pipeline kernel, the last < D steps
        # on the lane kernel, handing over the boundary column).  Fewer than D steps or a ragged tile are
        # refused when forced (automatic selection falls back), never mis-computed
        with pytest.raises(ocs.OcsError) as e:
            g.compute_states(pg, x0, u)
        assert e.value.code == -6
        return
    x, J = g.compute_states(pg, x0, u)
    if mapping == "pipeline" and batch % tile != 0:
        # (the ragged last tile is the state pass's: the wave-specialised ADJOINT kernel takes whole tiles only)
        ref = oracle.batch_states_adjoints(po, tspan, x0, u)
        assert relerr(x, ref["x"]) < RTOL and relerr(J, ref["J"]) < RTOL
        with pytest.raises(ocs.OcsError) as e:
            g.compute_adjoints(pg, u)
        assert e.value.code == -6
        return
    if mapping == "scan" and N < 4:
        with pytest.raises(ocs.OcsError) as e:
            g.compute_adjoints(pg, u)
        assert e.value.code == -6
        return
    lam, dJdu = g.compute_adjoints(pg, u)
    ref = oracle.batch_states_adjoints(po, tspan, x0, u)
    assert relerr(x, ref["x"]) < RTOL and relerr(J, ref["J"]) < RTOL
    assert relerr(lam, ref["lam"]) < RTOL and relerr(dJdu, ref["dJdu"]) < RTOL
    assert np.array_equal(x[-1, -1, :], J) and np.all(lam[-1] == 1.0)
    lamT = np.random.default_rng(1).normal(size=(nS + 1, batch))
    lam2 = g.compute_adjoints(pg, u, lamT, nargout=1)
    go = oracle.RK4Integrator(tspan)
    for b in sorted({0, batch - 1}):
        go.compute_states(po, x0[:, b], u[:, :, b])
        assert relerr(lam2[:, :, b], go.compute_adjoints(po, u[:, :, b], lamT[:, b], want_dJdu=False)) < RTOL


def test_rowsplit_rejects_unsupported(ocs):
    pg = ocs.TestOCProblem(P, BOUNDS)  # nS = 1: nothing to split
    g = ocs.RK4Integrator(np.linspace(0, 1, 11)).set_mapping("rowsplit")
    with pytest.raises(ocs.OcsError) as e:
        g.compute_states(pg, [1.0], np.zeros((1, 21)))
    assert e.value.code == -6


def test_single_trajectory_shapes_and_nonuniform_grid(ocs, oracle):
    # batch = 1 must reproduce the reference's shapes exactly; h = diff(tspan) may be non-uniform
    tspan = np.sort(np.concatenate([[0.0, 6.0], np.random.default_rng(8).uniform(0, 6, 49)]))
    pg, po = ocs.TestOCProblem(P, BOUNDS), oracle.TestOCProblem(P, BOUNDS)
    g, go = ocs.RK4Integrator(tspan), oracle.RK4Integrator(tspan)
    assert np.array_equal(g.t, go.t) and np.array_equal(g.h, go.h)
    u = np.random.default_rng(9).uniform(0, 1, (1, 101))
    x, J = g.compute_states(pg, [1.0], u)
    lam, dJdu = g.compute_adjoints(pg, u)
    xo, Jo = go.compute_states(po, [1.0], u)
    lamo, do = go.compute_adjoints(po, u)
    assert x.shape == (2, 51) and lam.shape == (2, 51) and dJdu.shape == (1, 101) and isinstance(J, float)
    assert relerr(x, xo) < RTOL and abs(J - Jo) < RTOL * abs(Jo)
    assert relerr(lam, lamo) < RTOL and relerr(dJdu, do) < RTOL
    # explicit terminal adjoint (RK4InfiniteIntegrator's use, RK4Integrator.m:63-69)
    lamT = np.array([0.37, 1.0])
    lam2 = g.compute_adjoints(pg, u, lamT, nargout=1)
    assert relerr(lam2, go.compute_adjoints(po, u, lamT, want_dJdu=False)) < RTOL


def test_ordering_contract_and_errors(ocs):
    pg = ocs.TestOCProblem(P, BOUNDS)
    g = ocs.RK4Integrator(np.linspace(0, 1, 11))
    u = np.zeros((1, 21))
    with pytest.raises(ocs.OcsError) as e:
        g.compute_adjoints(pg, u)  # before compute_states
    assert e.value.code == -3
    g.compute_states(pg, [1.0], u)
    g.compute_adjoints(pg, u)
    with pytest.raises(ocs.OcsError):
        ocs.RK4Integrator([0.0])
    with pytest.raises(ocs.OcsError):
        ocs.RK4Integrator([0.0, 1.0, 0.5])
    with pytest.raises(ocs.OcsError):
        ocs.LogisticProblem([3.0] * 9, 1.5, 0.05, BOUNDS)  # not in the kernel registry


def test_nonfinite_status(ocs):
    pg = ocs.TestOCProblem(P, BOUNDS)
    g = ocs.RK4Integrator(np.linspace(0, 50, 11))  # h = 5: RK4 blows up for the logistic state
    g.compute_states(pg, [50.0], np.zeros((1, 21)))
    assert g.status == 1  # OCS_NUM_NONFINITE


def test_per_trajectory_status_and_tracing(ocs):
    # a status word per trajectory (the global OCS_NUM_NONFINITE says only "some"), host and device forms;
    # roctx ranges around the entry points when a marker library is present (it is on this image)
    import torch
    pg = ocs.TestOCProblem(P, BOUNDS)
    g = ocs.RK4Integrator(np.linspace(0, 10, 21))      # h = 0.5: RK4 overflows from x0 = 50, is stable from x0 = 1
    x0 = np.array([[50.0, 1.0, 50.0, 1.0, 1.0]])
    _, J = g.compute_states(pg, x0, np.zeros((1, 41, 5)))
    assert g.status == 1
    st = g.trajectory_status(5)
    assert list(st) == [1, 0, 1, 0, 0] and list(~np.isfinite(J)) == [True, False, True, False, False]
    Jd = torch.tensor(J, device="cuda:0")
    std = ocs.trajectory_status_dev(Jd)
    torch.cuda.synchronize()
    assert std.cpu().tolist() == [1, 0, 1, 0, 0]
    with pytest.raises(ocs.OcsError):
        g.trajectory_status(7)                         # no call with that batch
    assert ocs.tracing_enabled()
    # device wrappers refuse wrongly shaped / placed tensors instead of faulting on the GPU
    xd = torch.empty((21, 2, 5), dtype=torch.float64, device="cuda:0")
    with pytest.raises(ValueError):
        g.compute_states_dev(pg, torch.ones((1, 5), dtype=torch.float64, device="cuda:0"),
                             torch.zeros((40, 1, 5), dtype=torch.float64, device="cuda:0"), xd)
    with pytest.raises(ValueError):
        g.compute_states_dev(pg, torch.ones((1, 5), dtype=torch.float64), torch.zeros((41, 1, 5), dtype=torch.float64,
                                                                                        device="cuda:0"), xd)


def test_blow_up_trajectories_agree(ocs, oracle):
    # harvest above the maximum sustainable yield of the m = 1.5 state: x -> -inf, then NaN.
    # Overflow happens at the same step on both sides; finite entries still meet the tolerance.
    N, batch = 200, 66
    tspan = oracle.linspace(0.0, 10.0, N + 1)
    m = [3.0, 1.5]
    pg, po = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS), oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    u = np.full((1, 2 * N + 1, batch), 0.9)
    x0 = np.tile([[1.0], [0.6]], (1, batch))
    g = ocs.RK4Integrator(tspan)
    x, J = g.compute_states(pg, x0, u)
    ref = oracle.batch_states_adjoints(po, tspan, x0, u, want=("x", "J"))
    assert g.status == 1 and not np.all(np.isfinite(ref["x"]))
    fin = np.isfinite(ref["x"])
    assert np.array_equal(fin, np.isfinite(x))
    # past |x| ~ 100 last-bit differences are amplified beyond any tolerance on the way to overflow
    sane = fin & (np.abs(ref["x"]) < 100)
    assert np.max(np.abs(x[sane] - ref["x"][sane]) / np.maximum(1, np.abs(ref["x"][sane]))) < 1e-9


def test_per_trajectory_parameters(ocs, oracle):
    # BL-3 style: c differs per instance
    batch, N = 70, 100
    tspan, x0, u = _inputs(oracle, 1, N, batch, seed=5)
    cs = np.random.default_rng(6).uniform(1, 2, batch)
    pg = ocs.TestOCProblem(P, BOUNDS)
    pg.set_batch_params([0], cs[None, :])
    g = ocs.RK4Integrator(tspan)
    x, J = g.compute_states(pg, x0, u)
    lam, dJdu = g.compute_adjoints(pg, u)
    for b in (0, 17, 69):
        po = oracle.TestOCProblem({"c": cs[b], "m": P["m"], "r": P["r"]}, BOUNDS)
        go = oracle.RK4Integrator(tspan)
        xo, Jo = go.compute_states(po, x0[:, b], u[:, :, b])
        lamo, do = go.compute_adjoints(po, u[:, :, b])
        assert relerr(x[:, :, b], xo) < RTOL and abs(J[b] - Jo) < RTOL * abs(Jo)
        assert relerr(lam[:, :, b], lamo) < RTOL and relerr(dJdu[:, :, b], do) < RTOL
    with pytest.raises(ocs.OcsError):
        pg.set_batch_params([2], cs[None, :])  # r feeds the time-coefficient table


def test_device_path_batch_minor(ocs, oracle):
    import torch
    nS, N, batch = 4, 250, 200
    m = [3.0, 2.5, 2.0, 1.5]
    tspan, x0, u = _inputs(oracle, nS, N, batch, seed=77)
    pg, po = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS), oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    g = ocs.RK4Integrator(tspan)
    dev = torch.device("cuda:0")
    x0d = torch.tensor(x0, device=dev)                                         # [nS][B]
    ud = torch.tensor(np.ascontiguousarray(u.transpose(1, 0, 2)), device=dev)   # [2N+1][nC][B]
    xd = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev)
    lamd, dd = torch.empty_like(xd), torch.empty_like(ud)
    _, Jd = g.compute_states_dev(pg, x0d, ud, xd)
    g.compute_adjoints_dev(pg, ud, None, lamd, dd)
    torch.cuda.synchronize()
    ref = oracle.batch_states_adjoints(po, tspan, x0, u)
    assert relerr(xd.cpu().numpy().transpose(1, 0, 2), ref["x"]) < RTOL
    assert relerr(Jd.cpu().numpy(), ref["J"]) < RTOL
    assert relerr(lamd.cpu().numpy().transpose(1, 0, 2), ref["lam"]) < RTOL
    assert relerr(dd.cpu().numpy().transpose(1, 0, 2), ref["dJdu"]) < RTOL
    # objective+gradient only: no trajectory outputs
    _, J2 = g.compute_states_dev(pg, x0d, ud, None)
    d2 = torch.empty_like(ud)
    g.compute_adjoints_dev(pg, ud, None, None, d2)
    torch.cuda.synchronize()
    assert torch.equal(J2, Jd) and torch.equal(d2, dd)


def test_full_size_properties_bl2(ocs, oracle):
    # BASELINE config 2 at full size (nS=4, N=1000, batch=4096): spot-check against the oracle
    # and check size-independent properties on all trajectories.
    import torch
    nS, N, batch = 4, 1000, 4096
    m = [3.0, 2.5, 2.0, 1.5]
    tspan, x0, u = _inputs(oracle, nS, N, batch, seed=20260401)
    pg, po = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS), oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    g = ocs.RK4Integrator(tspan)
    x, J = g.compute_states(pg, x0, u)
    lam, dJdu = g.compute_adjoints(pg, u)
    assert np.all(np.isfinite(x)) and np.all(np.isfinite(dJdu))
    assert np.array_equal(x[-1, -1, :], J)                 # J = x(end,end)
    assert np.all(lam[-1] == 1.0) and np.all(lam[:nS, -1, :] == 0.0)
    assert np.all(np.diff(x[-1], axis=0) >= 0)             # running cost of a non-negative integrand
    idx = np.array([0, 1, 63, 64, 2047, 4095])
    ref = oracle.batch_states_adjoints(po, tspan, x0[:, idx], u[:, :, idx])
    assert relerr(x[:, :, idx], ref["x"]) < RTOL and relerr(lam[:, :, idx], ref["lam"]) < RTOL
    assert relerr(dJdu[:, :, idx], ref["dJdu"]) < RTOL and relerr(J[idx], ref["J"]) < RTOL
    # linearity of the adjoint in lamT: lam(lamT1 + lamT2) = lam(lamT1) + lam(lamT2)
    rng = np.random.default_rng(1)
    l1, l2 = rng.normal(size=(nS + 1, batch)), rng.normal(size=(nS + 1, batch))
    a = g.compute_adjoints(pg, u, l1, nargout=1)
    b = g.compute_adjoints(pg, u, l2, nargout=1)
    c = g.compute_adjoints(pg, u, l1 + l2, nargout=1)
    assert relerr(a + b, c) < 1e-11


@pytest.mark.parametrize("nS,N,batch,T", [(1, 1, 3, 0.05), (1, 47, 70, 2.0), (1, 48, 64, 2.0), (1, 49, 129, 2.0),
                                          (2, 95, 33, 3.0), (2, 96, 64, 3.0), (4, 97, 17, 3.0), (4, 500, 48, 10.0),
                                          (1, 1000, 64, 10.0), (4, 2, 16, 0.1), (2, 4, 5, 0.2), (4, 5, 1, 0.2)])
def test_adjoint_scan_over_time(ocs, oracle, nS, N, batch, T):
    # the adjoint pass as a scan over time (row-separable problems): step counts around the superblock length
    # (identity-padded last superblock), ragged batches, explicit lamT, the lam-only and dJdu-only variants, lam0,
    # and a non-uniform grid
    import torch
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    tspan, x0, u = _inputs(oracle, nS, N, batch, seed=700 + nS + N, T=T)
    if N >= 5:
        tspan = np.sort(np.concatenate([[0.0, T], np.random.default_rng(N).uniform(0, T, N - 1)]))
    pg, po = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS), oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    g = ocs.RK4Integrator(tspan).set_mapping("scan")
    x, J = g.compute_states(pg, x0, u)
    if N < 4:
        # the scan takes whole chunks of 4 steps (a remainder goes to the lane kernel, which hands over the boundary
        # column); fewer than 4 steps are refused when forced (automatic selection falls back), never mis-computed
        with pytest.raises(ocs.OcsError) as e:
            g.compute_adjoints(pg, u)
        assert e.value.code == -6
        g = ocs.RK4Integrator(tspan)
        g.compute_states(pg, x0, u)
    lam, dJdu = g.compute_adjoints(pg, u)
    ref = oracle.batch_states_adjoints(po, tspan, x0, u)
    assert relerr(x, ref["x"]) < RTOL and relerr(J, ref["J"]) < RTOL
    assert relerr(lam, ref["lam"]) < RTOL and relerr(dJdu, ref["dJdu"]) < RTOL
    assert np.all(lam[-1] == 1.0) and np.all(lam[:nS, -1, :] == 0.0)
    lamT = np.random.default_rng(3).normal(size=(nS + 1, batch))
    lam2, d2 = g.compute_adjoints(pg, u, lamT)
    go = oracle.RK4Integrator(tspan)
    for b in sorted({0, batch // 2, batch - 1}):
        go.compute_states(po, x0[:, b], u[:, :, b])
        lo, do = go.compute_adjoints(po, u[:, :, b], lamT[:, b])
        assert relerr(lam2[:, :, b], lo) < RTOL and relerr(d2[:, :, b], do) < RTOL
    if N % 4:
        return
    # device entry points: lam only, dJdu only (same bits as the combined pass)
    dev = torch.device("cuda:0")
    x0d = torch.tensor(np.ascontiguousarray(x0), device=dev)
    ud = torch.tensor(np.ascontiguousarray(np.transpose(u, (1, 0, 2))), device=dev)
    xd = torch.empty((N + 1, nS + 1, batch), dtype=torch.float64, device=dev)
    g.compute_states_dev(pg, x0d, ud, xd)
    lamd, dd = torch.empty_like(xd), torch.empty_like(ud)
    g.compute_adjoints_dev(pg, ud, None, lamd, dd)
    lam_only, d_only = torch.empty_like(xd), torch.empty_like(ud)
    g.compute_adjoints_dev(pg, ud, None, lam_only, None)
    g.compute_adjoints_dev(pg, ud, None, None, d_only)
    torch.cuda.synchronize()
    assert torch.equal(lamd, lam_only) and torch.equal(dd, d_only)
    assert relerr(np.transpose(lamd.cpu().numpy(), (1, 0, 2)), ref["lam"]) < RTOL


@pytest.mark.parametrize("N,batch,T", [(1000, 64, 10.0), (24, 128, 1.0), (37, 64, 1.5), (8, 192, 0.4)])
def test_single_state_pipeline_adjoint(ocs, oracle, N, batch, T):
    # nS = 1 (TestOCProblem itself): the wave-specialised kernels with one lane per trajectory, forward and adjoint,
    # including a step count that is not a multiple of the block length, explicit lamT and the lam-only / dJdu-only variants
    tspan, x0, u = _inputs(oracle, 1, N, batch, seed=900 + N, T=T)
    pg, po = ocs.TestOCProblem(P, BOUNDS), oracle.TestOCProblem(P, BOUNDS)
    g = ocs.RK4Integrator(tspan).set_mapping("pipeline")
    x, J = g.compute_states(pg, x0, u)
    lam, dJdu = g.compute_adjoints(pg, u)
    ref = oracle.batch_states_adjoints(po, tspan, x0, u)
    assert relerr(x, ref["x"]) < RTOL and relerr(J, ref["J"]) < RTOL
    assert relerr(lam, ref["lam"]) < RTOL and relerr(dJdu, ref["dJdu"]) < RTOL
    assert np.all(lam[-1] == 1.0)
    gl = ocs.RK4Integrator(tspan).set_mapping("lane")
    gl.compute_states(pg, x0, u)
    laml, dl = gl.compute_adjoints(pg, u)
    assert relerr(lam, laml) < 1e-13 and relerr(dJdu, dl) < 1e-13
    lamT = np.random.default_rng(2).normal(size=(2, batch))
    g.compute_states(pg, x0, u)
    lam2 = g.compute_adjoints(pg, u, lamT, nargout=1)
    go = oracle.RK4Integrator(tspan)
    for b in (0, batch - 1):
        go.compute_states(po, x0[:, b], u[:, :, b])
        assert relerr(lam2[:, :, b], go.compute_adjoints(po, u[:, :, b], lamT[:, b], want_dJdu=False)) < RTOL


@pytest.mark.parametrize("mapping", ["auto", "lane", "rowsplit"])
def test_known_answers_without_the_oracle(ocs, mapping):
    """Known answers that owe nothing to the oracle, on the GPU kernels themselves (TestOCProblem.m:22-38 generalised to nS
    rows): (1) with u = 0 the state equation x' = x (m - x) has the closed form m x0 e^{mt} / (m + x0 (e^{mt} - 1)); the
    kernels' RK4 (RK4Integrator.m:37-51) must approach it at fourth order under grid refinement; (2) at the constant control
    u* and state x* = (m + sqrt(m^2 - 4 u*)) / 2 -- an equilibrium of the row -- the state stays put and
    J = c u*^2 ... + x*^2 summed against the quadrature of e^{-rt}: sum_i h/6 (e_A + 4 e_M + e_B) (sum_r x*_r^2 + c u*^2)."""
    m = [3.0, 2.5, 2.0, 1.5]
    c, r, T, batch = 1.5, 0.05, 2.0, 64
    prob = ocs.LogisticProblem(m, c, r, BOUNDS)
    rng = np.random.default_rng(11)
    x0 = rng.uniform(0.2, 2.0, (4, batch))
    errs = []
    for N in (64, 128):
        tspan = np.linspace(0.0, T, N + 1)
        g = ocs.RK4Integrator(tspan).set_mapping(mapping)
        x, _ = g.compute_states(prob, x0, np.zeros((1, 2 * N + 1, batch)))
        mm = np.asarray(m)[:, None]
        exact = mm * x0 * np.exp(mm * T) / (mm + x0 * (np.exp(mm * T) - 1.0))
        errs.append(np.max(np.abs(x[:4, -1, :] - exact)))
    assert errs[0] < 2e-6 and 12.0 < errs[0] / errs[1] < 20.0   # fourth order: halving h divides the error by ~16
    # (2) equilibrium of every row under a constant control
    ustar, N = 0.4, 96
    xs = np.array([(mk + np.sqrt(mk * mk - 4 * ustar)) / 2 for mk in m])
    tspan = np.linspace(0.0, T, N + 1)
    g = ocs.RK4Integrator(tspan).set_mapping(mapping)
    x, J = g.compute_states(prob, np.repeat(xs[:, None], batch, axis=1), np.full((1, 2 * N + 1, batch), ustar))
    assert np.max(np.abs(x[:4] - xs[:, None, None])) < 1e-13
    L = np.longdouble
    h = np.diff(tspan.astype(L))
    e = np.exp(-L(r) * g.t.astype(L))
    w = np.sum(h / 6 * (e[0:-1:2] + 4 * e[1::2] + e[2::2]))
    Jr = float(w * (np.sum(xs.astype(L) ** 2) + L(c) * L(ustar) ** 2))
    assert np.max(np.abs(J - Jr)) < 1e-12 * Jr


@pytest.mark.parametrize("kind,mapping", [("logistic4", "auto"), ("logistic4", "lane"), ("logistic3", "auto"), ("logistic1", "auto"),
                                          ("lq32", 2), ("lq32", 3), ("lq20", 1), ("predprey", "auto"), ("predprey", "lane")])
def test_adjoint_is_the_gradient_of_the_state_pass_without_the_oracle(ocs, kind, mapping):
    """The adjoint kernels against the state kernels alone (no oracle): dJdu of compute_adjoints (RK4Integrator.m:59-121) is
    the exact gradient of the discrete J of compute_states (:28-56), so its product with a direction d equals the
    directional derivative of J, taken here by a fourth-order central difference
    (8 (J(u + e d) - J(u - e d)) - (J(u + 2 e d) - J(u - 2 e d))) / (12 e) -- the four shifted controls of every trajectory
    ride in the batch.  Agreement to < 5e-9 relative pins every adjoint mapping to its state pass."""
    rng = np.random.default_rng(sum(map(ord, kind)))
    N, nb, eps = 64, 64, 1e-3
    if kind.startswith("logistic"):
        nS, nC, T = int(kind[-1]), 1, 2.0
        prob = ocs.LogisticProblem([3.0, 2.5, 2.0, 1.5][:nS], 1.5, 0.05, BOUNDS)
        x0 = rng.uniform(0.8, 1.6, (nS, nb))
        u = rng.uniform(0.1, 0.4, (nC, 2 * N + 1, nb))
    elif kind.startswith("lq"):
        from tests.user_problems import lq_matrices
        nS, nC, T = int(kind[2:]), 3, 0.4
        A, Bu, q, rd = lq_matrices(nS, nC)
        prob = ocs.LQProblem(A, Bu, q, rd, 0.05, [[-1.0, 1.0]] * nC)
        x0 = rng.normal(size=(nS, nb))
        u = rng.uniform(-1, 1, (nC, 2 * N + 1, nb))
    else:
        from tests.user_problems import PREDPREY_PARAMS, PREDPREY_SRC
        nS, nC, T = 2, 1, 3.0
        prob = ocs.UserProblem(PREDPREY_SRC, 2, 1, PREDPREY_PARAMS, BOUNDS)
        x0 = rng.uniform(1.0, 2.5, (nS, nb))
        u = rng.uniform(0.0, 1.0, (nC, 2 * N + 1, nb))
    d = rng.normal(size=u.shape)
    g = ocs.RK4Integrator(np.linspace(0.0, T, N + 1)).set_mapping(mapping)
    _, J0 = g.compute_states(prob, x0, u)
    _, dJdu = g.compute_adjoints(prob, u)
    shifts = (1.0, -1.0, 2.0, -2.0)
    ub = np.concatenate([u + s * eps * d for s in shifts], axis=2)     # four shifted copies of the batch
    _, Jb = g.compute_states(prob, np.tile(x0, (1, 4)), ub)
    Jp, Jm, Jpp, Jmm = (Jb[k * nb:(k + 1) * nb] for k in range(4))
    fd = (8.0 * (Jp - Jm) - (Jpp - Jmm)) / (12.0 * eps)
    an = np.sum(dJdu * d, axis=(0, 1))
    scale = np.maximum(np.abs(an), np.abs(J0) * 1e-3 + 1e-6)
    print(kind, mapping, "max relative difference", float(np.max(np.abs(an - fd) / scale)))
    assert np.max(np.abs(an - fd) / scale) < 5e-9   # (observed 2e-11 .. 4e-10: truncation of the difference formula)


def test_randomised_pass_pair_stress(ocs, oracle):
    """tests/stress_rk4.py, 40 random cases: nS 1..4, step counts around the block and chunk sizes of the kernels, whole and ragged
    tiles, uniform and non-uniform grids, every mapping (forced mappings a shape does not admit are refused, -6), default and
    explicit lamT -- x, J, lam, dJdu of sampled trajectories against the oracle at 1e-12 (the long run: python tests/stress_rk4.py)."""
    from tests.stress_rk4 import run
    lines = []
    failed, worst = run(ocs, oracle, 40, seed=4, log=lines.append)
    assert failed == 0, "\n".join(l for l in lines if "FAILED" in l)
    assert sum("refused" in l for l in lines[:-1]) < 30      # (most cases actually compute)
