"""GPU parity of the grid forward-backward sweep (A9/A10) against the CPU oracle's restatement of
the same scheme (fb_sweep.m with odevr7 -> grid RK4 + pchip coupling).  north_star tolerance for
fb_sweep output: 1e-10 relative; the sweep is a contraction here, so GPU/CPU last-bit differences
do not grow across sweeps and the passes themselves agree to ~1e-13."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = {"c": 1.5, "m": 3.0, "r": 0.05}
BOUNDS = [[0.0, 1.0]]
RTOL = 1e-10


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


@pytest.fixture(scope="module")
def ocs():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    return g.load_package()


def test_compute_x_lam_matches_oracle(ocs, oracle):
    rng = np.random.default_rng(2)
    for m, N, tspan in (([3.0], 200, oracle.linspace(0, 10, 201)),
                        ([3.0, 2.5], 60, np.sort(np.concatenate([[0.0, 6.0], rng.uniform(0, 6, 59)]))),
                        ([3.0], 1, np.array([0.0, 0.1])), ([3.0], 2, np.array([0.0, 0.1, 0.25]))):
        nS, N = len(m), tspan.size - 1
        pg, po = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS), oracle.LogisticProblem(m, P["c"], P["r"], BOUNDS)
        batch = 67
        u = rng.uniform(0.05, 0.45, (1, 2 * N + 1, batch))
        x0 = rng.uniform(0.9, 2.2, (nS, batch))
        x, lam, J = ocs.compute_x_lam_J(pg, x0, tspan, u)
        go = oracle.RK4Integrator(tspan)
        for b in (0, 1, 65, 66):
            xo, lo, Jo = oracle.compute_x_lam(go, po, x0[:, b], u[:, :, b], want_J=True)
            assert relerr(x[:, :, b], xo) < 1e-12 and relerr(lam[:, :, b], lo) < 1e-12
            assert abs(J[b] - Jo) < 1e-12 * max(1.0, abs(Jo))
        assert np.all(lam[:, -1, :] == 0.0)  # lam(TF) = 0*x0  compute_x_lam.m:4
        assert relerr(ocs.compute_J(pg, x0, tspan, u), J) < 1e-13   # functions/compute_J.m: the state pass alone


def test_fb_sweep_single_instance_like_the_reference(ocs, oracle):
    prob = ocs.TestOCProblem(P, BOUNDS)
    tspan = oracle.linspace(0, 10, 1001)
    soln = ocs.fb_sweep(prob, [1.0], tspan)
    assert set(soln) == {"x", "lam", "u", "J"}
    ref = oracle.fb_sweep(oracle.TestOCProblem(P, BOUNDS), [1.0], tspan)
    assert ref["_sweeps"] > 0
    tq = ref["_interpPts"]
    assert abs(soln["J"] - ref["J"]) < RTOL * abs(ref["J"])
    assert relerr(soln["u"](tq), ref["u"]) < RTOL
    assert relerr(soln["x"](tspan), ref["x"]) < RTOL and relerr(soln["lam"](tspan), ref["lam"]) < RTOL
    # turnpike: the optimal harvest sits at the analytic equilibrium in mid-horizon (SURVEY KAT 1)
    assert abs(soln["u"](np.array([5.0]))[0, 0] - 0.72336878009798256) < 1e-3
    # non-convergence -> empty struct (fb_sweep.m:77)
    assert ocs.fb_sweep(prob, [1.0], tspan, {"nSWEEPS": 2}) == {}


def test_fb_sweep_batch_bl3_style(ocs, oracle):
    """SURVEY BL-3: instances differ in x0 ~ U(0.5,2.5) and c ~ U(1,2); per-instance sweep counts,
    maxChange histories and solutions must match the oracle run instance by instance."""
    rng = np.random.default_rng(20260402)
    batch, N = 70, 400
    tspan = oracle.linspace(0, 10, N + 1)
    x0 = rng.uniform(0.5, 2.5, (1, batch))
    cs = rng.uniform(1.0, 2.0, batch)
    prob = ocs.TestOCProblem(P, BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    opts = {"nERROR_PTS": 401, "nINTERP_PTS": 201}
    r = ocs.fb_sweep_batch(prob, x0, tspan, opts)
    assert r["status"] == 0 and r["sweeps"].min() >= 3 and r["sweeps"].max() <= 20
    assert len(set(r["sweeps"].tolist())) > 1  # instances really stop at different sweeps
    for b in (0, 1, 13, 63, 64, 69):
        ref = oracle.fb_sweep(oracle.TestOCProblem({"c": cs[b], "m": P["m"], "r": P["r"]}, BOUNDS), x0[:, b], tspan, opts)
        k = ref["_sweeps"]
        assert r["sweeps"][b] == k
        mc = r["maxChange"][:, b]
        assert relerr(mc[:k], ref["_maxChange"][:k]) < 1e-6 and np.all(np.isnan(mc[k:]))
        assert abs(r["J"][b] - ref["J"]) < RTOL * abs(ref["J"])
        assert relerr(r["x"][:, :, b], ref["x"]) < RTOL and relerr(r["lam"][:, :, b], ref["lam"]) < RTOL
        assert relerr(r["u"][:, :, b], ref["u"]) < RTOL


def test_fb_sweep_options_u0_and_offgrid_error_points(ocs, oracle):
    # numeric u0 (evenly spaced samples -> pchip, fb_sweep.m:61-66) and error points that are NOT grid nodes
    tspan = oracle.linspace(0, 8, 161)
    u0 = np.array([[0.2, 0.6, 0.9, 0.4, 0.1]])
    opts = {"u0": u0, "nERROR_PTS": 333, "nINTERP_PTS": 77, "uRelTol": 1e-6, "uAbsTol": 1e-6}
    prob = ocs.LogisticProblem([3.0, 2.5], P["c"], P["r"], BOUNDS)
    r = ocs.fb_sweep_batch(prob, np.array([[1.0], [1.5]]), tspan, opts)
    ref = oracle.fb_sweep(oracle.LogisticProblem([3.0, 2.5], P["c"], P["r"], BOUNDS), [1.0, 1.5], tspan, opts)
    assert r["sweeps"][0] == ref["_sweeps"] > 0
    assert abs(r["J"][0] - ref["J"]) < RTOL * abs(ref["J"])
    assert relerr(r["u"][:, :, 0], ref["u"]) < RTOL and relerr(r["lam"][:, :, 0], ref["lam"]) < RTOL
    # callable u0
    r2 = ocs.fb_sweep_batch(prob, np.array([[1.0], [1.5]]), tspan, {"u0": lambda t: 0.3 + 0.0 * np.atleast_2d(t)})
    ref2 = oracle.fb_sweep(oracle.LogisticProblem([3.0, 2.5], P["c"], P["r"], BOUNDS), [1.0, 1.5], tspan,
                           {"u0": lambda t: 0.3 + 0.0 * np.atleast_2d(t)})
    assert r2["sweeps"][0] == ref2["_sweeps"] and abs(r2["J"][0] - ref2["J"]) < RTOL * abs(ref2["J"])


def test_fb_sweep_full_size_properties(ocs, oracle):
    """BASELINE config 3 shape on device buffers (batch reduced to 4096 for test time): every instance
    converges, solutions satisfy the optimality system, spot checks against the oracle."""
    import torch
    rng = np.random.default_rng(20260402)
    batch, N = 4096, 1000
    tspan = oracle.linspace(0, 10, N + 1)
    x0 = rng.uniform(0.5, 2.5, (1, batch))
    cs = rng.uniform(1.0, 2.0, batch)
    prob = ocs.TestOCProblem(P, BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    integ = ocs.RK4Integrator(tspan)
    r = ocs.fb_sweep_dev(prob, integ, torch.tensor(x0, device="cuda:0"))
    torch.cuda.synchronize()
    sw = r["sweeps"].cpu().numpy()
    assert r["status"] == 0 and sw.min() > 0
    u = r["u"].cpu().numpy()[:, 0, :]            # [nINTERP][B]
    lam = r["lam"].cpu().numpy()[:, 0, :]        # [N+1][B]
    assert np.all(u >= 0.0) and np.all(u <= 1.0) and np.all(lam[-1] == 0.0)
    # ControlChar identity at the nodes: u = clamp(lam e^{rt} / (2c), 0, 1)   (interpPts == nodes here)
    expect = np.clip(lam * np.exp(P["r"] * tspan)[:, None] / (2 * cs[None, :]), 0.0, 1.0)
    assert np.max(np.abs(u - expect)) < 1e-12
    for b in (0, 4095):
        ref = oracle.fb_sweep(oracle.TestOCProblem({"c": cs[b], "m": P["m"], "r": P["r"]}, BOUNDS), x0[:, b], tspan)
        assert sw[b] == ref["_sweeps"] and abs(r["J"][b].item() - ref["J"]) < RTOL * abs(ref["J"])


@pytest.mark.parametrize("nS,N,batch", [(1, 400, 70), (1, 37, 5), (2, 203, 130), (4, 64, 64), (3, 9, 3), (1, 8, 64),
                                        (1, 5, 2), (1, 120, 300), (4, 48, 200), (1, 400, 128), (2, 16, 96),
                                        (4, 24, 48), (2, 208, 64)])
def test_fused_costate_update_equals_separate_kernels(ocs, oracle, nS, N, batch):
    """With the error points on the grid nodes the costate pass, the in-place control update and the weighted
    change (fb_sweep.m:95-96, :107) run as one kernel (update waves trailing the marching wave through an LDS ring of
    lam nodes).  It must reproduce the separate kernels: same sweep counts, same maxChange history, same solution;
    and the oracle instance by instance."""
    rng = np.random.default_rng(N * 10 + nS)
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    T = 10.0 if N >= 200 else 1.0
    tspan = oracle.linspace(0, T, N + 1)
    x0 = rng.uniform(0.8, 2.0, (nS, batch))
    cs = rng.uniform(1.0, 2.0, batch)
    prob = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    base = {"nERROR_PTS": N + 1, "nINTERP_PTS": 33, "nSWEEPS": 40}
    ra = ocs.fb_sweep_batch(prob, x0, tspan, dict(base))
    rb = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, fused_update_off=1))
    if batch >= 128:
        # the same with the batch cut into windows that run their sweep loops on separate streams: an instance's
        # result does not depend on which window it is in
        rw = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, nWINDOWS=2))
        assert np.array_equal(ra["sweeps"], rw["sweeps"])
        # (a ragged last window runs the lane kernel: another order of the objective sum, and the plain form of the
        # logistic rows where the pipeline kernel marches about their vertex -- round-off level differences)
        for key in ("x", "lam", "u", "J"):
            assert relerr(ra[key], rw[key]) < 1e-12, key
        assert relerr(np.nan_to_num(ra["maxChange"]), np.nan_to_num(rw["maxChange"])) < 1e-6
        assert np.array_equal(np.isnan(ra["maxChange"]), np.isnan(rw["maxChange"]))
    # ... without the fold of the control update into the state pass of the next sweep (option 3: sweeps >= 2 are
    # forward, costate, control update, advance instead of forward-with-ControlChar, costate-with-convergence-test)
    rd = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, fused_update_off=3))
    assert np.array_equal(ra["sweeps"], rd["sweeps"])
    for key in ("x", "lam", "u", "J"):
        assert relerr(ra[key], rd[key]) < 1e-12, key
    assert relerr(np.nan_to_num(ra["maxChange"]), np.nan_to_num(rd["maxChange"])) < 1e-6
    assert np.array_equal(np.isnan(ra["maxChange"]), np.isnan(rd["maxChange"]))
    # ... and with the pchip midpoints of x from their own kernel instead of inside the costate / control kernels
    rc = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, fused_update_off=2))
    assert np.array_equal(ra["sweeps"], rc["sweeps"])
    for key in ("x", "lam", "u", "J"):
        assert relerr(ra[key], rc[key]) < 1e-12, key
    assert np.array_equal(ra["sweeps"], rb["sweeps"]) and ra["sweeps"].min() > 0
    # near convergence the weighted change is |du| ~ 1e-8 over 1e-7: one ulp in u (the reciprocal in the fused
    # ControlChar) moves it by ~1e-9 relative
    assert relerr(np.nan_to_num(ra["maxChange"]), np.nan_to_num(rb["maxChange"])) < 1e-6
    assert np.array_equal(np.isnan(ra["maxChange"]), np.isnan(rb["maxChange"]))
    for key in ("x", "lam", "u", "J"):
        assert relerr(ra[key], rb[key]) < 1e-12, key
    for b in sorted({0, batch // 2, batch - 1}):
        ref = oracle.fb_sweep(oracle.LogisticProblem(m, cs[b], P["r"], BOUNDS), x0[:, b], tspan, base)
        assert ra["sweeps"][b] == ref["_sweeps"]
        assert abs(ra["J"][b] - ref["J"]) < RTOL * abs(ref["J"])
        assert relerr(ra["x"][:, :, b], ref["x"]) < RTOL and relerr(ra["lam"][:, :, b], ref["lam"]) < RTOL
        assert relerr(ra["u"][:, :, b], ref["u"]) < RTOL


@pytest.mark.parametrize("nS,N,batch", [(1, 64, 64), (2, 96, 64), (4, 40, 32)])
def test_fold_on_a_bitwise_uniform_grid(ocs, oracle, nS, N, batch):
    """Step 2^-5: every step size is the same bit pattern, so the state pass keeps h, h/2, h/6 in registers (another
    instance of the kernel that forms its control from the costate of the sweep before)."""
    rng = np.random.default_rng(N + nS)
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    tspan = np.arange(N + 1) / 32.0
    x0 = rng.uniform(0.8, 2.0, (nS, batch))
    cs = rng.uniform(1.0, 2.0, batch)
    prob = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    base = {"nERROR_PTS": N + 1, "nINTERP_PTS": 33, "nSWEEPS": 40}
    ra = ocs.fb_sweep_batch(prob, x0, tspan, dict(base))
    rd = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, fused_update_off=3))
    assert np.array_equal(ra["sweeps"], rd["sweeps"]) and ra["sweeps"].min() > 0
    for key in ("x", "lam", "u", "J"):
        assert relerr(ra[key], rd[key]) < 1e-12, key
    for b in (0, batch - 1):
        ref = oracle.fb_sweep(oracle.LogisticProblem(m, cs[b], P["r"], BOUNDS), x0[:, b], tspan, base)
        assert ra["sweeps"][b] == ref["_sweeps"]
        assert abs(ra["J"][b] - ref["J"]) < RTOL * abs(ref["J"])
        assert relerr(ra["x"][:, :, b], ref["x"]) < RTOL and relerr(ra["lam"][:, :, b], ref["lam"]) < RTOL
        assert relerr(ra["u"][:, :, b], ref["u"]) < RTOL


def test_fold_first_sweep_with_a_nonzero_lower_bound(ocs, oracle):
    """The folded path starts from a costate whose ControlChar is the lower bound (u0 = ControlBounds(:,1), fb_sweep.m:23):
    with lb != 0 the first sweep's weighted change, the end node included, must be taken against that bound."""
    bounds = [[0.15, 0.6]]
    rng = np.random.default_rng(77)
    N, batch = 80, 64
    tspan = oracle.linspace(0, 4, N + 1)
    x0 = rng.uniform(0.8, 2.0, (1, batch))
    cs = rng.uniform(1.0, 2.0, batch)
    prob = ocs.LogisticProblem([3.0], P["c"], P["r"], bounds)
    prob.set_batch_params([0], cs[None, :])
    base = {"nERROR_PTS": N + 1, "nINTERP_PTS": 33, "nSWEEPS": 40}
    ra = ocs.fb_sweep_batch(prob, x0, tspan, dict(base))
    rd = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, fused_update_off=3))
    assert np.array_equal(ra["sweeps"], rd["sweeps"]) and ra["sweeps"].min() > 0
    assert relerr(np.nan_to_num(ra["maxChange"]), np.nan_to_num(rd["maxChange"])) < 1e-6
    for key in ("x", "lam", "u", "J"):
        assert relerr(ra[key], rd[key]) < 1e-12, key
    for b in (0, batch - 1):
        ref = oracle.fb_sweep(oracle.LogisticProblem([3.0], cs[b], P["r"], bounds), x0[:, b], tspan, base)
        k = ref["_sweeps"]
        assert ra["sweeps"][b] == k
        assert relerr(ra["maxChange"][:k, b], ref["_maxChange"][:k]) < 1e-6
        assert abs(ra["J"][b] - ref["J"]) < RTOL * abs(ref["J"])
        assert relerr(ra["u"][:, :, b], ref["u"]) < RTOL and relerr(ra["lam"][:, :, b], ref["lam"]) < RTOL


@pytest.mark.parametrize("n,nComp,batch,nq", [(30, 3, 70, 211), (2, 1, 5, 9), (3, 2, 300, 17), (1001, 1, 64, 1001)])
def test_batched_vector_interpolant_on_device(ocs, oracle, n, nComp, batch, nq):
    """vectorInterpolant.m:1-12 for a batch of sample sets on the device (the resampling step after the solvers,
    single_shooting.m:128-130, fb_sweep.m:123): every instance against the oracle's pchip / linear / previous,
    query points on the nodes, between them and outside the grid."""
    import torch
    rng = np.random.default_rng(n + nq)
    x = np.sort(rng.uniform(0, 5, n)) if n > 3 else np.linspace(0.0, 1.0, n)
    v = rng.normal(size=(n, nComp, batch))
    v[:, 0, :] = np.sin(2 * x)[:, None] * rng.uniform(0.5, 2.0, batch)[None, :]
    if nComp > 1:
        v[:, 1, :] = np.where(x > 2, 1.0, 0.0)[:, None]      # flat pieces and a jump: the zero-slope rule
    q = np.concatenate([x, rng.uniform(x[0] - 0.3, x[-1] + 0.3, nq - n)]) if nq > n else rng.uniform(-0.2, 1.2, nq)
    vd = torch.tensor(v, device="cuda:0")
    for name, m in (("pchip", oracle.INTERP_PCHIP), ("linear", oracle.INTERP_LINEAR), ("previous", oracle.INTERP_PREVIOUS)):
        got = ocs.vectorInterpolant_dev(x, vd, name)(q).cpu().numpy()
        assert got.shape == (q.size, nComp, batch)
        for b in sorted({0, batch // 2, batch - 1}):
            ref = oracle.vector_interp(x, v[:, :, b].T, m, q)     # nComp x nq
            # pchip / linear on device: fused multiply-adds and another order of the divisions than the oracle's loops
            assert np.allclose(got[:, :, b].T, ref, rtol=1e-12, atol=1e-13, equal_nan=True), (name, b)
            if name == "previous":
                assert np.array_equal(got[:, :, b].T, ref, equal_nan=True), (name, b)
    for name in ("nearest", "next"):      # sample-picking methods: the device picks what the host routine picks
        got = ocs.vectorInterpolant_dev(x, vd, name)(q).cpu().numpy()
        for b in sorted({0, batch - 1}):
            assert np.array_equal(got[:, :, b].T, ocs.vectorInterpolant(x, v[:, :, b].T, name)(q), equal_nan=True), (name, b)


def test_cost_row_option(ocs):
    """soln of the reference carries x, lam, u and the scalar J (fb_sweep.m:117-125); the running-objective row of the
    augmented state is written only on request.  With it: its last column is J and its first 0; the other outputs do
    not depend on the option."""
    import torch
    rng = np.random.default_rng(8)
    B, N = 128, 400
    x0 = torch.tensor(rng.uniform(0.5, 2.5, (1, B)), device="cuda:0")
    prob = ocs.TestOCProblem(P, BOUNDS)
    prob.set_batch_params([0], rng.uniform(1.0, 2.0, B)[None, :])
    integ = ocs.RK4Integrator(ocs.linspace(0, 10, N + 1))
    ra = ocs.fb_sweep_dev(prob, integ, x0)
    rb = ocs.fb_sweep_dev(prob, integ, x0, {"cost_row": 1})
    xa, xb = ra["xaug"].cpu().numpy(), rb["xaug"].cpu().numpy()
    assert np.array_equal(xa[:, 0, :], xb[:, 0, :]) and torch.equal(ra["J"], rb["J"]) and torch.equal(ra["u"], rb["u"])
    assert torch.equal(ra["lam"], rb["lam"]) and torch.equal(ra["sweeps"], rb["sweeps"])
    assert np.all(xb[0, 1, :] == 0.0) and np.array_equal(xb[-1, 1, :], rb["J"].cpu().numpy())
    assert np.all(np.diff(xb[:, 1, :], axis=0) > 0)          # the integrand x^2 + c u^2 is positive


@pytest.mark.parametrize("nSWEEPS", [1, 3, 40])
def test_sweeps_enqueued_ahead_equal_the_plain_loop(ocs, nSWEEPS):
    """On shapes the wave-specialised kernels take, sweep k+1 is enqueued before the host knows how many instances
    sweep k left active (its kernels return at once if none).  Same results as the loop that waits, also when the
    sweep limit ends the solve (no instance / not every instance converged: fb_sweep.m:77, :86)."""
    import torch
    rng = np.random.default_rng(3)
    B, N = 192, 160
    x0 = torch.tensor(rng.uniform(0.5, 2.5, (1, B)), device="cuda:0")
    prob = ocs.TestOCProblem(P, BOUNDS)
    prob.set_batch_params([0], rng.uniform(1.0, 2.0, B)[None, :])
    integ = ocs.RK4Integrator(ocs.linspace(0, 10, N + 1))
    ra = ocs.fb_sweep_dev(prob, integ, x0, {"nSWEEPS": nSWEEPS, "nERROR_PTS": N + 1, "nINTERP_PTS": 41})
    rb = ocs.fb_sweep_dev(prob, integ, x0, {"nSWEEPS": nSWEEPS, "nERROR_PTS": N + 1, "nINTERP_PTS": 41,
                                            "fused_update_off": 2})
    assert torch.equal(ra["sweeps"], rb["sweeps"]) and ra["status"] == rb["status"]
    sw = ra["sweeps"].cpu().numpy()
    if nSWEEPS == 1:
        assert sw.max() == 0 and ra["status"] == 2            # OCS_NUM_NOT_CONVERGED: nobody converges in one sweep
    if nSWEEPS == 40:
        assert sw.min() > 0 and ra["status"] == 0
    for key in ("xaug", "lam", "u", "J"):
        a, b = ra[key].cpu().numpy(), rb[key].cpu().numpy()
        if key == "xaug":
            a, b = a[:, :1, :], b[:, :1, :]
        assert relerr(a, b) < 1e-12, key
    ma, mb = ra["maxChange"].cpu().numpy(), rb["maxChange"].cpu().numpy()
    assert np.array_equal(np.isnan(ma), np.isnan(mb)) and relerr(np.nan_to_num(ma), np.nan_to_num(mb)) < 1e-6


def test_damped_update_extension(ocs, oracle):
    """The reference has no damping (u = uNew, fb_sweep.m:85).  uRelax (an extension, off by default) replaces :85 by
    u = u + uRelax (uNew - u) after the unchanged convergence test.  On TestOCProblem the undamped sweep is already a
    contraction, so damping only slows it down; what is checked is the rule itself: instance by instance the oracle's
    loop with the same rule (sweep counts, change history, solution), the same results from every mapping of the update
    (fused with the change of the control / separate kernels), the same fixed point as the undamped sweep."""
    rng = np.random.default_rng(5)
    N, batch = 96, 64
    tspan = oracle.linspace(0, 3.0, N + 1)
    x0 = rng.uniform(0.8, 1.6, (1, batch))
    cs = rng.uniform(1.0, 2.0, batch)
    prob = ocs.LogisticProblem([3.0], P["c"], P["r"], BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    base = {"nERROR_PTS": N + 1, "nINTERP_PTS": 33, "nSWEEPS": 120}
    plain = ocs.fb_sweep_batch(prob, x0, tspan, dict(base))
    damped = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, uRelax=0.6))
    assert plain["sweeps"].min() > 0 and damped["sweeps"].min() > 0
    assert np.all(damped["sweeps"] > plain["sweeps"])
    assert relerr(damped["J"], plain["J"]) < 1e-6 and relerr(damped["u"], plain["u"]) < 1e-5      # the same fixed point
    sep = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, uRelax=0.6, fused_update_off=1))
    assert np.array_equal(damped["sweeps"], sep["sweeps"])
    for key in ("x", "lam", "u", "J"):
        assert relerr(damped[key], sep[key]) < 1e-12, key
    for b in (0, batch // 2, batch - 1):
        ref = oracle.fb_sweep(oracle.LogisticProblem([3.0], cs[b], P["r"], BOUNDS), x0[:, b], tspan, dict(base, uRelax=0.6))
        k = ref["_sweeps"]
        assert damped["sweeps"][b] == k > 0
        assert relerr(damped["maxChange"][:k, b], ref["_maxChange"][:k]) < 1e-6
        assert abs(damped["J"][b] - ref["J"]) < RTOL * abs(ref["J"])
        assert relerr(damped["u"][:, :, b], ref["u"]) < RTOL and relerr(damped["lam"][:, :, b], ref["lam"]) < RTOL
    with pytest.raises(Exception):
        ocs.fb_sweep_batch(prob, x0, tspan, dict(base, uRelax=1.5))


def test_numpy_linspace_grid_takes_the_same_path(ocs, oracle):
    """fb_sweep.m:69 builds the error points with MATLAB's linspace; numpy's differs from it in the last bit of some nodes.
    Error points within a few ulp of the nodes count as the nodes (so such a tspan does not fall back to the unfused
    kernels); the results agree with those on the MATLAB-style grid to round-off and with the oracle on the same grid."""
    rng = np.random.default_rng(9)
    N, batch = 200, 64
    ta, tb = oracle.linspace(0, 10, N + 1), np.linspace(0, 10, N + 1)
    assert not np.array_equal(ta, tb)          # the premise: the two grids differ in some last bits
    x0 = rng.uniform(0.5, 2.5, (1, batch))
    cs = rng.uniform(1.0, 2.0, batch)
    prob = ocs.TestOCProblem(P, BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    opts = {"nERROR_PTS": N + 1, "nINTERP_PTS": 41}
    ra = ocs.fb_sweep_batch(prob, x0, ta, opts)
    rb = ocs.fb_sweep_batch(prob, x0, tb, opts)
    assert np.array_equal(ra["sweeps"], rb["sweeps"]) and ra["sweeps"].min() > 0
    for key in ("x", "lam", "u", "J"):
        assert relerr(ra[key], rb[key]) < 1e-11, key
    for b in (0, batch - 1):
        ref = oracle.fb_sweep(oracle.TestOCProblem({"c": cs[b], "m": P["m"], "r": P["r"]}, BOUNDS), x0[:, b], tb, opts)
        assert rb["sweeps"][b] == ref["_sweeps"]
        assert abs(rb["J"][b] - ref["J"]) < RTOL * abs(ref["J"]) and relerr(rb["lam"][:, :, b], ref["lam"]) < RTOL


def test_fold_stress_per_instance(ocs, oracle):
    """fb_sweep.m:79-87, 99-115 on randomly drawn problems, checked PER INSTANCE (tests/stress_fold.py): the folded
    kernels, the unfolded path and the oracle agree on the sweep count of every instance (0 = not converged, the
    reference's empty struct); every converged instance -- the cases drawn here hold instances that need 26 to 59 sweeps
    (lower bound -0.2) next to instances that never converge -- meets 1e-11 fold-vs-unfolded and 1e-10 against
    oracle.fb_sweep (the three slowest-converging instances of a case, both ends of the batch, one that fails)."""
    from tests import stress_fold
    rng = np.random.default_rng(1)
    want = {3: 26, 13: 40, 21: 26, 25: 40, 33: 38}     # case -> sweeps at least one converged instance needs
    for case in range(max(want) + 1):
        c = stress_fold.draw_case(rng, case, ocs)
        if case not in want:
            continue
        r = stress_fold.run_case(ocs, oracle, c)
        assert r["ok"], (case, r["failures"][:4])
        assert r["sweeps"].max() >= want[case] and (r["sweeps"] == 0).any(), (case, r["sweeps"].max())
        assert r["err_fold"] < 1e-11 and r["err_oracle"] < 1e-10


@pytest.mark.parametrize("nS,batch", [(1, 70), (1, 1000), (2, 98), (4, 54), (4, 1000)])
def test_two_kernel_sweep_on_ragged_batches(ocs, oracle, nS, batch):
    """A batch that is not a multiple of the tile of 64/nS instances: the last workgroup of every sweep kernel takes the LAST tile,
    overlapping its neighbour (even batches beyond one tile) -- the folded sweep (path 4) instead of the kernel-by-kernel
    sequence; same sweep counts and solution as that sequence for every instance, the oracle at both ends of the batch."""
    rng = np.random.default_rng(batch + nS)
    N = 64
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    tspan = oracle.linspace(0, 1.0, N + 1)
    x0 = rng.uniform(0.8, 2.0, (nS, batch))
    cs = rng.uniform(1.0, 2.0, batch)
    prob = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    base = {"nERROR_PTS": N + 1, "nINTERP_PTS": 17, "nSWEEPS": 40}
    ga, gb = ocs.RK4Integrator(tspan), ocs.RK4Integrator(tspan)
    ra = ocs.fb_sweep_batch(prob, x0, tspan, dict(base), integrator=ga)
    rb = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, fused_update_off=1), integrator=gb)
    assert ocs.fb_sweep_path(ga) == 4 and ocs.fb_sweep_path(gb) == 1 and batch % (64 // nS) != 0
    assert np.array_equal(ra["sweeps"], rb["sweeps"]) and ra["sweeps"].min() > 0
    for key in ("x", "lam", "u", "J"):
        assert relerr(ra[key], rb[key]) < 1e-12, key
    for b in (0, batch - 2, batch - 1):
        ref = oracle.fb_sweep(oracle.LogisticProblem(m, cs[b], P["r"], BOUNDS), x0[:, b], tspan, base)
        assert ra["sweeps"][b] == ref["_sweeps"] and abs(ra["J"][b] - ref["J"]) < RTOL * abs(ref["J"])
        assert relerr(ra["x"][:, :, b], ref["x"]) < RTOL and relerr(ra["u"][:, :, b], ref["u"]) < RTOL


def test_device_pchip_against_scipy(ocs):
    """vectorInterpolant.m:6 ('pchip') on the device against scipy's PchipInterpolator (Fritsch-Carlson with MATLAB's end
    slopes): no oracle involved.  Non-uniform nodes, monotone and oscillating data, query points on nodes and between."""
    import torch
    from scipy.interpolate import PchipInterpolator
    rng = np.random.default_rng(5)
    n, batch = 41, 7
    x = np.concatenate([[0.0], np.sort(rng.uniform(0, 3, n - 2)), [3.0]])
    v = np.stack([np.cumsum(rng.uniform(0, 1, (n, batch)), axis=0), np.sin(3 * x)[:, None] * rng.normal(size=(1, batch))])   # [2][n][batch]
    q = np.concatenate([x[::5], rng.uniform(0, 3, 50)])
    vd = torch.tensor(np.ascontiguousarray(v.transpose(1, 0, 2)), device="cuda")   # [n][comp][batch]
    got = ocs.vectorInterpolant_dev(x, vd, "pchip")(q).cpu().numpy()               # [nq][comp][batch]
    for c in range(2):
        for b in range(batch):
            ref = PchipInterpolator(x, v[c, :, b])(q)
            assert np.max(np.abs(got[:, c, b] - ref)) < 1e-13 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("nS,batch", [(1, 128), (2, 64), (4, 32)])
def test_fb_sweep_solution_properties_without_the_oracle(ocs, nS, batch):
    """Properties of soln = fb_sweep(...) that need no oracle (fb_sweep.m:79-125): the returned control is ControlChar of
    the returned costate at the interpolation points (:123; here = the grid nodes, numpy restatement of
    TestOCProblem's ControlChar), the recorded change of the control is <= 1 exactly in the sweep an instance converged
    in and > 1 in every sweep before (:108-110), lam(TF) = 0 (compute_x_lam.m:4), and the control the solution
    implies reproduces the solution's state and costate through compute_x_lam to the order of the sweep's tolerance."""
    c, r, N, T = 1.5, 0.05, 200, 8.0
    m = [3.0, 2.5, 2.0, 3.5][:nS]
    rng = np.random.default_rng(nS)
    x0 = rng.uniform(0.8, 1.6, (nS, batch))
    tspan = ocs.linspace(0, T, N + 1)
    prob = ocs.LogisticProblem(m, c, r, [[0.0, 1.0]])
    g = ocs.RK4Integrator(tspan)
    s = ocs.fb_sweep_batch(prob, x0, tspan, {"nERROR_PTS": N + 1, "nINTERP_PTS": N + 1}, integrator=g)
    ok = s["sweeps"] > 0
    assert ok.mean() > 0.75   # (with four states a few instances do not converge in 20 sweeps)
    t = np.asarray(tspan)
    u_cc = np.clip(np.sum(s["lam"], axis=0) * np.exp(r * t)[:, None] / (2 * c), 0.0, 1.0)       # [N+1][batch]
    assert np.max(np.abs(s["u"][0][:, ok] - u_cc[:, ok])) < 1e-13
    assert np.all(s["lam"][:, -1, :][:, ok] == 0.0)
    mc = s["maxChange"]
    for b in np.flatnonzero(ok)[:16]:
        k = s["sweeps"][b]
        assert mc[k - 1, b] <= 1.0 and np.all(mc[:k - 1, b] > 1.0) and np.all(np.isnan(mc[k:, b]))
    # the control the solution implies, on the 2N+1 grid (midpoints: ControlChar of the pchip midpoints of lam), through
    # compute_x_lam: x and lam come back to within the sweep's tolerance (uRelTol = 1e-3 on the control)
    from scipy.interpolate import PchipInterpolator
    tg = np.asarray(g.t)
    bsel = [int(i) for i in np.flatnonzero(ok)[:4]]
    ug = np.empty((1, 2 * N + 1, len(bsel)))
    for j, b in enumerate(bsel):
        lam_g = np.stack([PchipInterpolator(t, s["lam"][k, :, b])(tg) for k in range(nS)])
        ug[0, :, j] = np.clip(lam_g.sum(axis=0) * np.exp(r * tg) / (2 * c), 0.0, 1.0)
    x2, lam2 = ocs.compute_x_lam(prob, x0[:, bsel], tspan, ug, integrator=ocs.RK4Integrator(tspan))
    assert np.max(np.abs(x2 - s["x"][:, :, bsel])) < 2e-2 and np.max(np.abs(lam2 - s["lam"][:, :, bsel])) < 2e-2


@pytest.mark.parametrize("nS,batch", [(1, 40000), (2, 24000), (4, 10000)])
def test_two_kernel_sweep_beyond_512_workgroups(ocs, oracle, nS, batch):
    """Until round 4 the folded sweep stopped at 512 workgroups of 64/nS instances; beyond it the loop ran the lane kernels.  The
    fold at 625-750 workgroups: same sweep counts and solution as the kernel-by-kernel sequence (fused_update_off = 1) for every
    instance, and the oracle on sampled instances."""
    rng = np.random.default_rng(batch)
    N = 64
    m = [3.0, 2.5, 2.0, 1.5][:nS]
    tspan = oracle.linspace(0, 1.0, N + 1)
    x0 = rng.uniform(0.8, 2.0, (nS, batch))
    cs = rng.uniform(1.0, 2.0, batch)
    prob = ocs.LogisticProblem(m, P["c"], P["r"], BOUNDS)
    prob.set_batch_params([0], cs[None, :])
    base = {"nERROR_PTS": N + 1, "nINTERP_PTS": 17, "nSWEEPS": 40}
    ga, gb = ocs.RK4Integrator(tspan), ocs.RK4Integrator(tspan)
    ra = ocs.fb_sweep_batch(prob, x0, tspan, dict(base), integrator=ga)
    rb = ocs.fb_sweep_batch(prob, x0, tspan, dict(base, fused_update_off=1), integrator=gb)
    assert ocs.fb_sweep_path(ga) == 4 and ocs.fb_sweep_path(gb) == 1 and batch * nS // 64 > 512
    assert np.array_equal(ra["sweeps"], rb["sweeps"]) and ra["sweeps"].min() > 0
    for key in ("x", "lam", "u", "J"):
        assert relerr(ra[key], rb[key]) < 1e-12, key
    for b in (0, batch // 3, batch - 1):
        ref = oracle.fb_sweep(oracle.LogisticProblem(m, cs[b], P["r"], BOUNDS), x0[:, b], tspan, base)
        assert ra["sweeps"][b] == ref["_sweeps"] and abs(ra["J"][b] - ref["J"]) < RTOL * abs(ref["J"])
        assert relerr(ra["x"][:, :, b], ref["x"]) < RTOL and relerr(ra["u"][:, :, b], ref["u"]) < RTOL
