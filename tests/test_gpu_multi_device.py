"""The multi-device entry points of the C-ABI (include/ocs.h ocs_multi_*) on the one GPU a test box has:
one communicator over one device (ncclCommInitAll), every entry point against the one-device entry point of the same
name (bit-equal: the same kernels on the same block), the RCCL reductions against numpy.  N > 1 devices cannot be run
here; the block arithmetic that N > 1 adds is covered without a GPU in tests/test_host_logic.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0
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
    assert st["count"] == B and st["argmin"] == int(np.argmin(Jr)) and abs(st["min_J"] - Jr.min()) < 1e-13 * abs(Jr.min())
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


def _device_tensors(torch, arrays, dev):
    return [torch.tensor(np.ascontiguousarray(a), device=dev) for a in arrays]


def test_multi_device_dev_entry_points_equal_single_device(ocs, oracle):
    """ocs_multi_*_dev with one device: device-resident blocks, asynchronous on the handle's stream, reductions enqueued
    behind the kernels -- bit-equal to the one-device _dev entry points; the caller's current device is untouched."""
    import torch
    dev = torch.device("cuda:0")
    md = ocs.MultiDevice([0])
    assert md.has_communicator and md.stream(0) != 0
    N, B, nB = 64, 320, 12
    tspan = oracle.linspace(0, 4, N + 1)
    rng = np.random.default_rng(5)
    integs = md.replicate(lambda: ocs.RK4Integrator(tspan))
    probs = md.replicate(lambda: ocs.LogisticProblem([3.0, 2.5], P["c"], P["r"], BOUNDS))
    g1, p1 = ocs.RK4Integrator(tspan), ocs.LogisticProblem([3.0, 2.5], P["c"], P["r"], BOUNDS)
    x0 = torch.tensor(rng.uniform(0.8, 1.6, (2, B)), device=dev)
    u = torch.tensor(rng.uniform(0.0, 0.6, (2 * N + 1, 1, B)), device=dev)
    x, lam = (torch.empty((N + 1, 3, B), dtype=torch.float64, device=dev) for _ in range(2))
    J, dJdu = torch.empty(B, dtype=torch.float64, device=dev), torch.empty_like(u)
    torch.cuda.synchronize()
    rc = md.compute_states_dev(integs, probs, [x0], [u], [x], [J], reduce=True)
    md.compute_adjoints_dev(integs, probs, [u], [lam], [dJdu])
    st = md.stats()
    md.synchronize()
    xr, lamr, dr = torch.empty_like(x), torch.empty_like(lam), torch.empty_like(dJdu)
    _, Jr = g1.compute_states_dev(p1, x0, u, xr)
    g1.compute_adjoints_dev(p1, u, None, lamr, dr)
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(x, xr) and torch.equal(J, Jr) and torch.equal(lam, lamr) and torch.equal(dJdu, dr)
    Jh = Jr.cpu().numpy()
    assert st["count"] == B and st["argmin"] == int(np.argmin(Jh)) and st["min_J"] == Jh.min()
    assert abs(st["sum_J"] - Jh.sum()) < 1e-12 * abs(Jh.sum())
    with pytest.raises(ocs.OcsError):
        md.stats()   # nothing enqueued since the last fetch
    # nlpObjective on device-resident coefficient blocks
    ctrls = md.replicate(lambda: ocs.ChebyshevControl(integs[0].t, nB, 1))
    c1 = ocs.ChebyshevControl(g1.t, nB, 1)
    V = 0.05 * rng.normal(size=(nB, B)) / np.arange(1, nB + 1)[:, None]
    V[0] += 0.4
    Vd = torch.tensor(V, device=dev)
    Jm, Gm = torch.empty(B, dtype=torch.float64, device=dev), torch.empty_like(Vd)
    Js, Gs = torch.empty_like(Jm), torch.empty_like(Vd)
    torch.cuda.synchronize()
    md.nlp_objective_dev(integs, probs, ctrls, [x0.clone()], [Vd], [Jm], [Gm], reduce=True)
    stn = md.stats()
    ocs.nlp_objective_dev(g1, p1, c1, x0.clone(), Vd, (), Js, Gs)
    torch.cuda.synchronize()
    assert torch.equal(Jm, Js) and torch.equal(Gm, Gs) and stn["argmin"] == int(torch.argmin(Js))
    # fb_sweep on a device block; statistics over the converged instances only
    pb = md.replicate(lambda: ocs.LogisticProblem([3.0], P["c"], P["r"], [[-0.2, 6.0]]))
    ps = ocs.LogisticProblem([3.0], P["c"], P["r"], [[-0.2, 6.0]])
    ts2 = oracle.linspace(0, 4.5, 169)
    ig, gs = md.replicate(lambda: ocs.RK4Integrator(ts2)), ocs.RK4Integrator(ts2)
    x0b = torch.tensor(rng.uniform(0.8, 1.6, (1, 128)), device=dev)
    opt = {"nERROR_PTS": 169, "nINTERP_PTS": 17, "nSWEEPS": 30}
    rm = md.fb_sweep_dev(ig, pb, [x0b], opt, reduce=True)[0]
    stf = md.stats()
    rs = ocs.fb_sweep_dev(ps, gs, x0b, opt)
    torch.cuda.synchronize()
    conv = (rs["sweeps"] > 0).cpu().numpy()
    assert torch.equal(rm["sweeps"], rs["sweeps"]) and conv.any() and (~conv).any()
    for k in ("xaug", "lam", "u"):
        assert np.array_equal(rm[k].cpu().numpy()[:, :1, conv], rs[k].cpu().numpy()[:, :1, conv])
    Jc = rs["J"].cpu().numpy()[conv]
    assert stf["count"] == int(conv.sum()) and stf["argmin"] == int(np.flatnonzero(conv)[np.argmin(Jc)])
    assert torch.cuda.current_device() == 0


def test_multi_device_reductions_on_the_host_without_a_communicator(ocs, oracle, monkeypatch):
    """OCS_MULTI_NO_RCCL=1: no communicator is created, the same four numbers come from the host branch."""
    monkeypatch.setenv("OCS_MULTI_NO_RCCL", "1")
    md = ocs.MultiDevice([0])
    assert not md.has_communicator
    N, B = 32, 100
    tspan = oracle.linspace(0, 2, N + 1)
    rng = np.random.default_rng(6)
    integs = md.replicate(lambda: ocs.RK4Integrator(tspan))
    probs = md.replicate(lambda: ocs.TestOCProblem(P, BOUNDS))
    x0, u = rng.uniform(0.8, 1.6, (1, B)), rng.uniform(0.0, 0.6, (1, 2 * N + 1, B))
    u[:, :, 11] = np.nan
    _, J, st, rc = md.compute_states(integs, probs, x0, u, want_x=False)
    ok = np.isfinite(J)
    assert rc == 1 and st["count"] == B - 1 and st["argmin"] == int(np.flatnonzero(ok)[np.argmin(J[ok])])
    assert st["min_J"] == J[ok].min() and abs(st["sum_J"] - J[ok].sum()) < 1e-12 * abs(J[ok].sum())


@pytest.mark.parametrize("nslots", [2, 3])
def test_multi_device_several_slots_on_one_gpu(ocs, oracle, monkeypatch, nslots):
    """The N > 1 code path on a one-GPU box: with OCS_MULTI_ALLOW_DUPLICATES=1 device 0 is listed several times, so the
    persistent worker threads, the block arithmetic (unequal blocks), the per-slot handles and the reductions (on the host:
    RCCL refuses a duplicate device) all run -- host and device entry points against the one-device results."""
    import torch
    monkeypatch.setenv("OCS_MULTI_ALLOW_DUPLICATES", "1")
    md = ocs.MultiDevice([0] * nslots)
    assert md.size == nslots and not md.has_communicator
    dev = torch.device("cuda:0")
    N, B = 48, 203   # 203 = 68 + 68 + 67 / 102 + 101
    tspan = oracle.linspace(0, 3, N + 1)
    rng = np.random.default_rng(7 + nslots)
    integs = md.replicate(lambda: ocs.RK4Integrator(tspan))
    probs = md.replicate(lambda: ocs.LogisticProblem([3.0, 2.5], P["c"], P["r"], BOUNDS))
    g1, p1 = ocs.RK4Integrator(tspan), ocs.LogisticProblem([3.0, 2.5], P["c"], P["r"], BOUNDS)
    x0, u = rng.uniform(0.8, 1.6, (2, B)), rng.uniform(0.0, 0.6, (1, 2 * N + 1, B))
    for rep in range(3):   # the workers are reused from call to call
        x, J, st, rc = md.compute_states(integs, probs, x0, u)
        lam, dJdu = md.compute_adjoints(integs, probs, u)
    xr, Jr = g1.compute_states(p1, x0, u)
    lamr, dr = g1.compute_adjoints(p1, u)
    # (a block and the whole batch may take different kernel mappings -- an even block of 102 trajectories runs the tiled state
    #  pass with an overlapping last tile, the odd batch of 203 does not --: equal to round-off, not bit for bit)
    assert rc == 0 and relerr(x, xr) < 1e-13 and relerr(J, Jr) < 1e-13 and relerr(lam, lamr) < 1e-13 and relerr(dJdu, dr) < 1e-13
    assert st["count"] == B and st["argmin"] == int(np.argmin(Jr)) and abs(st["min_J"] - Jr.min()) < 1e-13 * abs(Jr.min())
    assert abs(st["sum_J"] - Jr.sum()) < 1e-12 * abs(Jr.sum())
    # device-resident blocks of unequal size
    bounds = [md.shard(B, k) for k in range(nslots)]
    assert bounds[0][0] == 0 and bounds[-1][1] == B and all(b[1] == c[0] for b, c in zip(bounds, bounds[1:]))
    ub = np.ascontiguousarray(u.transpose(1, 0, 2))   # [2N+1][nC][B]
    x0d = [torch.tensor(np.ascontiguousarray(x0[:, lo:hi]), device=dev) for lo, hi in bounds]
    ud = [torch.tensor(np.ascontiguousarray(ub[:, :, lo:hi]), device=dev) for lo, hi in bounds]
    xd = [torch.empty((N + 1, 3, hi - lo), dtype=torch.float64, device=dev) for lo, hi in bounds]
    ld = [torch.empty_like(t) for t in xd]
    Jd = [torch.empty(hi - lo, dtype=torch.float64, device=dev) for lo, hi in bounds]
    dd = [torch.empty_like(t) for t in ud]
    torch.cuda.synchronize()
    md.compute_states_dev(integs, probs, x0d, ud, xd, Jd, reduce=True)
    md.compute_adjoints_dev(integs, probs, ud, ld, dd)
    st2 = md.stats()
    md.synchronize()
    assert relerr(np.concatenate([t.cpu().numpy() for t in Jd]), Jr) < 1e-13
    assert relerr(np.concatenate([t.cpu().numpy() for t in dd], axis=2), dr.transpose(1, 0, 2)) < 1e-13
    assert relerr(np.concatenate([t.cpu().numpy() for t in ld], axis=2), lamr.transpose(1, 0, 2)) < 1e-13
    assert st2["argmin"] == int(np.argmin(Jr)) and st2["count"] == B
    # an error on one slot comes back with its device named, and the workers survive it
    with pytest.raises(ocs.OcsError):
        md.compute_states_dev(integs, probs, x0d, ud[:1] + [None] * (nslots - 1), xd, Jd)
    md.compute_states_dev(integs, probs, x0d, ud, xd, Jd)
    md.synchronize()
    assert torch.cuda.current_device() == 0


def test_device_buffer_helpers_and_resident_loop_on_two_slots(ocs, oracle, monkeypatch):
    """ocs_device_malloc / _upload / _download / _free (what a MATLAB host uses to keep blocks on the devices) and the loop of
    single_shooting.m:114,137-150 on resident blocks over two slots: nlpObjective evaluated repeatedly with the iterate updated
    on the devices, nothing but the four statistics coming back; fb_sweep on resident blocks of two slots."""
    import ctypes as C

    import torch
    lib = ocs._lib.lib
    h = np.arange(12, dtype=np.float64) * 1.5
    back = np.zeros_like(h)
    p = C.c_void_p()
    assert lib.ocs_device_malloc(C.byref(p), h.nbytes) == 0 and p.value
    assert lib.ocs_device_upload(p, h.ctypes.data_as(C.c_void_p), h.nbytes, None) == 0
    assert lib.ocs_device_download(back.ctypes.data_as(C.c_void_p), p, h.nbytes, None) == 0
    assert np.array_equal(h, back) and lib.ocs_device_free(p) == 0
    monkeypatch.setenv("OCS_MULTI_ALLOW_DUPLICATES", "1")
    md = ocs.MultiDevice([0, 0])
    dev = torch.device("cuda:0")
    N, nB, B = 64, 8, 150
    tspan = oracle.linspace(0, 4, N + 1)
    integs = md.replicate(lambda: ocs.RK4Integrator(tspan))
    probs = md.replicate(lambda: ocs.TestOCProblem(P, BOUNDS))
    ctrls = md.replicate(lambda: ocs.ChebyshevControl(integs[0].t, nB, 1))
    g1, p1, c1 = ocs.RK4Integrator(tspan), ocs.TestOCProblem(P, BOUNDS), ocs.ChebyshevControl(integs[0].t, nB, 1)
    rng = np.random.default_rng(11)
    V = 0.05 * rng.normal(size=(nB, B)) / np.arange(1, nB + 1)[:, None]
    V[0] += 0.4
    bounds = [md.shard(B, k) for k in range(2)]
    vb = [torch.tensor(np.ascontiguousarray(V[:, lo:hi]), device=dev) for lo, hi in bounds]
    x0b = [torch.full((1, hi - lo), 1.2, dtype=torch.float64, device=dev) for lo, hi in bounds]
    Jb = [torch.empty(hi - lo, dtype=torch.float64, device=dev) for lo, hi in bounds]
    gb = [torch.empty_like(v) for v in vb]
    Vs, x0s = torch.tensor(V, device=dev), torch.full((1, B), 1.2, dtype=torch.float64, device=dev)
    Js, Gs = torch.empty(B, dtype=torch.float64, device=dev), torch.empty_like(Vs)
    torch.cuda.synchronize()
    for it in range(4):   # steepest descent with a fixed step, the iterate never leaves the device
        md.nlp_objective_dev(integs, probs, ctrls, x0b, vb, Jb, gb, reduce=True)
        st = md.stats()
        ocs.nlp_objective_dev(g1, p1, c1, x0s, Vs, (), Js, Gs)
        torch.cuda.synchronize()
        # (blocks of 75 candidates and the batch of 150 may take different mappings: round-off level)
        assert relerr(torch.cat(Jb).cpu().numpy(), Js.cpu().numpy()) < 1e-13 and relerr(torch.cat(gb, dim=1).cpu().numpy(), Gs.cpu().numpy()) < 1e-12
        assert st["argmin"] == int(torch.argmin(Js)) and st["count"] == B
        for v, g in zip(vb, gb):
            v -= 1e-2 * g
        Vs -= 1e-2 * Gs
        torch.cuda.synchronize()
    # fb_sweep on two resident blocks against one solve of the whole batch
    X0 = rng.uniform(0.8, 1.6, (1, 128))
    ts2 = oracle.linspace(0, 4.5, 169)
    ig = md.replicate(lambda: ocs.RK4Integrator(ts2))
    opt = {"nERROR_PTS": 169, "nINTERP_PTS": 17}
    blocks = [torch.tensor(np.ascontiguousarray(X0[:, lo:hi]), device=dev) for lo, hi in (md.shard(128, 0), md.shard(128, 1))]
    outs = md.fb_sweep_dev(ig, probs, blocks, opt, reduce=True)
    stf = md.stats()
    ref = ocs.fb_sweep_dev(p1, ocs.RK4Integrator(ts2), torch.tensor(X0, device=dev), opt)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([o["sweeps"] for o in outs]), ref["sweeps"]) and bool((ref["sweeps"] > 0).all())
    assert torch.equal(torch.cat([o["J"] for o in outs]), ref["J"]) and torch.equal(torch.cat([o["lam"] for o in outs], dim=2), ref["lam"])
    assert stf["count"] == 128 and stf["argmin"] == int(torch.argmin(ref["J"]))
