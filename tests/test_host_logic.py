"""CPU tests of the product's host-side logic (no GPU compute): basis matrices, initial v,
NLP bounds, uFunc and vectorInterpolant sampling of libocs against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ocs():
    import __graft_entry__ as g
    g.build()
    return g.load_package()


def _grid(oracle, N=500, T=10.0):
    return oracle.RK4Integrator(oracle.linspace(0.0, T, N + 1)).t


def test_basis_matrices_bitwise_equal_to_oracle(ocs, oracle):
    t = _grid(oracle)
    for cls, ocl, n in ((ocs.PWLinearControl, oracle.PWLinearControl, 101),
                        (ocs.PWLinearControl, oracle.PWLinearControl, 2),
                        (ocs.PWConstantControl, oracle.PWConstantControl, 50),
                        (ocs.PWConstantControl, oracle.PWConstantControl, 1),
                        (ocs.ChebyshevControl, oracle.ChebyshevControl, 16),
                        (ocs.ChebyshevControl, oracle.ChebyshevControl, 1)):
        a, b = cls(t, n, 2), ocl(t, n, 2)
        assert np.array_equal(a.B, b.B), (cls.__name__, n)
        assert np.array_equal(a._pts, b._pts)
    # non-uniform integrator grid
    tn = oracle.RK4Integrator(np.sort(np.concatenate([[0, 3.0], np.random.default_rng(0).uniform(0, 3, 30)]))).t
    assert np.array_equal(ocs.PWLinearControl(tn, 7, 1).B, oracle.PWLinearControl(tn, 7, 1).B)


def test_initial_v_bounds_ufunc(ocs, oracle):
    t = _grid(oracle, 40, 4.0)
    rng = np.random.default_rng(2)
    for cls, ocl, n in ((ocs.PWLinearControl, oracle.PWLinearControl, 9), (ocs.PWConstantControl, oracle.PWConstantControl, 8),
                        (ocs.ChebyshevControl, oracle.ChebyshevControl, 6)):
        a, b = cls(t, n, 2), ocl(t, n, 2)
        assert np.array_equal(a.compute_initial_v([0.25, 0.5]), b.compute_initial_v([0.25, 0.5]))
        v = rng.normal(size=2 * n)
        tq = np.concatenate([t, rng.uniform(t[0], t[-1], 50)])
        assert np.array_equal(a.compute_uFunc(v)(tq), b.compute_uFunc(v)(tq))
    a, b = ocs.PWLinearControl(t, 9, 2), oracle.PWLinearControl(t, 9, 2)
    bounds = [[0.0, 1.0], [-2.0, 3.0]]
    for x, y in zip(a.compute_nlp_bounds(bounds), b.compute_nlp_bounds(bounds)):
        assert np.array_equal(x, y)
    full = rng.normal(size=18)
    assert np.array_equal(a.compute_initial_v(full), full)
    with pytest.raises(ocs.OcsError):
        a.compute_initial_v([1.0, 2.0, 3.0])
    assert not hasattr(ocs.ChebyshevControl(t, 4, 1), "compute_nlp_bounds")  # ChebyshevControl.m has none
    with pytest.raises(ocs.OcsError):
        ocs.PWLinearControl(t, 1, 1)


def test_vector_interpolant_matches_oracle(ocs, oracle):
    rng = np.random.default_rng(5)
    x = np.sort(rng.uniform(0, 5, 30))
    v = np.vstack([np.sin(2 * x), np.cumsum(rng.normal(size=30)), np.where(x > 2, 1.0, 0.0)])
    q = np.concatenate([x, rng.uniform(x[0], x[-1], 200)])
    for name, m in (("pchip", oracle.INTERP_PCHIP), ("linear", oracle.INTERP_LINEAR), ("previous", oracle.INTERP_PREVIOUS)):
        got = ocs.vectorInterpolant(x, v, name)(q)
        ref = oracle.vector_interp(x, v, m, q)
        assert np.array_equal(got, ref), name
    # the other sample-picking methods of griddedInterpolant (no caller in the reference; vectorInterpolant.m:4 passes any
    # method through): SciPy's interp1d inside the grid, the documented rules at ties and outside it
    from scipy.interpolate import interp1d
    qi = np.concatenate([x, rng.uniform(x[0], x[-1], 200), 0.5 * (x[:-1] + x[1:])])
    for name, kind in (("nearest", "nearest-up"), ("next", "next")):
        gotn = ocs.vectorInterpolant(x, v, name)(qi)
        refn = interp1d(x, v, kind=kind, axis=1)(qi)
        ties = np.isin(qi, 0.5 * (x[:-1] + x[1:])) & (name == "nearest")    # (rounded midpoints: either neighbour is "nearest")
        assert np.array_equal(gotn[:, ~ties], refn[:, ~ties]), name
    out = np.array([x[0] - 1.0, x[-1] + 1.0])
    assert np.array_equal(ocs.vectorInterpolant(x, v, "nearest")(out), v[:, [0, -1]])
    nx = ocs.vectorInterpolant(x, v, "next")(out)
    assert np.array_equal(nx[:, 0], v[:, 0]) and np.all(np.isnan(nx[:, 1]))
    assert np.array_equal(ocs.vectorInterpolant([0.0, 1.0, 3.0], [[1.0, 2.0, 4.0]], "nearest")([0.5, 2.0, 0.25]), [[2.0, 4.0, 1.0]])
    one = ocs.vectorInterpolant(x, v[0], "pchip")(q)  # nCOMPONENTS == 1 branch of vectorInterpolant.m:3-4
    assert one.shape == (1, q.size) and np.array_equal(one[0], got[0] * 0 + ocs.vectorInterpolant(x, v[:1], "pchip")(q)[0])
    assert np.array_equal(ocs.heval(ocs.vectorInterpolant(x, v, "linear"), q, [2, 0]),
                          oracle.vector_interp(x, v, oracle.INTERP_LINEAR, q)[[2, 0]])


def test_user_problem_sources_compile_for_gfx950(ocs):
    """hipRTC path: user plugin source + kernel templates compile without a GPU;
    a broken plugin reports the compiler's message."""
    from tests.user_problems import LOGISTIC2_SRC, PREDPREY_SRC, lq_source
    ocs.UserProblem.check_source(LOGISTIC2_SRC, 2, 1, 4, has_control_char=True)
    ocs.UserProblem.check_source(PREDPREY_SRC, 2, 1, 8)
    ocs.UserProblem.check_source(lq_source(6, 2), 6, 2, 1 + 36 + 12 + 6 + 2)   # > 16 parameters: uniform block
    # every other form of a plugin, each with the kernel templates it selects: row functions (state pass, scan adjoint, costate
    # scan reading u), + ControlChar of the costate alone (two-kernel sweep) with the tabulated ControlChar coefficient,
    # ControlChar that reads x, full-vector methods with the tabulated time coefficient (vector mappings, costate vscan)
    from tests.user_problems import (LOGISTIC_ROWS_SRC, LOGISTIC_ROWS_CC_SRC, LOGISTIC_ROWS_CCT_SRC, PROPHARVEST_ROWS_CC_SRC,
                                     PREDPREY_TC_SRC)
    ocs.UserProblem.check_source(LOGISTIC_ROWS_SRC, 4, 1, 6, row_separable=True)
    ocs.UserProblem.check_source(LOGISTIC_ROWS_CC_SRC, 1, 1, 3, has_control_char=True, row_separable=True, control_from_costate=True)
    ocs.UserProblem.check_source(LOGISTIC_ROWS_CCT_SRC, 2, 1, 4, has_control_char=True, row_separable=True, control_from_costate=True)
    ocs.UserProblem.check_source(PROPHARVEST_ROWS_CC_SRC, 2, 1, 4, has_control_char=True, row_separable=True)
    ocs.UserProblem.check_source(PREDPREY_TC_SRC, 2, 1, 8)
    with pytest.raises(ocs.OcsError):   # the declaration needs row functions and ocs_ControlChar
        ocs.UserProblem(LOGISTIC2_SRC, 2, 1, [1.5, 0.05, 3.0, 2.5], [[0.0, 1.0]], has_control_char=True, control_from_costate=True)
    with pytest.raises(ocs.OcsError) as e:
        ocs.UserProblem.check_source("__device__ void ocs_F(double t) { syntax error }", 1, 1, 0)
    assert e.value.code == -1 and "error" in str(e.value)


def test_registry_and_option_validation_without_gpu(ocs):
    """Handle creation and argument checks of the C-ABI that need no device: the LQ registry entry (BASELINE config 5),
    the fusion switch of the control bases, MATLAB's linspace formula, the forward-backward-sweep option block."""
    rng = np.random.default_rng(0)
    A, Bu = rng.normal(size=(7, 7)), rng.normal(size=(7, 3))
    p = ocs.LQProblem(A, Bu, np.ones(7), np.ones(3), 0.05, [[-1, 1]] * 3)
    assert (p.nS, p.nC, p.nAug) == (7, 3, 8)
    for nS, nC in ((33, 2), (4, 5)):          # no kernel instantiated: refused at creation, not at launch
        with pytest.raises(ocs.OcsError) as e:
            ocs.LQProblem(rng.normal(size=(nS, nS)), rng.normal(size=(nS, nC)), np.ones(nS), np.ones(nC), 0.05,
                          [[-1, 1]] * nC)
        assert e.value.code == -6
    from ocs_amd import _lib
    import oracle.oracle as orc0
    t = _grid(orc0, 40, 4.0)
    c = ocs.ChebyshevControl(t, 6, 1)
    for mode in ("auto", "off", "on", "lane"):
        assert c.set_fusion(mode) is c
    assert _lib.lib.ocs_control_set_fusion(c._h, 7) == -1      # OCS_ERR_INVALID
    # MATLAB linspace: end points pinned, (k*(b-a))/(n-1) rounding (differs from numpy's start + k*step)
    import oracle.oracle as orc
    for a, b, n in ((0.0, 10.0, 1001), (0.3, 7.7, 17), (5.0, 5.0, 3), (-2.0, 1.0, 2), (1.0, 2.0, 1)):
        assert np.array_equal(ocs.linspace(a, b, n), orc.linspace(a, b, n))
    o = _lib.FbsOptions()
    assert _lib.lib.ocs_fbs_default_options(C.byref(o)) == 0
    assert (o.uRelTol, o.uAbsTol, o.nSWEEPS, o.nERROR_PTS, o.nINTERP_PTS, o.fused_update_off) == (1e-7, 1e-7, 50, 1001, 1001, 0)
    so = _lib.SsOptions()                       # single_shooting.m:20-21 defaults of the batched shooting driver
    assert _lib.lib.ocs_ss_default_options(C.byref(so)) == 0 and _lib.lib.ocs_ss_default_options(None) == -1
    assert (so.TolX, so.TolFun, so.MaxIter, so.memory, so.maxBacktracks) == (1e-5, 3e-4, 500, 10, 25)


def test_multi_device_block_arithmetic_matches_the_python_sharding():
    """include/ocs.h ocs_multi_shard cuts a batch as distributed.shard_bounds does (contiguous blocks, sizes differ by at
    most one): the C restatement in csrc/ocs_multi.cpp against the Python one, without a device (the handle cannot be
    created without a GPU, so the arithmetic is restated here from the header's contract and checked for every rank)."""
    import __graft_entry__ as g
    ocs = g.load_package()
    for total in (1, 7, 8, 9, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            covered = []
            for k in range(world):
                lo, hi = ocs.distributed.shard_bounds(total, world, k)
                base, rem = divmod(total, world)
                assert (lo, hi) == (k * base + min(k, rem), k * base + min(k, rem) + base + (1 if k < rem else 0))
                covered += list(range(lo, hi))
            assert covered == list(range(total))


def test_compiled_plugins_are_cached_on_disk(tmp_path, monkeypatch):
    """ocs_problem_check_source compiles with hipRTC (no GPU needed); the code object lands in OCS_JIT_CACHE_DIR keyed by the
    generated source, this build's kernel headers and the kernel names, and a second process-level compile is a file read.
    A damaged file is ignored (checksum) and rewritten; OCS_JIT_CACHE=0 switches the cache off."""
    import subprocess
    import sys
    import time
    code = (
        "import sys, time; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as g; ocs = g.load_package()\n"
        "from tests.user_problems import PREDPREY_SRC\n"
        "t0 = time.time(); ocs.UserProblem.check_source(PREDPREY_SRC + '// %%s\\n' %% sys.argv[1], 2, 1, 8); print(time.time() - t0)\n"
    ) % __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    env = dict(__import__("os").environ, OCS_JIT_CACHE_DIR=str(tmp_path))

    def run(tag, extra=None):
        e = dict(env, **(extra or {}))
        out = subprocess.run([sys.executable, "-c", code, tag], env=e, capture_output=True, text=True, check=True)
        return float(out.stdout.strip().splitlines()[-1])
    t_first = run("a")
    files = list(tmp_path.glob("*.ocsjit"))
    assert len(files) == 1 and files[0].stat().st_size > 10000
    t_second = run("a")
    assert t_second < 0.2 * t_first and t_second < 0.5
    files[0].write_bytes(files[0].read_bytes()[:-100])          # truncated: rejected, compiled again, rewritten
    assert run("a") > 5 * t_second and files[0].stat().st_size > 10000
    run("b", {"OCS_JIT_CACHE": "0"})
    assert len(list(tmp_path.glob("*.ocsjit"))) == 1            # another source with the cache off: nothing new
