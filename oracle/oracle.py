"""ctypes front-end of the CPU oracle (oracle/ocs_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.

PARITY UNPINNED: see oracle/ocs_oracle.h.  The classes mirror the reference's MATLAB
classes one-to-one (TestOCProblem, RK4Integrator, RK4InfiniteIntegrator,
PWLinearControl, PWConstantControl, ChebyshevControl) so that tests read like
tests/backprop_test.m / tests/solve_test_problem.m.
All arrays are numpy float64, Fortran (column-major) order, MATLAB shapes.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

PROBLEM_TEST, PROBLEM_LOGISTIC, PROBLEM_LQ = 1, 2, 3
CONTROL_PWLINEAR, CONTROL_PWCONSTANT, CONTROL_CHEBYSHEV = 1, 2, 3
INTERP_LINEAR, INTERP_LINEAR_NEAREST, INTERP_PREVIOUS, INTERP_PCHIP = 0, 1, 2, 3


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("ocs_oracle.c", "ocs_oracle.h", "Makefile", "ocs_cpu_fast.c")]
    fast = os.path.join(_HERE, "_build", "libcpufast.so")
    stale = (not os.path.exists(_LIB_PATH)) or (not os.path.exists(fast)) or any(
        os.path.getmtime(s) > min(os.path.getmtime(_LIB_PATH), os.path.getmtime(fast)) for s in src
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B" if force else "-s"], check=True)
    return _LIB_PATH


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f(a, shape=None):
    a = np.asfortranarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = np.asfortranarray(a.reshape(shape, order="F"))
    return a


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.ocs_or_problem_create.restype = C.c_void_p
        L.ocs_or_problem_create.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp]
        L.ocs_or_problem_destroy.argtypes = [C.c_void_p]
        for name in ("ocs_or_F", "ocs_or_stateRHS", "ocs_or_objective"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]
        L.ocs_or_dFdx_times_vec.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.ocs_or_dFdu_times_vec.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.ocs_or_adjointRHS.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.ocs_or_ControlChar.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]
        L.ocs_or_linspace.argtypes = [C.c_double, C.c_double, C.c_int, _dp]
        L.ocs_or_interp1.argtypes = [C.c_int, _dp, _dp, C.c_int, C.c_int, _dp, _dp]
        L.ocs_or_pchip_slopes.argtypes = [C.c_int, _dp, _dp, _dp]
        L.ocs_or_vector_interp.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, _dp, _dp]
        L.ocs_or_rk4_create.restype = C.c_void_p
        L.ocs_or_rk4_create.argtypes = [_dp, C.c_int]
        L.ocs_or_rk4_destroy.argtypes = [C.c_void_p]
        L.ocs_or_rk4_nsteps.argtypes = [C.c_void_p]
        for name in ("ocs_or_rk4_t", "ocs_or_rk4_h", "ocs_or_rk4_xK", "ocs_or_rk4inf_t"):
            getattr(L, name).restype = _dp
            getattr(L, name).argtypes = [C.c_void_p]
        L.ocs_or_rk4_compute_states.argtypes = [C.c_void_p, C.c_void_p, _dp, _dp, _dp, _dp]
        L.ocs_or_rk4_compute_adjoints.argtypes = [C.c_void_p, C.c_void_p, _dp, _dp, _dp, _dp]
        L.ocs_or_rk4inf_create.restype = C.c_void_p
        L.ocs_or_rk4inf_create.argtypes = [_dp, C.c_int, _dp, C.c_int, _dp, C.c_int]
        L.ocs_or_rk4inf_destroy.argtypes = [C.c_void_p]
        L.ocs_or_rk4inf_nsteps.argtypes = [C.c_void_p]
        L.ocs_or_rk4inf_compute_states.argtypes = [C.c_void_p, C.c_void_p, _dp, _dp, _dp, _dp]
        L.ocs_or_rk4inf_compute_adjoints.argtypes = [C.c_void_p, C.c_void_p, _dp, _dp, _dp]
        L.ocs_or_control_create.restype = C.c_void_p
        L.ocs_or_control_create.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_int]
        L.ocs_or_control_destroy.argtypes = [C.c_void_p]
        L.ocs_or_control_B.restype = _dp
        L.ocs_or_control_B.argtypes = [C.c_void_p]
        L.ocs_or_control_pts.restype = _dp
        L.ocs_or_control_pts.argtypes = [C.c_void_p]
        L.ocs_or_control_compute_u.argtypes = [C.c_void_p, _dp, _dp]
        L.ocs_or_control_compute_dJdv.argtypes = [C.c_void_p, _dp, _dp]
        L.ocs_or_control_compute_initial_v.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        L.ocs_or_control_compute_nlp_bounds.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.ocs_or_control_eval_uFunc.argtypes = [C.c_void_p, _dp, C.c_int, _dp, _dp]
        L.ocs_or_nlp_objective.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _dp, _dp,
                                           C.c_int, _ip, _dp, _dp]
        L.ocs_or_fbs_default_options.argtypes = [C.c_void_p]
        L.ocs_or_compute_x_lam.argtypes = [C.c_void_p, C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        L.ocs_or_control_from_x_lam.argtypes = [C.c_void_p, C.c_void_p, _dp, _dp, C.c_int, _dp, _dp]
        L.ocs_or_fb_sweep.restype = C.c_int
        L.ocs_or_fb_sweep.argtypes = [C.c_void_p, C.c_void_p, _dp, C.c_void_p, _dp, _dp, _dp, _dp,
                                      _dp, _dp, _dp]
        L.ocs_or_max_threads.restype = C.c_int
        L.ocs_or_batch_states_adjoints.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp, _dp,
                                                   C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                                   C.c_int]
        _lib = L
    return _lib


class FbsOptions(C.Structure):
    _fields_ = [("uRelTol", C.c_double), ("uAbsTol", C.c_double), ("nSWEEPS", C.c_int),
                ("nERROR_PTS", C.c_int), ("nINTERP_PTS", C.c_int), ("uRelax", C.c_double)]


def linspace(a, b, n):
    out = np.empty(n)
    lib().ocs_or_linspace(a, b, n, _p(out))
    return out


def interp1(x, v, method, xq):
    x, v, xq = _f(x), _f(v), _f(xq)
    out = np.empty(xq.size)
    lib().ocs_or_interp1(x.size, _p(x), _p(v), method, xq.size, _p(xq), _p(out))
    return out


def pchip_slopes(x, y):
    x, y = _f(x), _f(y)
    d = np.empty(x.size)
    lib().ocs_or_pchip_slopes(x.size, _p(x), _p(y), _p(d))
    return d


def vector_interp(x, v, method, tq):
    """functions/vectorInterpolant.m: v is nComp x n."""
    x, tq = _f(x), _f(tq)
    v = _f(np.atleast_2d(v))
    out = np.empty((v.shape[0], tq.size), order="F")
    lib().ocs_or_vector_interp(v.shape[0], x.size, _p(x), _p(v), method, tq.size, _p(tq), _p(out))
    return out


class Problem:
    """OCProblem/OCProblem.m plugin; concrete problems from the registry."""

    def __init__(self, pid, nS, nC, params, bounds):
        self.id, self.nS, self.nC, self.nAug = pid, nS, nC, nS + 1
        self.params = _f(params).ravel()
        self.ControlBounds = _f(bounds, (nC, 2))
        self._h = lib().ocs_or_problem_create(pid, nS, nC, _p(self.params), self.params.size,
                                              _p(self.ControlBounds))
        if not self._h:
            raise ValueError("bad problem definition")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ocs_or_problem_destroy(self._h)
            self._h = None

    def _tk(self, t):
        t = _f(np.atleast_1d(t)).ravel()
        return t, t.size

    def F(self, t, y, u):
        t, k = self._tk(t)
        y, u = _f(y, (self.nAug, k)), _f(u, (self.nC, k))
        out = np.empty((self.nAug, k), order="F")
        lib().ocs_or_F(self._h, k, _p(t), _p(y), _p(u), _p(out))
        return out

    def dFdx_times_vec(self, t, y, u, v):
        t, k = self._tk(t)
        y, u, v = _f(y, (self.nAug, k)), _f(u, (self.nC, k)), _f(v, (self.nAug, k))
        out = np.empty((self.nAug, k), order="F")
        lib().ocs_or_dFdx_times_vec(self._h, k, _p(t), _p(y), _p(u), _p(v), _p(out))
        return out

    def dFdu_times_vec(self, t, y, u, v):
        t, k = self._tk(t)
        y, u, v = _f(y, (self.nAug, k)), _f(u, (self.nC, k)), _f(v, (self.nAug, k))
        out = np.empty((self.nC, k), order="F")
        lib().ocs_or_dFdu_times_vec(self._h, k, _p(t), _p(y), _p(u), _p(v), _p(out))
        return out

    # Gen-1 fields through the A9 adapter
    def stateRHS(self, t, x, u):
        t, k = self._tk(t)
        x, u = _f(x, (self.nS, k)), _f(u, (self.nC, k))
        out = np.empty((self.nS, k), order="F")
        lib().ocs_or_stateRHS(self._h, k, _p(t), _p(x), _p(u), _p(out))
        return out

    def objective(self, t, x, u):
        t, k = self._tk(t)
        x, u = _f(x, (self.nS, k)), _f(u, (self.nC, k))
        out = np.empty((1, k), order="F")
        lib().ocs_or_objective(self._h, k, _p(t), _p(x), _p(u), _p(out))
        return out

    def adjointRHS(self, t, x, lam, u):
        t, k = self._tk(t)
        x, lam, u = _f(x, (self.nS, k)), _f(lam, (self.nS, k)), _f(u, (self.nC, k))
        out = np.empty((self.nS, k), order="F")
        lib().ocs_or_adjointRHS(self._h, k, _p(t), _p(x), _p(lam), _p(u), _p(out))
        return out

    def ControlChar(self, t, x, lam):
        t, k = self._tk(t)
        x, lam = _f(x, (self.nS, k)), _f(lam, (self.nS, k))
        out = np.empty((self.nC, k), order="F")
        lib().ocs_or_ControlChar(self._h, k, _p(t), _p(x), _p(lam), _p(out))
        return out


def TestOCProblem(p, ControlBounds):
    """tests/TestOCProblem.m:16-20; p is a dict with c, m, r."""
    return Problem(PROBLEM_TEST, 1, 1, [p["c"], p["m"], p["r"]], ControlBounds)


def LogisticProblem(m, c, r, ControlBounds):
    m = np.atleast_1d(np.asarray(m, dtype=np.float64))
    return Problem(PROBLEM_LOGISTIC, m.size, 1, np.concatenate([[c, r], m]), ControlBounds)


def LQProblem(A, Bu, q, rdiag, r, ControlBounds):
    A, Bu = _f(A), _f(np.atleast_2d(Bu))
    nS, nC = A.shape[0], Bu.shape[1]
    par = np.concatenate([[r], A.ravel(order="F"), Bu.ravel(order="F"), np.ravel(q), np.ravel(rdiag)])
    return Problem(PROBLEM_LQ, nS, nC, par, ControlBounds)


class RK4Integrator:
    """Integrator/RK4Integrator.m"""

    def __init__(self, tspan):
        self.tspan = _f(tspan).ravel()
        self._h = lib().ocs_or_rk4_create(_p(self.tspan), self.tspan.size)
        self.nSTEPS = lib().ocs_or_rk4_nsteps(self._h)
        N = self.nSTEPS
        self.t = np.ctypeslib.as_array(lib().ocs_or_rk4_t(self._h), (2 * N + 1,)).copy()
        self.h = np.ctypeslib.as_array(lib().ocs_or_rk4_h(self._h), (N,)).copy()
        self._nAug = None

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ocs_or_rk4_destroy(self._h)
            self._h = None

    @property
    def xK(self):
        N = self.nSTEPS
        a = np.ctypeslib.as_array(lib().ocs_or_rk4_xK(self._h), (self._nAug * (N + 1) * 4,))
        return a.copy().reshape((self._nAug, N + 1, 4), order="F")

    def compute_states(self, prob, x0, u):
        N = self.nSTEPS
        x0 = _f(x0).ravel()
        u = _f(u, (prob.nC, 2 * N + 1))
        x = np.empty((prob.nAug, N + 1), order="F")
        J = C.c_double()
        lib().ocs_or_rk4_compute_states(self._h, prob._h, _p(x0), _p(u), _p(x), C.byref(J))
        self._nAug = prob.nAug
        return x, J.value

    def compute_adjoints(self, prob, u, lamT=None, want_dJdu=True):
        N = self.nSTEPS
        u = _f(u, (prob.nC, 2 * N + 1))
        lam = np.empty((prob.nAug, N + 1), order="F")
        dJdu = np.empty((prob.nC, 2 * N + 1), order="F") if want_dJdu else None
        lt = None if lamT is None else _f(lamT).ravel()
        lib().ocs_or_rk4_compute_adjoints(self._h, prob._h, _p(u), _p(lt), _p(lam), _p(dJdu))
        return (lam, dJdu) if want_dJdu else lam


class RK4InfiniteIntegrator:
    """Integrator/RK4InfiniteIntegrator.m"""

    def __init__(self, tspan, tspanExtra, uStar):
        self.tspan, self.tspanExtra = _f(tspan).ravel(), _f(tspanExtra).ravel()
        self.uStar = _f(np.atleast_1d(uStar)).ravel()
        self._h = lib().ocs_or_rk4inf_create(_p(self.tspan), self.tspan.size, _p(self.tspanExtra),
                                             self.tspanExtra.size, _p(self.uStar), self.uStar.size)
        self.nSTEPS = lib().ocs_or_rk4inf_nsteps(self._h)
        self.t = np.ctypeslib.as_array(lib().ocs_or_rk4inf_t(self._h), (2 * self.nSTEPS + 1,)).copy()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ocs_or_rk4inf_destroy(self._h)
            self._h = None

    def compute_states(self, prob, x0, u):
        N = self.nSTEPS
        x0, u = _f(x0).ravel(), _f(u, (prob.nC, 2 * N + 1))
        x = np.empty((prob.nAug, N + 1), order="F")
        J = C.c_double()
        lib().ocs_or_rk4inf_compute_states(self._h, prob._h, _p(x0), _p(u), _p(x), C.byref(J))
        return x, J.value

    def compute_adjoints(self, prob, u):
        N = self.nSTEPS
        u = _f(u, (prob.nC, 2 * N + 1))
        lam = np.empty((prob.nAug, N + 1), order="F")
        dJdu = np.empty((prob.nC, 2 * N + 1), order="F")
        lib().ocs_or_rk4inf_compute_adjoints(self._h, prob._h, _p(u), _p(lam), _p(dJdu))
        return lam, dJdu


class _Control:
    kind = None

    def __init__(self, t, nBasis, nControls):
        self.t = _f(t).ravel()
        self.nBasis, self.nControls = int(nBasis), int(nControls)
        self._h = lib().ocs_or_control_create(self.kind, _p(self.t), self.t.size, self.nBasis,
                                              self.nControls)
        self.B = np.ctypeslib.as_array(lib().ocs_or_control_B(self._h),
                                       (self.nBasis * self.t.size,)).copy().reshape(
            (self.nBasis, self.t.size), order="F")
        self._pts = np.ctypeslib.as_array(lib().ocs_or_control_pts(self._h), (self.nBasis,)).copy()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ocs_or_control_destroy(self._h)
            self._h = None

    def compute_u(self, v):
        v = _f(v).ravel()
        u = np.empty((self.nControls, self.t.size), order="F")
        lib().ocs_or_control_compute_u(self._h, _p(v), _p(u))
        return u

    def compute_dJdv(self, dJdu):
        dJdu = _f(dJdu, (self.nControls, self.t.size))
        out = np.empty(self.nControls * self.nBasis)
        lib().ocs_or_control_compute_dJdv(self._h, _p(dJdu), _p(out))
        return out

    def compute_initial_v(self, u0):
        u0 = _f(np.atleast_1d(u0)).ravel()
        v = np.empty(self.nControls * self.nBasis)
        rc = lib().ocs_or_control_compute_initial_v(self._h, _p(u0), u0.size, _p(v))
        if rc != 0:
            raise ValueError("compute_initial_v: unsupported length of u0")
        return v

    def compute_nlp_bounds(self, bounds):
        b = _f(bounds, (self.nControls, 2))
        Lb, Ub = np.empty(self.nControls * self.nBasis), np.empty(self.nControls * self.nBasis)
        lib().ocs_or_control_compute_nlp_bounds(self._h, _p(b), _p(Lb), _p(Ub))
        return Lb, Ub

    def compute_uFunc(self, v):
        v = _f(v).ravel().copy()

        def uFunc(tq):
            tq = _f(np.atleast_1d(tq)).ravel()
            out = np.empty((self.nControls, tq.size), order="F")
            lib().ocs_or_control_eval_uFunc(self._h, _p(v), tq.size, _p(tq), _p(out))
            return out

        return uFunc


class PWLinearControl(_Control):
    kind = CONTROL_PWLINEAR

    @property
    def controlPts(self):
        return self._pts


class PWConstantControl(_Control):
    kind = CONTROL_PWCONSTANT

    @property
    def intervalStarts(self):
        return self._pts


class ChebyshevControl(_Control):
    kind = CONTROL_CHEBYSHEV


def nlp_objective(integrator, prob, control, x0, v, FreeInitStates=()):
    """functions/single_shooting.m:137-150; returns (J, dJdv, x0_after)."""
    kind = 1 if isinstance(integrator, RK4InfiniteIntegrator) else 0
    x0 = _f(x0).ravel().copy()
    v = _f(v).ravel()
    nFree = len(FreeInitStates)
    fis = (C.c_int * max(nFree, 1))(*FreeInitStates)
    J = C.c_double()
    dJdv = np.empty(control.nControls * control.nBasis + nFree)
    lib().ocs_or_nlp_objective(kind, integrator._h, prob._h, control._h, _p(x0), _p(v), nFree, fis,
                               C.byref(J), _p(dJdv))
    return J.value, dJdv, x0


def compute_x_lam(integrator, prob, x0, ugrid, want_J=False):
    """compute_x_lam.m / compute_x_lam_J.m on the grid (odevr7 -> RK4).  Returns node samples."""
    N = integrator.nSTEPS
    x0, ugrid = _f(x0).ravel(), _f(ugrid, (prob.nC, 2 * N + 1))
    x = np.empty((prob.nS, N + 1), order="F")
    lam = np.empty((prob.nS, N + 1), order="F")
    J = C.c_double()
    lib().ocs_or_compute_x_lam(integrator._h, prob._h, _p(x0), _p(ugrid), _p(x), _p(lam), C.byref(J))
    return (x, lam, J.value) if want_J else (x, lam)


def control_from_x_lam(integrator, prob, x, lam, tq):
    tq = _f(np.atleast_1d(tq)).ravel()
    N = integrator.nSTEPS
    x, lam = _f(x, (prob.nS, N + 1)), _f(lam, (prob.nS, N + 1))
    out = np.empty((prob.nC, tq.size), order="F")
    lib().ocs_or_control_from_x_lam(integrator._h, prob._h, _p(x), _p(lam), tq.size, _p(tq), _p(out))
    return out


def fb_sweep(prob, x0, tspan, options=None):
    """functions/fb_sweep.m on the grid.  Returns a dict; empty when the sweep did not converge
    (fb_sweep.m:77), plus bookkeeping under '_sweeps' / '_maxChange'."""
    options = dict(options or {})
    integ = RK4Integrator(tspan)
    N = integ.nSTEPS
    o = FbsOptions()
    lib().ocs_or_fbs_default_options(C.byref(o))
    for k in ("uRelTol", "uAbsTol", "nSWEEPS", "nERROR_PTS", "nINTERP_PTS", "uRelax"):
        if k in options:
            setattr(o, k, options[k])
    T0, TF = integ.t[0], integ.t[-1]
    errorPts = linspace(T0, TF, o.nERROR_PTS)
    interpPts = linspace(T0, TF, o.nINTERP_PTS)
    lb = prob.ControlBounds[:, 0:1]
    if "u0" in options:
        u0 = options["u0"]
        if callable(u0):
            u0g, u0e = _f(u0(integ.t), (prob.nC, -1)), _f(u0(errorPts), (prob.nC, -1))
        else:  # numeric: evenly spaced samples -> pchip (fb_sweep.m:61-66)
            u0 = _f(np.atleast_2d(u0))
            time = linspace(T0, TF, u0.shape[1])
            u0g = vector_interp(time, u0, INTERP_PCHIP, integ.t)
            u0e = vector_interp(time, u0, INTERP_PCHIP, errorPts)
    else:  # fb_sweep.m:23 lower bound
        u0g = _f(lb @ np.ones((1, 2 * N + 1)))
        u0e = _f(lb @ np.ones((1, o.nERROR_PTS)))
    x0 = _f(x0).ravel()
    x = np.empty((prob.nS, N + 1), order="F")
    lam = np.empty((prob.nS, N + 1), order="F")
    uI = np.empty((prob.nC, o.nINTERP_PTS), order="F")
    J = C.c_double()
    mc = np.full(o.nSWEEPS, np.nan)
    k = lib().ocs_or_fb_sweep(integ._h, prob._h, _p(x0), C.byref(o), _p(_f(u0g)), _p(_f(u0e)), _p(x),
                              _p(lam), _p(uI), C.byref(J), _p(mc))
    soln = {"_sweeps": k, "_maxChange": mc, "_tspan": integ.tspan, "_interpPts": interpPts}
    if k > 0:
        soln.update(x=x, lam=lam, u=uI, J=J.value)
    return soln


def max_threads():
    return lib().ocs_or_max_threads()


def batch_states_adjoints(prob, tspan, x0, u, nthreads=1, want=("x", "J", "lam", "dJdu"), out=None):
    """OpenMP-over-batch driver for the cpu_baseline timing.  x0: nS x batch, u: nC x (2N+1) x batch.
    Pass the dict returned by an earlier call as `out` to reuse (already touched) output arrays."""
    tspan = _f(tspan).ravel()
    N = tspan.size - 1
    x0 = _f(np.atleast_2d(x0))
    batch = x0.shape[1]
    u = _f(u, (prob.nC, 2 * N + 1, batch))
    if out is None:
        out = {}
        out["x"] = np.empty((prob.nAug, N + 1, batch), order="F") if "x" in want else None
        out["J"] = np.empty(batch) if "J" in want else None
        out["lam"] = np.empty((prob.nAug, N + 1, batch), order="F") if "lam" in want else None
        out["dJdu"] = np.empty((prob.nC, 2 * N + 1, batch), order="F") if "dJdu" in want else None
    lib().ocs_or_batch_states_adjoints(prob.id, prob.nS, prob.nC, _p(prob.params), prob.params.size,
                                       _p(prob.ControlBounds), _p(tspan), tspan.size, batch, _p(x0),
                                       _p(u), _p(out["x"]), _p(out["J"]), _p(out["lam"]),
                                       _p(out["dJdu"]), nthreads)
    return out


_fast = None


def fast_logistic_pair(m, c, r, tspan, x0, u, nthreads=1, out=None):
    """The TUNED CPU implementation of the bench workload (ocs_cpu_fast.c: batch-minor arrays, vector loops over blocks of
    64 trajectories, stage states recomputed, FMA on) -- bench.py's cpu_baseline beside the literal restatement.
    x0 [nS][B], u [2N+1][B] (batch-minor) -> dict x, lam [N+1][nS+1][B], dJdu [2N+1][B], J [B]."""
    global _fast
    if _fast is None:
        build()
        _fast = C.CDLL(os.path.join(_HERE, "_build", "libcpufast.so"))
        _fast.ocs_fast_logistic_pair.argtypes = [C.c_int, C.c_int, C.c_long, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp,
                                                 _dp, _dp, C.c_int]
        _fast.ocs_fast_logistic_pair.restype = None
    m = np.ascontiguousarray(m, dtype=np.float64).ravel()
    tspan = np.ascontiguousarray(tspan, dtype=np.float64).ravel()
    x0, u = np.ascontiguousarray(x0, dtype=np.float64), np.ascontiguousarray(u, dtype=np.float64)
    nS, B, N = m.size, x0.shape[-1], tspan.size - 1
    assert nS <= 8 and x0.shape == (nS, B) and u.reshape(2 * N + 1, B).shape == (2 * N + 1, B)
    if out is None:
        out = {"x": np.empty((N + 1, nS + 1, B)), "lam": np.empty((N + 1, nS + 1, B)), "dJdu": np.empty((2 * N + 1, B)),
               "J": np.empty(B)}
    _fast.ocs_fast_logistic_pair(nS, N, B, _p(tspan), _p(m), float(c), float(r), _p(x0), _p(u), _p(out["x"]), _p(out["J"]),
                                 _p(out["lam"]), _p(out["dJdu"]), int(nthreads))
    return out
