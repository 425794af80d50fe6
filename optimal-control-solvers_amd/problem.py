"""OCProblem plugin (reference: OCProblem/OCProblem.m, tests/TestOCProblem.m).

A kernel cannot call back into a user method, so a problem is a registry id plus a
parameter block (include/ocs.h); F / dFdx_times_vec / dFdu_times_vec keep their names,
argument order and shapes and are evaluated by the device functor.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, dp, lib

PROBLEM_TEST, PROBLEM_LOGISTIC, PROBLEM_LQ = 1, 2, 3


def _f(a, shape=None):
    a = np.asarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape, order="F")
    return np.asfortranarray(a)


def _p(a):
    return None if a is None else a.ctypes.data_as(dp)


class OCProblem:
    """OCProblem/OCProblem.m:1-24 -- abstract interface; concrete problems come from the registry."""

    def __init__(self, problem_id, nS, nC, params, ControlBounds):
        self.nS, self.nC, self.nAug = int(nS), int(nC), int(nS) + 1
        self.params = _f(params).ravel()
        self.ControlBounds = _f(ControlBounds, (self.nC, 2))
        h = C.c_void_p()
        check(lib.ocs_problem_create(C.byref(h), problem_id, self.nS, self.nC, _p(self.params),
                                     self.params.size, _p(self.ControlBounds)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:
            lib.ocs_problem_destroy(h)
            self._h = None

    def set_batch_params(self, index, values):
        """Per-trajectory overrides (batch extension): values is len(index) x batch."""
        index = np.asarray(index, dtype=np.int32).ravel()
        if index.size == 0:
            check(lib.ocs_problem_set_batch_params(self._h, 0, None, 0, None))
            return
        values = _f(np.atleast_2d(values))
        if values.shape[0] != index.size:
            raise ValueError("values must be len(index) x batch")
        check(lib.ocs_problem_set_batch_params(self._h, values.shape[1],
                                               index.ctypes.data_as(_lib.ip), index.size, _p(values)))

    def _cols(self, t):
        t = _f(np.atleast_1d(t)).ravel()
        return t, t.size

    def F(self, t, y, u):
        """OCProblem.m:12 / TestOCProblem.m:22-26"""
        t, k = self._cols(t)
        y, u = _f(y, (self.nAug, k)), _f(u, (self.nC, k))
        out = np.empty((self.nAug, k), order="F")
        check(lib.ocs_problem_F(self._h, k, _p(t), _p(y), _p(u), _p(out)))
        return out

    def dFdx_times_vec(self, t, y, u, v):
        """OCProblem.m:16 / TestOCProblem.m:29-33"""
        t, k = self._cols(t)
        y, u, v = _f(y, (self.nAug, k)), _f(u, (self.nC, k)), _f(v, (self.nAug, k))
        out = np.empty((self.nAug, k), order="F")
        check(lib.ocs_problem_dFdx_times_vec(self._h, k, _p(t), _p(y), _p(u), _p(v), _p(out)))
        return out

    def dFdu_times_vec(self, t, y, u, v):
        """OCProblem.m:19 / TestOCProblem.m:36-38"""
        t, k = self._cols(t)
        y, u, v = _f(y, (self.nAug, k)), _f(u, (self.nC, k)), _f(v, (self.nAug, k))
        out = np.empty((self.nC, k), order="F")
        check(lib.ocs_problem_dFdu_times_vec(self._h, k, _p(t), _p(y), _p(u), _p(v), _p(out)))
        return out

    def ControlChar(self, t, x, lam):
        """The Gen-1 method of make_from_symbolic.m:33-38 (clamped to ControlBounds, :111) that fb_sweep.m:96,123 evaluates;
        x, lam: nS x k."""
        t, k = self._cols(t)
        x, lam = _f(x, (self.nS, k)), _f(lam, (self.nS, k))
        out = np.empty((self.nC, k), order="F")
        check(lib.ocs_problem_ControlChar(self._h, k, _p(t), _p(x), _p(lam), _p(out)))
        return out


class UserProblem(OCProblem):
    """An OCProblem subclass written by the user: the three plugin methods F, dFdx_times_vec and
    dFdu_times_vec (OCProblem/OCProblem.m:8-21) are given as device C++ source (functions ocs_F,
    ocs_dFdx_times_vec, ocs_dFdu_times_vec, optionally ocs_ControlChar; contract in
    csrc/ocs_user_functor.hpp) and compiled with hipRTC for gfx950 when the object is created."""

    def __init__(self, source, nS, nC, params, ControlBounds, has_control_char=False, row_separable=False,
                 control_from_costate=False):
        """row_separable: the source defines ROW functions (ocs_row_F, ocs_row_q, ocs_row_dFdy, ocs_row_dFdu; contract in
        csrc/ocs_user_functor.hpp) instead of the three full-vector methods, which are derived from them; the problem
        then also runs on the wave-specialised state pass and the scan adjoint pass (nC = 1, nS in {1, 2, 4}).
        control_from_costate (with row_separable and has_control_char): the problem declares that ocs_ControlChar does
        not read x and ocs_row_dFdy does not read u; fb_sweep then runs its two-kernel sweep (state pass with the control
        update folded in, costate pass as a scan with the convergence test) as for the registry problems."""
        self.nS, self.nC, self.nAug = int(nS), int(nC), int(nS) + 1
        self.params = _f(params).ravel()
        self.ControlBounds = _f(ControlBounds, (self.nC, 2))
        self.source = source
        h = C.c_void_p()
        check(lib.ocs_problem_create_from_source(C.byref(h), source.encode(), self.nS, self.nC, _p(self.params),
                                                 self.params.size, _p(self.ControlBounds),
                                                 int(bool(has_control_char)) | (2 if row_separable else 0) |
                                                 (4 if control_from_costate else 0)))
        self._h = h

    @staticmethod
    def check_source(source, nS, nC, nparams, has_control_char=False, row_separable=False, control_from_costate=False):
        """Compile only (works without a GPU); raises OcsError with the compiler log on failure."""
        check(lib.ocs_problem_check_source(source.encode(), int(nS), int(nC), int(nparams),
                                           int(bool(has_control_char)) | (2 if row_separable else 0) |
                                           (4 if control_from_costate else 0)))


class TestOCProblem(OCProblem):
    """tests/TestOCProblem.m:16-20: prob = TestOCProblem(p, ControlBounds), p has fields c, m, r."""
    __test__ = False

    def __init__(self, p, ControlBounds):
        self.c, self.m, self.r = float(p["c"]), float(p["m"]), float(p["r"])
        super().__init__(PROBLEM_TEST, 1, 1, [self.c, self.m, self.r], ControlBounds)


class LogisticProblem(OCProblem):
    """LogisticK (build-defined generalisation): nS uncoupled logistic states
    sharing one harvest control; nS = 1 is TestOCProblem."""

    def __init__(self, m, c, r, ControlBounds):
        m = np.atleast_1d(np.asarray(m, dtype=np.float64))
        self.c, self.m, self.r = float(c), m, float(r)
        super().__init__(PROBLEM_LOGISTIC, m.size, 1, np.concatenate([[c, r], m]), ControlBounds)


class LQProblem(OCProblem):
    """Build-defined linear-quadratic problem: F = [A x + Bu u ; e^{-rt}(x'diag(q)x + u'diag(rdiag)u)].
    The Jacobian A is shared by the whole batch, so the integrator passes run on the matrix cores
    (csrc/ocs_lq_kernels.hip).  nS <= 32, nC <= 4."""

    def __init__(self, A, Bu, q, rdiag, r, ControlBounds):
        A = np.asarray(A, dtype=np.float64)
        nS = A.shape[0]
        Bu = np.asarray(Bu, dtype=np.float64).reshape(nS, -1)
        nC = Bu.shape[1]
        self.A, self.Bu, self.r = A, Bu, float(r)
        self.q = np.asarray(q, dtype=np.float64).reshape(nS)
        self.rdiag = np.asarray(rdiag, dtype=np.float64).reshape(nC)
        par = np.concatenate([[float(r)], A.ravel(order="F"), Bu.ravel(order="F"), self.q, self.rdiag])
        super().__init__(PROBLEM_LQ, nS, nC, par, ControlBounds)
