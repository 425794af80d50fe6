"""Control parametrisations (reference: Control/Control.m, PWLinearControl.m, PWConstantControl.m,
ChebyshevControl.m): a fixed basis matrix B (nBasis x nT), u = reshape(v, nC, []) * B and
dJdv = reshape(dJdu * B', [], 1).  B is built on the host by the library exactly as the
reference's constructors do; the products run on the GPU with B kept sparse."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import check, lib
from .integrator import _dptr, _stream
from .interp import vectorInterpolant
from .problem import _f, _p

CONTROL_PWLINEAR, CONTROL_PWCONSTANT, CONTROL_CHEBYSHEV = 1, 2, 3


class Control:
    """Control/Control.m:4-14."""
    kind = None

    def __init__(self, t, nBasis, nControls):
        self.t = _f(t).ravel()
        self.nBasis, self.nControls = int(nBasis), int(nControls)
        h = C.c_void_p()
        check(lib.ocs_control_create(C.byref(h), self.kind, _p(self.t), self.t.size, self.nBasis, self.nControls))
        self._h = h
        self.B = np.empty((self.nBasis, self.t.size), order="F")
        check(lib.ocs_control_basis(h, _p(self.B)))
        self._pts = np.empty(self.nBasis)
        check(lib.ocs_control_points(h, _p(self._pts)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:
            lib.ocs_control_destroy(h)
            self._h = None

    def _batch(self, a, rows):
        a = np.asarray(a, dtype=np.float64)
        batched = a.ndim == (2 if rows is None else 3)
        return a, batched

    def compute_u(self, v):
        """u = compute_u(obj, v): v (nC*nBasis) [x batch] -> u nC x nT [x batch]."""
        v = np.asarray(v, dtype=np.float64)
        batched = v.ndim == 2
        batch = v.shape[1] if batched else 1
        v = _f(v, (self.nControls * self.nBasis, batch))
        u = np.empty((self.nControls, self.t.size, batch), order="F")
        check(lib.ocs_control_compute_u(self._h, batch, _p(v), _p(u)))
        return u if batched else u[:, :, 0]

    def compute_dJdv(self, dJdu):
        """dJdv = compute_dJdv(obj, dJdu): dJdu nC x nT [x batch] -> (nC*nBasis) [x batch]."""
        dJdu = np.asarray(dJdu, dtype=np.float64)
        batched = dJdu.ndim == 3
        batch = dJdu.shape[2] if batched else 1
        dJdu = _f(dJdu, (self.nControls, self.t.size, batch))
        out = np.empty((self.nControls * self.nBasis, batch), order="F")
        check(lib.ocs_control_compute_dJdv(self._h, batch, _p(dJdu), _p(out)))
        return out if batched else out[:, 0]

    def compute_u_dev(self, v, u=None):
        """device, batch-minor: v [nBasis][nC][B] -> u [nT][nC][B]."""
        B = v.shape[-1]
        if u is None:
            u = torch.empty((self.t.size, self.nControls, B), dtype=torch.float64, device=v.device)
        check(lib.ocs_control_compute_u_dev(self._h, B, _dptr(v), _dptr(u), _stream()))
        return u

    def compute_dJdv_dev(self, dJdu, dJdv=None):
        B = dJdu.shape[-1]
        if dJdv is None:
            dJdv = torch.empty((self.nBasis, self.nControls, B), dtype=torch.float64, device=dJdu.device)
        check(lib.ocs_control_compute_dJdv_dev(self._h, B, _dptr(dJdu), _dptr(dJdv), _stream()))
        return dJdv

    def compute_initial_v(self, u0):
        u0 = _f(np.atleast_1d(u0)).ravel()
        v = np.empty(self.nControls * self.nBasis)
        check(lib.ocs_control_compute_initial_v(self._h, _p(u0), u0.size, _p(v)))
        return v

    def compute_uFunc(self, v):
        """uFunc = compute_uFunc(obj, v): a callable t (1 x k) -> nC x k."""
        v = _f(v).ravel().copy()

        def uFunc(tq):
            tq = _f(np.atleast_1d(tq)).ravel()
            out = np.empty((self.nControls, tq.size), order="F")
            check(lib.ocs_control_eval_uFunc(self._h, _p(v), tq.size, _p(tq), _p(out)))
            return out

        return uFunc


def _set_fusion(self, mode):
    """Where the basis is dense with <= 32 functions, nlp_objective can apply it inside the RK4 kernels:
    "auto" (large batches), "off", "on" (whenever supported), "lane" (as "on", but on the lane-per-trajectory kernels
    even where the wave-specialised ones with the basis products on the matrix cores apply)."""
    check(lib.ocs_control_set_fusion(self._h, {"auto": 0, "off": 1, "on": 2, "lane": 3}[mode]))
    return self


Control.set_fusion = _set_fusion


class _BoundedControl(Control):
    def compute_nlp_bounds(self, controlBounds):
        """[Lb, Ub] = compute_nlp_bounds(obj, controlBounds)."""
        b = _f(controlBounds, (self.nControls, 2))
        Lb, Ub = np.empty(self.nControls * self.nBasis), np.empty(self.nControls * self.nBasis)
        check(lib.ocs_control_compute_nlp_bounds(self._h, _p(b), _p(Lb), _p(Ub)))
        return Lb, Ub


class PWLinearControl(_BoundedControl):
    """Control/PWLinearControl.m: obj = PWLinearControl(t, nControlPts, nControls)."""
    kind = CONTROL_PWLINEAR

    def __init__(self, t, nControlPts, nControls):
        super().__init__(t, nControlPts, nControls)
        self.nControlPts = int(nControlPts)

    @property
    def controlPts(self):
        return self._pts


class PWConstantControl(_BoundedControl):
    """Control/PWConstantControl.m: obj = PWConstantControl(t, nControlIntervals, nControls)."""
    kind = CONTROL_PWCONSTANT

    def __init__(self, t, nControlIntervals, nControls):
        super().__init__(t, nControlIntervals, nControls)
        self.nControlIntervals = int(nControlIntervals)

    @property
    def intervalStarts(self):
        return self._pts


class ChebyshevControl(Control):
    """Control/ChebyshevControl.m: obj = ChebyshevControl(t, nControlBasis, nControls).
    The reference class has no compute_nlp_bounds, and its compute_lincon is declared with an EMPTY body (:51-53:
    single_shooting.m:106-107 would fail on it): u = v*B itself is unclamped (nlp_objective evaluates it as such).
    `compute_lincon` below fills that hook in the way its signature asks for -- the control bounds as linear
    inequalities on the coefficients at grid points -- so that single_shooting can honour ControlBounds with this basis."""
    kind = CONTROL_CHEBYSHEV

    def __init__(self, t, nControlBasis, nControls, lincon_points=None):
        super().__init__(t, nControlBasis, nControls)
        self.nControlBasis = int(nControlBasis)
        self.lincon_points = lincon_points   # None: every grid point; an int n: n points spread evenly over the grid

    def compute_lincon(self, ControlBounds):
        """[A, b] = compute_lincon(obj, ControlBounds)   ChebyshevControl.m:51 (empty in the reference): A v <= b with
        lb_c <= u_c(t_j) = sum_k v(c + nC (k-1)) B(k, j) <= ub_c at the chosen grid points t_j (v: control index fastest,
        compute_u :34-38).  Rows with an infinite bound are left out."""
        cb = _f(ControlBounds, (self.nControls, 2))
        B = self.B                                           # nBasis x nT
        nT, nB, nC = B.shape[1], self.nBasis, self.nControls
        cols = np.arange(nT) if self.lincon_points is None else np.unique(
            np.round(np.linspace(0, nT - 1, int(self.lincon_points))).astype(int))
        rows, rhs = [], []
        for c in range(nC):
            for j in cols:
                a = np.zeros(nB * nC)
                a[c::nC] = B[:, j]
                if np.isfinite(cb[c, 1]):
                    rows.append(a)
                    rhs.append(cb[c, 1])
                if np.isfinite(cb[c, 0]):
                    rows.append(-a)
                    rhs.append(-cb[c, 0])
        if not rows:
            return np.zeros((0, nB * nC)), np.zeros(0)
        return np.vstack(rows), np.asarray(rhs, dtype=np.float64)

    @property
    def controlPts(self):
        return self._pts
