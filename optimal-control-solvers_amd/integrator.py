"""Integrator plugin (reference: Integrator/Integrator.m, Integrator/RK4Integrator.m).

Host calls take numpy arrays in MATLAB shapes with the batch as an optional trailing
dimension; `*_dev` calls take torch CUDA tensors in the device-native batch-minor layout
and run asynchronously on torch's current stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import check, dp, lib
from .problem import _f, _p


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dptr(t):
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


class Integrator:
    """Integrator/Integrator.m:6-15: property t, compute_states, compute_adjoints."""
    t = None

    def compute_states(self, prob, x0, u):
        raise NotImplementedError

    def compute_adjoints(self, prob, u, lamT=None):
        raise NotImplementedError


class RK4Integrator(Integrator):
    """Integrator/RK4Integrator.m:16-25: obj = RK4Integrator(tspan)."""

    def __init__(self, tspan):
        self.tspan = _f(tspan).ravel()
        h = C.c_void_p()
        check(lib.ocs_rk4_create(C.byref(h), _p(self.tspan), self.tspan.size))
        self._finish_init(h)

    def _finish_init(self, h):
        self._h = h
        n = C.c_int()
        check(lib.ocs_integrator_nsteps(h, C.byref(n)))
        self.nSTEPS = n.value
        self.t = np.empty(2 * self.nSTEPS + 1)
        self.h = np.empty(self.nSTEPS)
        check(lib.ocs_integrator_t(h, _p(self.t)))
        check(lib.ocs_integrator_h(h, _p(self.h)))
        self._batch_shape = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:
            lib.ocs_integrator_destroy(h)
            self._h = None

    def set_mapping(self, mapping):
        """Thread mapping of the RK4 kernels: "auto", "lane" (lane per trajectory) or "rowsplit"."""
        code = mapping if isinstance(mapping, int) else {"auto": 0, "lane": 1, "rowsplit": 2, "pipeline": 3, "scan": 4}[mapping]
        check(lib.ocs_integrator_set_mapping(self._h, code))
        return self

    # ---- host path (MATLAB shapes) -------------------------------------------------
    def compute_states(self, prob, x0, u):
        """[x, J] = compute_states(obj, prob, x0, u)   RK4Integrator.m:28-56.
        x0: nS (x batch); u: nC x (2N+1) (x batch).  Returns x nAug x (N+1) (x batch), J."""
        N = self.nSTEPS
        u = np.asarray(u, dtype=np.float64)
        batched = u.ndim == 3
        batch = u.shape[2] if batched else 1
        u = _f(u, (prob.nC, 2 * N + 1, batch))
        x0 = _f(x0, (prob.nS, batch))
        x = np.empty((prob.nAug, N + 1, batch), order="F")
        J = np.empty(batch)
        self.status = check(lib.ocs_compute_states(self._h, prob._h, batch, _p(x0), _p(u), _p(x), _p(J)))
        return (x, J) if batched else (x[:, :, 0], float(J[0]))

    def compute_adjoints(self, prob, u, lamT=None, nargout=2):
        """[lam, dJdu] = compute_adjoints(obj, prob, u, lamT)   RK4Integrator.m:59-121."""
        N = self.nSTEPS
        u = np.asarray(u, dtype=np.float64)
        batched = u.ndim == 3
        batch = u.shape[2] if batched else 1
        u = _f(u, (prob.nC, 2 * N + 1, batch))
        lt = None if lamT is None else _f(lamT, (prob.nAug, batch))
        lam = np.empty((prob.nAug, N + 1, batch), order="F")
        dJdu = np.empty((prob.nC, 2 * N + 1, batch), order="F") if nargout > 1 else None
        check(lib.ocs_compute_adjoints(self._h, prob._h, batch, _p(u), _p(lt), _p(lam), _p(dJdu)))
        if not batched:
            lam = lam[:, :, 0]
            dJdu = None if dJdu is None else dJdu[:, :, 0]
        return (lam, dJdu) if nargout > 1 else lam

    # ---- device path (batch-minor torch tensors, async on the current stream) --------
    @staticmethod
    def _chk(name, t, shape, dev):
        """a wrongly sized or misplaced device tensor is an out-of-bounds access inside a kernel, i.e. a GPU fault:
        check shape, dtype, contiguity and device here"""
        if t is None:
            return
        if tuple(t.shape) != tuple(shape) or t.dtype != torch.float64 or not t.is_contiguous() or t.device != dev:
            raise ValueError(f"{name}: expected a contiguous float64 tensor of shape {tuple(shape)} on {dev}, "
                             f"got {tuple(t.shape)} {t.dtype} on {t.device}")

    def compute_states_dev(self, prob, x0, u, x=None, J=None):
        """x0 [nS][B], u [2N+1][nC][B] -> x [N+1][nAug][B] (optional), J [B]."""
        N = self.nSTEPS
        B = x0.shape[-1]
        nS, nC, dev = prob.nS, prob.ControlBounds.shape[0], x0.device
        if J is None:
            J = torch.empty(B, dtype=torch.float64, device=dev)
        self._chk("x0", x0, (nS, B), dev)
        self._chk("u", u, (2 * N + 1, nC, B), dev)
        self._chk("x", x, (N + 1, nS + 1, B), dev)
        self._chk("J", J, (B,), dev)
        check(lib.ocs_compute_states_dev(self._h, prob._h, B, _dptr(x0), _dptr(u), _dptr(x), _dptr(J),
                                         _stream()))
        self._ck_ref = x   # the adjoint pass re-reads it (the xK contract): keep it alive until then
        return x, J

    def compute_adjoints_dev(self, prob, u, lamT=None, lam=None, dJdu=None):
        N = self.nSTEPS
        B = u.shape[-1]
        nS, nC, dev = prob.nS, prob.ControlBounds.shape[0], u.device
        self._chk("u", u, (2 * N + 1, nC, B), dev)
        self._chk("lamT", lamT, (nS + 1, B), dev)
        self._chk("lam", lam, (N + 1, nS + 1, B), dev)
        self._chk("dJdu", dJdu, (2 * N + 1, nC, B), dev)
        check(lib.ocs_compute_adjoints_dev(self._h, prob._h, B, _dptr(u), _dptr(lamT), _dptr(lam),
                                           _dptr(dJdu), _stream()))
        return lam, dJdu

    def trajectory_status(self, batch):
        """per-trajectory flags of the last host compute_states / nlp_objective call on this handle
        (OCS_NUM_NONFINITE = 1 where the objective is NaN/Inf)"""
        st = np.zeros(batch, dtype=np.int32)
        check(lib.ocs_integrator_trajectory_status(self._h, batch, st.ctypes.data_as(C.POINTER(C.c_int))))
        return st


def trajectory_status_dev(J, status=None):
    """device: status [B] int32 from J [B] (asynchronous on the current stream)"""
    if status is None:
        status = torch.empty(J.shape[0], dtype=torch.int32, device=J.device)
    check(lib.ocs_trajectory_status_dev(J.shape[0], _dptr(J), C.c_void_p(status.data_ptr()), _stream()))
    return status


class RK4InfiniteIntegrator(RK4Integrator):
    """Integrator/RK4InfiniteIntegrator.m:12-30: RK4 on tspan, then a tail leg on tspanExtra under
    the constant control uStar; J = J1 + J2, terminal adjoint of leg 1 = lam2(:,1)."""

    def __init__(self, tspan, tspanExtra, uStar):
        self.tspan = _f(tspan).ravel()
        self.tspanExtra = _f(tspanExtra).ravel()
        self.uStar = _f(np.atleast_1d(uStar)).ravel()
        h = C.c_void_p()
        check(lib.ocs_rk4inf_create(C.byref(h), _p(self.tspan), self.tspan.size, _p(self.tspanExtra),
                                    self.tspanExtra.size, _p(self.uStar), self.uStar.size))
        self._finish_init(h)

    def compute_adjoints(self, prob, u, nargout=2):
        return super().compute_adjoints(prob, u, None, nargout)
