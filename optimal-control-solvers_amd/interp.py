"""vectorInterpolant / heval (reference: functions/vectorInterpolant.m, functions/heval.m).

The solvers return sample arrays; this wrapper turns (grid, samples, method) into the callable
t (1 x k) -> n x k that the reference's soln.x / soln.lam / soln.u are.  Evaluation is the
library's host-side ocs_interp (linear / nearest / previous / next / pchip with MATLAB's slope rule)."""
from __future__ import annotations

import numpy as np

from ._lib import check, lib
from .problem import _f, _p

_METHODS = {"linear": 0, "nearest": 1, "previous": 2, "pchip": 3, "next": 4}


def vectorInterpolant(x, v, interpType):
    x = _f(x).ravel().copy()
    v = np.array(np.atleast_2d(np.asarray(v, dtype=np.float64)), order="F", copy=True)
    method = _METHODS[interpType]
    nComp = v.shape[0]

    def fInterp(t):
        t = _f(np.atleast_1d(t)).ravel()
        out = np.empty((nComp, t.size), order="F")
        check(lib.ocs_interp(method, nComp, x.size, _p(x), _p(v), t.size, _p(t), _p(out)))
        return out

    return fInterp


def vectorInterpolant_dev(x, v, interpType):
    """The same for a batch of sample sets resident on the device: v is a torch tensor [n][nComp][batch] (the layout
    the *_dev solvers return); the callable maps t (k points, host) to a device tensor [k][nComp][batch]."""
    import torch

    from .integrator import _dptr, _stream
    x = _f(x).ravel().copy()
    method = _METHODS[interpType]
    if v.dim() == 2:
        v = v[:, None, :]
    v = v.contiguous()
    n, nComp, B = v.shape
    if n != x.size:
        raise ValueError("v must hold one sample per grid point")

    def fInterp(t):
        t = _f(np.atleast_1d(t)).ravel()
        out = torch.empty((t.size, nComp, B), dtype=torch.float64, device=v.device)
        check(lib.ocs_interp_dev(method, nComp, n, _p(x), _dptr(v), t.size, _p(t), _dptr(out), B, _stream()))
        return out

    return fInterp


def heval(func, tspan, components):
    """functions/heval.m:1-6 (components are 0-based here)."""
    return func(tspan)[components, :]


def linspace(d1, d2, n=100):
    """MATLAB linspace(d1, d2, n): d1 + (0:n-1)*(d2-d1)/(n-1) with both end points pinned.  The reference builds
    tspan and fb_sweep's error / interpolation points (fb_sweep.m:69-70) with it; numpy.linspace rounds
    differently (start + k*step), so grids meant to coincide with those points should come from here."""
    n = int(n)
    if n <= 0:
        return np.empty(0)
    if n == 1:
        return np.array([float(d2)])
    k = np.arange(n, dtype=np.float64)
    out = d1 + (k * (d2 - d1)) / (n - 1)
    out[0], out[-1] = d1, d2
    return out
