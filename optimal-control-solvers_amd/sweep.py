"""Forward-backward sweep (reference: functions/fb_sweep.m, compute_x_lam.m, compute_x_lam_J.m).

The Gen-1 drivers run on a Gen-2 OCProblem through the adapter of SURVEY A9; odevr7 is replaced
by RK4 on the grid RK4Integrator(tspan).  `fb_sweep` keeps the reference's signature
and output (a struct of callables, empty when the sweep did not converge); `fb_sweep_batch` is the
batch entry point returning sample arrays."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import FbsOptions, check, ip, lib
from .integrator import RK4Integrator, _dptr, _stream
from .interp import vectorInterpolant
from .problem import _f, _p


def matlab_linspace(a, b, n):
    k = np.arange(n, dtype=np.float64)
    y = a + (k * (b - a)) / (n - 1)
    y[0], y[-1] = a, b
    return y


def _options(options):
    o = FbsOptions()
    check(lib.ocs_fbs_default_options(C.byref(o)))
    options = dict(options or {})
    for k in ("uRelTol", "uAbsTol", "nSWEEPS", "nERROR_PTS", "nINTERP_PTS", "fused_update_off", "nWINDOWS", "cost_row",
              "uRelax"):
        if k in options:
            setattr(o, k, options[k])
    # RelTol / AbsTol (fb_sweep.m:18-19) steer odevr7's adaptive step control.  Here the passes are classical RK4 on
    # the caller's tspan grid: accuracy is set by the grid (4th order, tests/test_oracle_kat.py), not by these
    # tolerances -- say so instead of ignoring them silently.
    ignored = [k for k in ("RelTol", "AbsTol") if k in options]
    if ignored:
        import warnings
        warnings.warn(f"fb_sweep: {', '.join(ignored)} steer the reference's adaptive odevr7 integrator and have no effect "
                      "here: the state/costate passes are fixed-step RK4 on tspan (refine tspan for accuracy)",
                      RuntimeWarning, stacklevel=3)
    return o, options


def compute_x_lam(prob, x0, tspan, ugrid, integrator=None):
    """[x, lam] = compute_x_lam(prob, x0, tspan, u, ...) with u sampled on the 2N+1 grid.
    Returns node samples x, lam: nS x (N+1) [x batch]."""
    x, lam, _ = compute_x_lam_J(prob, x0, tspan, ugrid, integrator)
    return x, lam


def compute_x_lam_J(prob, x0, tspan, ugrid, integrator=None):
    integ = integrator or RK4Integrator(tspan)
    N = integ.nSTEPS
    ugrid = np.asarray(ugrid, dtype=np.float64)
    batched = ugrid.ndim == 3
    batch = ugrid.shape[2] if batched else 1
    ugrid = _f(ugrid, (prob.nC, 2 * N + 1, batch))
    x0 = _f(x0, (prob.nS, batch))
    x = np.empty((prob.nS, N + 1, batch), order="F")
    lam = np.empty((prob.nS, N + 1, batch), order="F")
    J = np.empty(batch)
    check(lib.ocs_compute_x_lam(integ._h, prob._h, batch, _p(x0), _p(ugrid), _p(x), _p(lam), _p(J)))
    if batched:
        return x, lam, J
    return x[:, :, 0], lam[:, :, 0], float(J[0])


def compute_J(prob, x0, tspan, ugrid, integrator=None):
    """J = compute_J(prob, x0, tspan, u, RelTol, AbsTol)   functions/compute_J.m:1-16: the objective of the state pass alone
    (the augmented state [x; int objective], :6-15) with u sampled on the 2N+1 grid -- the J of RK4Integrator.compute_states."""
    integ = integrator or RK4Integrator(tspan)
    _, J = integ.compute_states(prob, x0, ugrid)
    return J


def _sample_u0(u0, prob, pts_list, batch):
    """u0: callable t -> nC x k, or numeric nC x m (evenly spaced samples -> pchip, fb_sweep.m:61-66)."""
    outs = []
    for pts in pts_list:
        if callable(u0):
            s = np.asarray(u0(pts), dtype=np.float64).reshape(prob.nC, pts.size)
        else:
            a = np.atleast_2d(np.asarray(u0, dtype=np.float64))
            time = matlab_linspace(pts_list[0][0], pts_list[0][-1], a.shape[1])
            s = vectorInterpolant(time, a, "pchip")(pts)
        outs.append(np.asfortranarray(np.repeat(s[:, :, None], batch, axis=2)))
    return outs


def fb_sweep_batch(prob, x0, tspan, options=None, integrator=None):
    """Batch fb_sweep: x0 is nS x batch (instances may also differ through prob.set_batch_params).
    Returns a dict of sample arrays: x, lam (nS x (N+1) x batch on tspan), u (nC x nINTERP x batch on
    interpPts), J, sweeps (0 = not converged), maxChange (nSWEEPS x batch), tspan, interpPts."""
    o, options = _options(options)
    integ = integrator or RK4Integrator(tspan)
    N = integ.nSTEPS
    x0 = _f(np.atleast_2d(np.asarray(x0, dtype=np.float64)))
    if x0.shape[0] != prob.nS:
        x0 = _f(x0.reshape(prob.nS, -1))
    batch = x0.shape[1]
    T0, TF = integ.t[0], integ.t[-1]
    errorPts = matlab_linspace(T0, TF, o.nERROR_PTS)     # fb_sweep.m:69
    interpPts = matlab_linspace(T0, TF, o.nINTERP_PTS)   # :70
    u0g = u0e = None
    if "u0" in options:
        u0g, u0e = _sample_u0(options["u0"], prob, [integ.t, errorPts], batch)
    x = np.empty((prob.nS, N + 1, batch), order="F")
    lam = np.empty((prob.nS, N + 1, batch), order="F")
    uI = np.empty((prob.nC, o.nINTERP_PTS, batch), order="F")
    J = np.empty(batch)
    sweeps = np.zeros(batch, dtype=np.int32)
    mc = np.empty((o.nSWEEPS, batch), order="F")
    status = check(lib.ocs_fb_sweep(integ._h, prob._h, batch, _p(x0), C.byref(o), _p(u0g), _p(u0e), _p(x), _p(lam),
                                    _p(uI), _p(J), sweeps.ctypes.data_as(ip), _p(mc)))
    return {"x": x, "lam": lam, "u": uI, "J": J, "sweeps": sweeps, "maxChange": mc, "status": status,
            "tspan": integ.tspan, "interpPts": interpPts}


def fb_sweep_path(integrator):
    """Diagnostic (ocs.h ocs_fb_sweep_path): the sweep loop the last fb_sweep on this integrator ran -- 1 kernel by kernel
    as the reference sequences it, 2 fused control update, 3 windows, 4 the two-kernel sweep (fold)."""
    return int(lib.ocs_fb_sweep_path(integrator._h))


def fb_sweep(prob, x0, tspan, options=None):
    """soln = fb_sweep(prob, x0, tspan, options)   fb_sweep.m:1.  Returns a dict with the callables
    x, lam, u (pchip interpolants, vectorInterpolant.m) and J, or an empty dict when the sweep did not
    converge within nSWEEPS (fb_sweep.m:77; check `'u' in soln`, manual p.5)."""
    r = fb_sweep_batch(prob, np.asarray(x0, dtype=np.float64).reshape(prob.nS, 1), tspan, options)
    if r["sweeps"][0] == 0:
        return {}
    return {"x": vectorInterpolant(r["tspan"], r["x"][:, :, 0], "pchip"),
            "lam": vectorInterpolant(r["tspan"], r["lam"][:, :, 0], "pchip"),
            "u": vectorInterpolant(r["interpPts"], r["u"][:, :, 0], "pchip"),     # :123
            "J": float(r["J"][0])}


def fb_sweep_dev(prob, integ, x0, options=None, u0grid=None, u0err=None, out=None):
    """Device path: x0 [nS][B] torch tensor; returns device tensors (batch-minor).  `out`: the dict an earlier call
    with the same shapes returned -- its tensors are written again instead of allocating new ones (a loop of solves
    then performs no allocation at all)."""
    o, _ = _options(options)
    N, B = integ.nSTEPS, x0.shape[-1]
    dev = x0.device
    if out is not None:
        xaug, lam, uI, J, sweeps, mc = (out[k] for k in ("xaug", "lam", "u", "J", "sweeps", "maxChange"))
        if (tuple(xaug.shape) != (N + 1, prob.nAug, B) or tuple(lam.shape) != (N + 1, prob.nS, B)
                or tuple(uI.shape) != (o.nINTERP_PTS, prob.nC, B) or tuple(mc.shape) != (o.nSWEEPS, B)
                or J.numel() != B or sweeps.numel() != B or sweeps.dtype != torch.int32):
            raise ValueError("fb_sweep_dev: `out` does not have the shapes of this call")
    else:
        xaug = torch.empty((N + 1, prob.nAug, B), dtype=torch.float64, device=dev)
        lam = torch.empty((N + 1, prob.nS, B), dtype=torch.float64, device=dev)
        uI = torch.empty((o.nINTERP_PTS, prob.nC, B), dtype=torch.float64, device=dev)
        J = torch.empty(B, dtype=torch.float64, device=dev)
        sweeps = torch.zeros(B, dtype=torch.int32, device=dev)
        mc = torch.empty((o.nSWEEPS, B), dtype=torch.float64, device=dev)
    status = check(lib.ocs_fb_sweep_dev(integ._h, prob._h, B, _dptr(x0), C.byref(o), _dptr(u0grid), _dptr(u0err),
                                        _dptr(xaug), _dptr(lam), _dptr(uI), _dptr(J), C.c_void_p(sweeps.data_ptr()),
                                        _dptr(mc), _stream()))
    return {"xaug": xaug, "lam": lam, "u": uI, "J": J, "sweeps": sweeps, "maxChange": mc, "status": status}
