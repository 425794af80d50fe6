"""Solver drivers above the hot path (reference: functions/single_shooting.m).

`nlp_objective` is the composition v -> (J, dJdv) that fmincon calls (single_shooting.m:137-150);
it is batch-aware and runs entirely on the GPU.  `single_shooting` keeps the reference's call
signature and returns the same `soln` fields; MATLAB's fmincon('sqp') is a toolbox dependency that
does not exist here, so the outer NLP iteration is scipy's SLSQP (also an SQP method) on the host --
iterates differ from fmincon's, the optimum does not (scope)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import OcsError, check, lib
from .control import PWLinearControl
from .integrator import RK4Integrator, _dptr, _stream
from .interp import vectorInterpolant
from .problem import _f, _p


def _fis(FreeInitStates):
    n = len(FreeInitStates)
    arr = (C.c_int * max(n, 1))(*[int(k) for k in FreeInitStates])
    return n, arr


def nlp_objective(integrator, prob, control, x0, v, FreeInitStates=()):
    """[J, dJdv] = nlpObjective(v)   single_shooting.m:137-150.
    x0: nS [x batch]; v: (nC*nBasis + nFree) [x batch]; FreeInitStates are 1-based like MATLAB.
    Returns (J, dJdv, x0_after)."""
    v = np.asarray(v, dtype=np.float64)
    batched = v.ndim == 2
    batch = v.shape[1] if batched else 1
    nFree, fis = _fis(FreeInitStates)
    nV = control.nControls * control.nBasis + nFree
    v = _f(v, (nV, batch))
    x0 = np.array(_f(x0, (prob.nS, batch)), order="F", copy=True)
    J = np.empty(batch)
    dJdv = np.empty((nV, batch), order="F")
    status = check(lib.ocs_nlp_objective(integrator._h, prob._h, control._h, batch, _p(x0), _p(v), nFree, fis,
                                         _p(J), _p(dJdv)))
    integrator.status = status
    if batched:
        return J, dJdv, x0
    return float(J[0]), dJdv[:, 0], x0[:, 0]


def nlp_objective_dev(integrator, prob, control, x0, v, FreeInitStates=(), J=None, dJdv=None):
    """device, batch-minor: x0 [nS][B] (overwritten at FreeInitStates), v [nV][B] -> J [B], dJdv [nV][B]."""
    B = v.shape[-1]
    nFree, fis = _fis(FreeInitStates)
    if J is None:
        J = torch.empty(B, dtype=torch.float64, device=v.device)
    if dJdv is None:
        dJdv = torch.empty_like(v)
    check(lib.ocs_nlp_objective_dev(integrator._h, prob._h, control._h, B, _dptr(x0), _dptr(v), nFree, fis,
                                    _dptr(J), _dptr(dJdv), _stream()))
    return J, dJdv


def single_shooting(prob, x0, tspan, nCONTROL_PTS, **kw):
    """soln = single_shooting(prob, x0, tspan, nCONTROL_PTS, Name, Value, ...)   single_shooting.m:1-130.
    Name/value options and defaults as in :19-30 (TolX 1e-5, TolFun 3e-4, Algorithm 'sqp',
    Reporting, Control, Integrator, u0, FreeInitStates, FreeStateBounds)."""
    from scipy.optimize import minimize

    opt = dict(TolX=1e-5, TolFun=3e-4, Algorithm="sqp", Reporting=False, DerivativeCheck="off", Control=None,
               Integrator=None, u0=0.0, FreeInitStates=(), FreeStateBounds=None, MaxIter=400)
    unknown = set(kw) - set(opt)
    if unknown:
        raise TypeError(f"unknown option(s): {sorted(unknown)}")
    opt.update(kw)
    tspan = _f(tspan).ravel()
    nCONTROLS = prob.ControlBounds.shape[0]
    MinMax = getattr(prob, "MinMax", "Min")                                     # :11-15
    FreeInitStates = [int(k) for k in opt["FreeInitStates"]]
    nFREE = len(FreeInitStates)
    integrator = opt["Integrator"] or RK4Integrator(tspan)                       # :41-45
    control = opt["Control"] or PWLinearControl(integrator.t, nCONTROL_PTS, nCONTROLS)  # :48-52
    x0 = np.array(_f(x0).ravel(), copy=True)
    u0 = np.minimum(prob.ControlBounds[:, 1], np.maximum(prob.ControlBounds[:, 0],
                                                         np.broadcast_to(np.asarray(opt["u0"], dtype=np.float64).ravel(),
                                                                         (nCONTROLS,)) if np.size(opt["u0"]) in (1, nCONTROLS)
                                                         else np.asarray(opt["u0"], dtype=np.float64).ravel()))  # :56
    v0 = control.compute_initial_v(u0)                                           # :82-86
    if nFREE:
        v0 = np.concatenate([v0, x0[np.array(FreeInitStates) - 1]])
    bounds = None
    if hasattr(control, "compute_nlp_bounds"):                                   # :88-97
        Lb, Ub = control.compute_nlp_bounds(prob.ControlBounds)
        if nFREE:
            fsb = _f(opt["FreeStateBounds"], (nFREE, 2))
            Lb, Ub = np.concatenate([Lb, fsb[:, 0]]), np.concatenate([Ub, fsb[:, 1]])
        bounds = list(zip(Lb, Ub))

    nV0 = v0.size - nFREE
    cons = []
    if hasattr(control, "compute_nonlcon"):                                      # :99-104  [c, ceq(, gradc, gradceq)] = nonlcon(v)
        def _nl(v, k):
            out = control.compute_nonlcon(v[:nV0])
            val = np.atleast_1d(np.asarray(out[k], dtype=np.float64)).ravel()
            return -val if k == 0 else val                                       # fmincon: c(v) <= 0; SLSQP: fun(v) >= 0

        def _nlj(v, k):
            out = control.compute_nonlcon(v[:nV0])
            if len(out) < 4:
                return None
            G = np.atleast_2d(np.asarray(out[2 + k], dtype=np.float64))          # GradConstr 'on' (:101): columns = gradients
            G = G.reshape(nV0, -1).T
            G = np.hstack([G, np.zeros((G.shape[0], nFREE))])
            return -G if k == 0 else G
        probe = control.compute_nonlcon(v0[:nV0])
        for k, kind in ((0, "ineq"), (1, "eq")):
            if np.size(probe[k]):
                c = {"type": kind, "fun": (lambda v, k=k: _nl(v, k))}
                if len(probe) >= 4:
                    c["jac"] = (lambda v, k=k: _nlj(v, k))
                cons.append(c)
    if hasattr(control, "compute_lincon"):                                       # :106-111  A v <= b
        A, b = control.compute_lincon(prob.ControlBounds)
        A, b = np.atleast_2d(np.asarray(A, dtype=np.float64)), np.asarray(b, dtype=np.float64).ravel()
        if A.size:
            Af = np.hstack([A, np.zeros((A.shape[0], nFREE))])
            cons.append({"type": "ineq", "fun": (lambda v: b - Af @ v), "jac": (lambda v: -Af)})

    hist = []

    seen = {}

    def fun(v):                                                                  # nlpObjective :137-150
        J, dJdv, x0n = nlp_objective(integrator, prob, control, x0, v, FreeInitStates)
        hist.append(J)
        seen[np.asarray(v, dtype=np.float64).tobytes()] = J
        return J, dJdv

    class _StepBelowTolX(Exception):
        pass

    # TolX (:20, fmincon's step tolerance): SLSQP has no such option -- the iteration is ended from the callback when
    # an iterate moved by less than TolX in every coefficient.  TolFun (:21) goes to SLSQP's ftol, the tolerance on the
    # change of the objective (fmincon's is a first-order optimality measure): 1e-3 of it, as before, because SLSQP's
    # test is relative to nothing and 3e-4 of an O(10) objective stops it far from the KKT point.
    last = {"v": np.array(v0, dtype=np.float64), "stopped": False}

    def cb(vk):
        step = float(np.max(np.abs(vk - last["v"]))) if vk.size else 0.0
        last["v"] = np.array(vk, dtype=np.float64)
        if opt["TolX"] and 0.0 < step < opt["TolX"]:
            last["stopped"] = True
            raise _StepBelowTolX

    try:
        res = minimize(fun, v0, jac=True, method="SLSQP", bounds=bounds, constraints=cons, callback=cb,
                       options={"ftol": opt["TolFun"] * 1e-3, "maxiter": opt["MaxIter"], "disp": bool(opt["Reporting"])})
    except _StepBelowTolX:
        from scipy.optimize import OptimizeResult
        vl = last["v"]
        Jl = seen.get(vl.tobytes())
        if Jl is None:
            Jl = fun(vl)[0]
        res = OptimizeResult(x=vl, fun=Jl, nfev=len(hist), success=True, message="step smaller than TolX")
    vOpt = res.x
    soln = {"J": -res.fun if MinMax == "Max" else res.fun}                       # :117-119
    nV = vOpt.size - nFREE
    uOpt = control.compute_u(vOpt[:nV])                                          # :121
    if nFREE:
        x0[np.array(FreeInitStates) - 1] = vOpt[nV:]                             # :122-124
    xOpt, _ = integrator.compute_states(prob, x0, uOpt)                          # :125
    lamOpt = integrator.compute_adjoints(prob, uOpt, nargout=1)                  # :126
    soln["u"] = control.compute_uFunc(vOpt[:nV])                                 # :128
    soln["x"] = vectorInterpolant(tspan, xOpt[:-1, :], "pchip")                  # :129
    soln["lam"] = vectorInterpolant(tspan, lamOpt[:-1, :], "pchip")              # :130
    soln["_v"], soln["_nfev"], soln["_message"] = vOpt, res.nfev, res.message
    soln["_stopped_on_TolX"], soln["_constraints"] = last["stopped"], len(cons)
    return soln


def compute_equilibrium(prob, xGuess, lamGuess, uGuess, lb, ub, r):
    """[xStar, lamStar, uStar, resnorm, residual, exitflag] = compute_equilibrium(prob, xGuess, lamGuess,
    uGuess, lb, ub, r)   functions/compute_equilibrium.m:1-34.
    Steady state of the optimality system: F(0,x,u)(1:nS) = 0, r*lam - dFdx_times_vec(0,x,u,[lam;1])(1:nS) = 0,
    dFdu_times_vec(0,x,u,[lam;1]) = 0 (:10-21), inside lb <= [x; lam; u] <= ub.  Solved on the device
    (ocs_compute_equilibrium: one thread per instance, projected Levenberg-Marquardt on the reference's residual;
    lsqnonlin is a MATLAB toolbox).  Guesses may carry a trailing batch dimension (nS x B, nS x B, nC x B): every
    column is one instance (with the per-trajectory parameters of `prob`), and the outputs gain that dimension.
    exitflag: 1 converged, 0 iteration limit, -1 the residual is not finite (lsqnonlin errors there); a single
    instance with exitflag -1 raises, a batch reports it per instance."""
    xG, lG, uG = (np.asarray(a, dtype=np.float64) for a in (xGuess, lamGuess, uGuess))
    batched = xG.ndim == 2
    nS, nC = prob.nS, prob.ControlBounds.shape[0]
    B = xG.shape[1] if batched else 1
    yG = np.asfortranarray(np.vstack([xG.reshape(nS, B), lG.reshape(nS, B), uG.reshape(nC, B)]))
    n = 2 * nS + nC
    lb = np.ascontiguousarray(np.asarray(lb, dtype=np.float64).ravel())
    ub = np.ascontiguousarray(np.asarray(ub, dtype=np.float64).ravel())
    if lb.size != n or ub.size != n:
        raise ValueError(f"lb, ub need {n} entries ([x; lam; u])")
    y = np.empty((n, B), order="F")
    res = np.empty((n, B), order="F")
    resnorm = np.empty(B)
    flag = np.empty(B, dtype=np.int32)
    rc = lib.ocs_compute_equilibrium(prob._h, B, float(r), _p(yG), _p(lb), _p(ub), _p(y), _p(resnorm), _p(res),
                                     flag.ctypes.data_as(C.POINTER(C.c_int)))
    check(rc)
    if rc > 0 and not batched:   # one instance: undefined values are an error, as in lsqnonlin
        raise OcsError(rc, "compute_equilibrium: the residual of the optimality system is not finite at the guess")
    if batched:
        return y[:nS], y[nS:2 * nS], y[2 * nS:], resnorm, res, flag
    return y[:nS, 0], y[nS:2 * nS, 0], y[2 * nS:, 0], float(resnorm[0]), res[:, 0], int(flag[0])


def compute_equilibrium_dev(prob, yGuess, lb, ub, r):
    """device, batch-minor: yGuess [2 nS + nC][B] -> (y [n][B], resnorm [B], residual [n][B], exitflag [B] int32);
    lb, ub: device [n].  Asynchronous on the current stream."""
    n, B = yGuess.shape
    y, res = torch.empty_like(yGuess), torch.empty_like(yGuess)
    resnorm = torch.empty(B, dtype=torch.float64, device=yGuess.device)
    flag = torch.empty(B, dtype=torch.int32, device=yGuess.device)
    assert lb.numel() == n and ub.numel() == n and yGuess.is_contiguous() and yGuess.dtype == torch.float64
    check(lib.ocs_compute_equilibrium_dev(prob._h, B, float(r), _dptr(yGuess), _dptr(lb), _dptr(ub), _dptr(y),
                                          _dptr(resnorm), _dptr(res), C.c_void_p(flag.data_ptr()), _stream()))
    return y, resnorm, res, flag


def single_shooting_batch(prob, x0, tspan, nCONTROL_PTS, Control=None, Integrator=None, u0=0.0, TolX=1e-5,
                          TolFun=3e-4, MaxIter=500, memory=10, FreeInitStates=(), FreeStateBounds=None, v0=None,
                          constraints="warn"):
    """Batched direct single shooting: B independent NLPs  min_v J_b(v), Lb <= v <= Ub  --
    one per column of x0 and/or per per-trajectory parameter set of `prob` -- solved together on the GPU by the
    library's ocs_single_shooting_batch_dev.

    The reference runs fmincon('sqp') on one problem at a time (single_shooting.m:114); fmincon is a MATLAB
    toolbox, so the outer iteration is a batched spectral projected gradient (Birgin-Martinez-Raydan SPG:
    Barzilai-Borwein step, projection on the bounds of compute_nlp_bounds, non-monotone Armijo back-tracking),
    every instance with its own step length and stopping test, every objective/gradient evaluation one call
    of the hot path (nlpObjective) over the whole batch.  Iterates differ from fmincon's; the optimum
    is the same KKT point.  Stopping (per instance): ||P(v - g) - v||_inf <= TolFun or step <= TolX.
    Free initial states ride at the tail of v, starting at x0 (single_shooting.m:81-85), inside FreeStateBounds
    (nFree x 2, :91-94; default unbounded).
    Returns a dict of device tensors: v [nV(+nFree)][B], J [B], iterations [B], converged [B],
    projected_gradient [B], x0 [nS][B] and `soln_of(b)` (J, v and the control callable of instance b)."""
    from ._lib import SsOptions
    tspan = _f(tspan).ravel()
    integrator = Integrator or RK4Integrator(tspan)
    nC = prob.ControlBounds.shape[0]
    control = Control or PWLinearControl(integrator.t, nCONTROL_PTS, nC)
    dev = torch.device("cuda", torch.cuda.current_device())
    x0 = np.asarray(x0, dtype=np.float64).reshape(prob.nS, -1)
    B = x0.shape[1]
    x0d = torch.tensor(x0, device=dev).contiguous()
    nFree, fis = _fis(FreeInitStates)
    if v0 is None:
        u0c = np.minimum(prob.ControlBounds[:, 1], np.maximum(prob.ControlBounds[:, 0],
                                                               np.broadcast_to(np.asarray(u0, dtype=np.float64).ravel(), (nC,))))
        v0 = np.repeat(control.compute_initial_v(u0c)[:, None], B, axis=1)
    else:
        v0 = np.asarray(v0, dtype=np.float64)
        v0 = np.repeat(v0[:, None], B, axis=1) if v0.ndim == 1 else v0
    nV = v0.shape[0]
    if hasattr(control, "compute_nlp_bounds"):
        Lb, Ub = control.compute_nlp_bounds(prob.ControlBounds)
    else:
        Lb, Ub = np.full(nV, -np.inf), np.full(nV, np.inf)
    # The projected gradient iteration projects on a BOX: the constraint hooks of single_shooting.m:99-111
    # (compute_nonlcon, compute_lincon) have no counterpart here.  A control that defines them (ChebyshevControl's bounds
    # at the grid points) is iterated without them -- said once, not silently; single_shooting() honours them.
    if constraints == "error" and (hasattr(control, "compute_lincon") or hasattr(control, "compute_nonlcon")):
        raise ValueError("single_shooting_batch takes box bounds (compute_nlp_bounds) only; this control defines "
                         "compute_lincon / compute_nonlcon -- use single_shooting, or pass constraints='ignore'")
    if constraints == "warn" and (hasattr(control, "compute_lincon") or hasattr(control, "compute_nonlcon")):
        import warnings
        warnings.warn("single_shooting_batch: compute_lincon / compute_nonlcon of the control are not applied (box bounds "
                      "only); single_shooting honours them", RuntimeWarning, stacklevel=2)
    if nFree:                                                             # :84: v0 = [v0; x0(FreeInitStates)]
        idx = [int(i) - 1 for i in np.asarray(FreeInitStates).ravel()]
        v0 = np.vstack([v0, x0[idx, :]])
        fsb = (np.tile([-np.inf, np.inf], (nFree, 1)) if FreeStateBounds is None
               else np.asarray(FreeStateBounds, dtype=np.float64).reshape(nFree, 2))
        Lb = np.concatenate([Lb, fsb[:, 0]])                              # :91-94
        Ub = np.concatenate([Ub, fsb[:, 1]])
    Lb, Ub = _f(Lb).ravel().copy(), _f(Ub).ravel().copy()
    v = torch.tensor(np.ascontiguousarray(v0), device=dev).contiguous()
    J = torch.empty(B, dtype=torch.float64, device=dev)
    pgn = torch.empty(B, dtype=torch.float64, device=dev)
    iters = torch.empty(B, dtype=torch.int32, device=dev)
    conv = torch.empty(B, dtype=torch.int32, device=dev)
    o = SsOptions()
    check(lib.ocs_ss_default_options(C.byref(o)))
    o.TolX, o.TolFun, o.MaxIter, o.memory = float(TolX), float(TolFun), int(MaxIter), int(memory)
    rc = lib.ocs_single_shooting_batch_dev(integrator._h, prob._h, control._h, B, _dptr(x0d), _dptr(v), nFree, fis,
                                           _p(Lb), _p(Ub), C.byref(o), _dptr(J), C.c_void_p(iters.data_ptr()), C.c_void_p(conv.data_ptr()), _dptr(pgn),
                                           _stream())
    if rc < 0:
        check(rc)
    out = {"v": v, "J": J, "iterations": iters, "converged": conv.to(torch.bool), "projected_gradient": pgn, "x0": x0d,
           "control": control, "integrator": integrator}

    def soln_of(b):
        """J, the coefficient vector and soln.u (compute_uFunc) of instance b."""
        vb = v[:, b].cpu().numpy()
        return {"J": float(J[b]), "v": vb, "u": control.compute_uFunc(vb[:vb.size - nFree])}

    out["soln_of"] = soln_of
    return out
