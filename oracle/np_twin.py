"""Independent NumPy restatement of RK4Integrator / Control / TestOCProblem.

TEST INFRASTRUCTURE ONLY (same rule as oracle/ocs_oracle.h): a second, separately
written twin of the C oracle so that a transcription slip in one of them shows up
as a disagreement.  Works for real and complex dtypes, which gives the complex-step
derivative used to pin the discrete adjoint (F is analytic: polynomials and exp).
PARITY UNPINNED with respect to MATLAB itself (never run in this pipeline).

Written from the reference's semantics, vectorised over time where MATLAB is:
  Integrator/RK4Integrator.m:16-121, Control/PWLinearControl.m:31-62,
  Control/PWConstantControl.m:41-50, Control/ChebyshevControl.m:21-31,
  tests/TestOCProblem.m:22-38.
"""
from __future__ import annotations

import numpy as np


def matlab_linspace(a, b, n):
    k = np.arange(n, dtype=np.float64)
    y = a + (k * (b - a)) / (n - 1)
    y[0], y[-1] = a, b
    return y


class TestOCProblemNP:
    """tests/TestOCProblem.m with an optional vector of m (LogisticK, nS = len(m))."""
    __test__ = False

    def __init__(self, c, m, r, bounds=(0.0, 1.0)):
        self.c, self.r = c, r
        self.m = np.atleast_1d(np.asarray(m, dtype=np.float64))
        self.nS = self.m.size
        self.nC = 1
        self.ControlBounds = np.asarray(bounds, dtype=np.float64).reshape(1, 2)

    def F(self, t, y, u):
        x = y[: self.nS]
        e = np.exp(-self.r * t)
        top = x * (self.m[:, None] - x) - u
        s = x[0] * x[0]
        for k in range(1, self.nS):
            s = s + x[k] * x[k]
        return np.vstack([top, e * (s + self.c * (u[0] * u[0]))])

    def dFdx_times_vec(self, t, y, u, v):
        x = y[: self.nS]
        e = np.exp(-self.r * t)
        top = (self.m[:, None] - 2 * x) * v[: self.nS] + 2 * e * x * v[self.nS]
        return np.vstack([top, np.zeros_like(top[:1])])

    def dFdu_times_vec(self, t, y, u, v):
        s = -v[0]
        for k in range(1, self.nS):
            s = s - v[k]
        return (s + 2 * self.c * np.exp(-self.r * t) * u[0] * v[self.nS])[None, :]


class RK4IntegratorNP:
    def __init__(self, tspan):
        tspan = np.asarray(tspan, dtype=np.float64)
        self.h = np.diff(tspan)
        self.nSTEPS = self.h.size
        t = np.zeros(2 * self.nSTEPS + 1)
        t[0::2] = tspan
        t[1:-1:2] = (tspan[:-1] + tspan[1:]) / 2
        self.t = t
        self.xK = None

    def compute_states(self, prob, x0, u):
        N, t, h = self.nSTEPS, self.t, self.h
        x0 = np.atleast_1d(x0)
        dt = np.result_type(x0.dtype, u.dtype, np.float64)
        nA = x0.size + 1
        xK = np.full((nA, N + 1, 4), np.nan, dtype=dt)
        xK[:, 0, 0] = np.concatenate([x0, [0.0]])
        c = lambda a: a.reshape(-1, 1)
        for i in range(N):
            y = xK[:, i, 0]
            F1 = prob.F(t[2 * i], c(y), c(u[:, 2 * i]))[:, 0]
            xK[:, i, 1] = y + h[i] / 2 * F1
            F2 = prob.F(t[2 * i + 1], c(xK[:, i, 1]), c(u[:, 2 * i + 1]))[:, 0]
            xK[:, i, 2] = y + h[i] / 2 * F2
            F3 = prob.F(t[2 * i + 1], c(xK[:, i, 2]), c(u[:, 2 * i + 1]))[:, 0]
            xK[:, i, 3] = y + h[i] * F3
            F4 = prob.F(t[2 * i + 2], c(xK[:, i, 3]), c(u[:, 2 * i + 2]))[:, 0]
            xK[:, i + 1, 0] = y + h[i] / 6 * (F1 + 2 * F2 + 2 * F3 + F4)
        self.xK = xK
        x = xK[:, :, 0]
        return x, x[-1, -1]

    def compute_adjoints(self, prob, u, lamT=None):
        xK, N, t, h = self.xK, self.nSTEPS, self.t, self.h
        nA = xK.shape[0]
        if lamT is None:
            lamT = np.zeros(nA)
            lamT[-1] = 1
        lam = np.full((nA, N + 1), np.nan, dtype=xK.dtype)
        lam[:, -1] = lamT
        dJdk = np.full((nA, N, 4), np.nan, dtype=xK.dtype)
        c = lambda a: a.reshape(-1, 1)
        for i in range(N - 1, -1, -1):
            dJdk[:, i, 3] = h[i] / 6 * lam[:, i + 1]
            g3 = prob.dFdx_times_vec(t[2 * i + 2], c(xK[:, i, 3]), c(u[:, 2 * i + 2]), c(dJdk[:, i, 3]))[:, 0]
            dJdk[:, i, 2] = h[i] / 3 * lam[:, i + 1] + h[i] * g3
            g2 = prob.dFdx_times_vec(t[2 * i + 1], c(xK[:, i, 2]), c(u[:, 2 * i + 1]), c(dJdk[:, i, 2]))[:, 0]
            dJdk[:, i, 1] = h[i] / 3 * lam[:, i + 1] + h[i] / 2 * g2
            g1 = prob.dFdx_times_vec(t[2 * i + 1], c(xK[:, i, 1]), c(u[:, 2 * i + 1]), c(dJdk[:, i, 1]))[:, 0]
            dJdk[:, i, 0] = h[i] / 6 * lam[:, i + 1] + h[i] / 2 * g1
            g0 = prob.dFdx_times_vec(t[2 * i], c(xK[:, i, 0]), c(u[:, 2 * i]), c(dJdk[:, i, 0]))[:, 0]
            lam[:, i] = lam[:, i + 1] + g1 + g2 + g3 + g0
        # compute_dJdu, time-vectorised exactly like RK4Integrator.m:97-121
        dJdu = np.zeros(u.shape, dtype=xK.dtype)
        dJdu[:, 0] = prob.dFdu_times_vec(t[0:1], xK[:, 0:1, 0], u[:, 0:1], dJdk[:, 0:1, 0])[:, 0]
        dJdu[:, 1:-1:2] = (prob.dFdu_times_vec(t[1:-1:2], xK[:, :-1, 1], u[:, 1:-1:2], dJdk[:, :, 1])
                           + prob.dFdu_times_vec(t[1:-1:2], xK[:, :-1, 2], u[:, 1:-1:2], dJdk[:, :, 2]))
        if N > 1:
            dJdu[:, 2:-2:2] = (prob.dFdu_times_vec(t[2:-2:2], xK[:, 1:-1, 0], u[:, 2:-2:2], dJdk[:, 1:, 0])
                               + prob.dFdu_times_vec(t[2:-2:2], xK[:, :-2, 3], u[:, 2:-2:2], dJdk[:, :-1, 3]))
        dJdu[:, -1] = prob.dFdu_times_vec(t[-1:], xK[:, -2:-1, 3], u[:, -1:], dJdk[:, -1:, 3])[:, 0]
        return lam, dJdu


def pwlinear_basis(t, nPts):
    """max(0, 1 - |t - c_i| / dc) hats on controlPts = linspace(t0, t1, nPts) (PWLinearControl.m:31-50)."""
    cpts = matlab_linspace(t[0], t[-1], nPts)
    B = np.zeros((nPts, t.size))
    for i in range(nPts):
        if i > 0:
            m = (t >= cpts[i - 1]) & (t <= cpts[i])
            B[i, m] = (t[m] - cpts[i - 1]) / (cpts[i] - cpts[i - 1])
        if i < nPts - 1:
            m = (t >= cpts[i]) & (t <= cpts[i + 1])
            B[i, m] = 1 - (t[m] - cpts[i]) / (cpts[i + 1] - cpts[i])
    return B, cpts


def pwconstant_basis(t, nInt):
    starts = matlab_linspace(t[0], t[-1], nInt + 1)[:-1]
    B = np.zeros((nInt, t.size))
    for i in range(nInt - 1):
        B[i] = (t >= starts[i]) & (t < starts[i + 1])
    B[-1] = t >= starts[-1]
    return B, starts


def chebyshev_basis(t, nB):
    tau = 2 * (t - t[0]) / (t[-1] - t[0]) - 1
    B = np.zeros((nB, t.size))
    B[0] = 1
    if nB > 1:
        B[1] = tau
    for i in range(2, nB):
        B[i] = 2 * tau * B[i - 1] - B[i - 2]
    return B


def objective_and_gradient(prob, integ, B, x0, v, nC=1):
    """single_shooting.m:137-143 for a given basis matrix."""
    u = v.reshape(nC, -1, order="F") @ B
    _, J = integ.compute_states(prob, x0, u)
    lam, dJdu = integ.compute_adjoints(prob, u)
    return J, (dJdu @ B.T).reshape(-1, order="F"), lam
