"""User-written OCProblem plugins used by the tests: device source for the hipRTC path next to a NumPy
restatement of the same three methods (for oracle/np_twin.RK4IntegratorNP)."""
import numpy as np

# ---- LogisticK with nS = 2 written by hand (must agree with the built-in functor and the C oracle) ----
LOGISTIC2_SRC = r"""
// params: [c, r, m1, m2]
__device__ void ocs_F(double t, const double* y, const double* u, OCS_PARAMS p, double* f) {
  f[0] = y[0] * (p[2] - y[0]) - u[0];
  f[1] = y[1] * (p[3] - y[1]) - u[0];
  f[2] = exp(-p[1] * t) * (y[0] * y[0] + y[1] * y[1] + p[0] * u[0] * u[0]);
}
__device__ void ocs_dFdx_times_vec(double t, const double* y, const double* u, OCS_PARAMS p, const double* v, double* g) {
  const double e = exp(-p[1] * t);
  g[0] = (p[2] - 2 * y[0]) * v[0] + 2 * e * y[0] * v[2];
  g[1] = (p[3] - 2 * y[1]) * v[1] + 2 * e * y[1] * v[2];
}
__device__ void ocs_dFdu_times_vec(double t, const double* y, const double* u, OCS_PARAMS p, const double* v, double* g) {
  g[0] = -v[0] - v[1] + 2 * p[0] * exp(-p[1] * t) * u[0] * v[2];
}
__device__ void ocs_ControlChar(double t, const double* x, const double* lam, OCS_PARAMS p, const double* lb,
                                const double* ub, double* u) {
  u[0] = fmin(ub[0], fmax(lb[0], (lam[0] + lam[1]) * exp(p[1] * t) / (2 * p[0])));
}
"""

# ---- LogisticK written as ROW FUNCTIONS (row-separable user problems: the fast mappings of the registry problems) ----
# params: [c, r, m_1 .. m_NS]; the control cost c u^2 is charged to row 0
LOGISTIC_ROWS_SRC = r"""
__device__ double ocs_row_tcoef(double t, OCS_PARAMS p) { return exp(-p[1] * t); }   // hoisted: once per grid point
__device__ double ocs_row_F(double tc, double y, double u, OCS_PARAMS p, int r) { return y * (p[2 + r] - y) - u; }
__device__ double ocs_row_q(double tc, double y, double u, OCS_PARAMS p, int r) {
  return tc * (y * y + (r == 0 ? p[0] * u * u : 0.0));
}
__device__ void ocs_row_dFdy(double tc, double y, double u, OCS_PARAMS p, int r, double* dF, double* dq) {
  *dF = p[2 + r] - 2 * y;
  *dq = 2 * tc * y;
}
__device__ void ocs_row_dFdu(double tc, double y, double u, OCS_PARAMS p, int r, double* dF, double* dq) {
  *dF = -1.0;
  *dq = r == 0 ? 2 * p[0] * tc * u : 0.0;
}
"""

# ... with the minimum-principle control of the same problem (the Gen-1 ControlChar of make_from_symbolic.m:33-38 for this
# Hamiltonian): it reads the costate only, and ocs_row_dFdy above does not read u -> control_from_costate
LOGISTIC_ROWS_CC_SRC = LOGISTIC_ROWS_SRC + r"""
__device__ void ocs_ControlChar(double t, const double* x, const double* lam, OCS_PARAMS p, const double* lb,
                                const double* ub, double* u) {
  double s = lam[0];
  for (int k = 1; k < NS; ++k) s += lam[k];
  u[0] = fmin(ub[0], fmax(lb[0], s * exp(p[1] * t) / (2 * p[0])));
}
"""

# ... and with the exponential of ControlChar hoisted into the integrator's tables (OCS_USER_CC_TCOEF: ocs_ControlChar
# receives ocs_cc_tcoef(t, p) = e^{r t} in the place of t)
LOGISTIC_ROWS_CCT_SRC = LOGISTIC_ROWS_SRC + r"""
#define OCS_USER_CC_TCOEF 1
__device__ double ocs_cc_tcoef(double t, OCS_PARAMS p) { return exp(p[1] * t); }
__device__ void ocs_ControlChar(double ert, const double* x, const double* lam, OCS_PARAMS p, const double* lb,
                                const double* ub, double* u) {
  double s = lam[0];
  for (int k = 1; k < NS; ++k) s += lam[k];
  u[0] = fmin(ub[0], fmax(lb[0], s * ert / (2 * p[0])));
}
"""

# a row-separable problem whose control enters multiplicatively (dF/du reads y): x_r' = x_r (m_r - x_r) - u x_r / (1 + r),
# cost' = e^{-rt} (sum x_r^2 + c u^2);   params [c, r, m_1 .. m_NS]
PROPHARVEST_ROWS_SRC = r"""
__device__ double ocs_row_tcoef(double t, OCS_PARAMS p) { return exp(-p[1] * t); }
__device__ double ocs_row_F(double tc, double y, double u, OCS_PARAMS p, int r) { return y * (p[2 + r] - y) - u * y / (1 + r); }
__device__ double ocs_row_q(double tc, double y, double u, OCS_PARAMS p, int r) {
  return tc * (y * y + (r == 0 ? p[0] * u * u : 0.0));
}
__device__ void ocs_row_dFdy(double tc, double y, double u, OCS_PARAMS p, int r, double* dF, double* dq) {
  *dF = p[2 + r] - 2 * y - u / (1 + r);
  *dq = 2 * tc * y;
}
__device__ void ocs_row_dFdu(double tc, double y, double u, OCS_PARAMS p, int r, double* dF, double* dq) {
  *dF = -y / (1 + r);
  *dq = r == 0 ? 2 * p[0] * tc * u : 0.0;
}
"""

# ... with its minimum-principle control, which reads x (and ocs_row_dFdy above reads u): no control_from_costate
PROPHARVEST_ROWS_CC_SRC = PROPHARVEST_ROWS_SRC + r"""
__device__ void ocs_ControlChar(double t, const double* x, const double* lam, OCS_PARAMS p, const double* lb,
                                const double* ub, double* u) {
  double s = 0.0;
  for (int k = 0; k < NS; ++k) s += lam[k] * x[k] / (1 + k);
  u[0] = fmin(ub[0], fmax(lb[0], s * exp(p[1] * t) / (2 * p[0])));
}
"""


class PropHarvestNP:
    """NumPy twin of PROPHARVEST_ROWS_SRC with the OCProblem method signatures (columns vectorised)."""
    nC = 1

    def __init__(self, m, c, r):
        self.m, self.c, self.r = np.asarray(m, dtype=np.float64), float(c), float(r)
        self.nS = self.m.size
        self.sc = 1.0 / (1 + np.arange(self.nS))

    def F(self, t, y, u):
        x = y[:self.nS]
        f = x * (self.m[:, None] - x) - u[0] * x * self.sc[:, None]
        return np.vstack([f, np.exp(-self.r * t) * (np.sum(x * x, axis=0) + self.c * u[0] * u[0])])

    def dFdx_times_vec(self, t, y, u, v):
        x = y[:self.nS]
        g = (self.m[:, None] - 2 * x - u[0] * self.sc[:, None]) * v[:self.nS] + 2 * np.exp(-self.r * t) * x * v[self.nS]
        return np.vstack([g, np.zeros_like(g[:1])])

    def dFdu_times_vec(self, t, y, u, v):
        x = y[:self.nS]
        return (np.sum(-x * self.sc[:, None] * v[:self.nS], axis=0) + 2 * self.c * np.exp(-self.r * t) * u[0] * v[self.nS])[None, :]


# ---- predator-prey with harvested predator: coupled, not row-separable, not in the registry ----
# x1' = x1 (al - be x2),  x2' = x2 (de x1 - ga) - u x2,  cost' = e^{-rt} (c u^2 + q (x1 - xb)^2)
# params: [al, be, de, ga, c, q, xb, r]
PREDPREY_SRC = r"""
__device__ void ocs_F(double t, const double* y, const double* u, OCS_PARAMS p, double* f) {
  f[0] = y[0] * (p[0] - p[1] * y[1]);
  f[1] = y[1] * (p[2] * y[0] - p[3]) - u[0] * y[1];
  const double d = y[0] - p[6];
  f[2] = exp(-p[7] * t) * (p[4] * u[0] * u[0] + p[5] * d * d);
}
__device__ void ocs_dFdx_times_vec(double t, const double* y, const double* u, OCS_PARAMS p, const double* v, double* g) {
  const double e = exp(-p[7] * t);
  g[0] = (p[0] - p[1] * y[1]) * v[0] + p[2] * y[1] * v[1] + e * 2 * p[5] * (y[0] - p[6]) * v[2];
  g[1] = -p[1] * y[0] * v[0] + (p[2] * y[0] - p[3] - u[0]) * v[1];
}
__device__ void ocs_dFdu_times_vec(double t, const double* y, const double* u, OCS_PARAMS p, const double* v, double* g) {
  g[0] = -y[1] * v[1] + exp(-p[7] * t) * 2 * p[4] * u[0] * v[2];
}
"""
# ... the same problem with the discount factor hoisted into the integrator's tables (OCS_USER_TCOEF: the methods receive
# ocs_tcoef(t, p) = e^{-rt} in the place of t; the dynamics do not read the time)
PREDPREY_TC_SRC = r"""
#define OCS_USER_TCOEF 1
__device__ double ocs_tcoef(double t, OCS_PARAMS p) { return exp(-p[7] * t); }
__device__ void ocs_F(double e, const double* y, const double* u, OCS_PARAMS p, double* f) {
  f[0] = y[0] * (p[0] - p[1] * y[1]);
  f[1] = y[1] * (p[2] * y[0] - p[3]) - u[0] * y[1];
  const double d = y[0] - p[6];
  f[2] = e * (p[4] * u[0] * u[0] + p[5] * d * d);
}
__device__ void ocs_dFdx_times_vec(double e, const double* y, const double* u, OCS_PARAMS p, const double* v, double* g) {
  g[0] = (p[0] - p[1] * y[1]) * v[0] + p[2] * y[1] * v[1] + e * 2 * p[5] * (y[0] - p[6]) * v[2];
  g[1] = -p[1] * y[0] * v[0] + (p[2] * y[0] - p[3] - u[0]) * v[1];
}
__device__ void ocs_dFdu_times_vec(double e, const double* y, const double* u, OCS_PARAMS p, const double* v, double* g) {
  g[0] = -y[1] * v[1] + e * 2 * p[4] * u[0] * v[2];
}
"""
PREDPREY_PARAMS = [1.0, 0.5, 0.3, 0.6, 2.0, 1.5, 1.8, 0.05]


class PredPreyNP:
    """NumPy twin of PREDPREY_SRC with the OCProblem method signatures (columns vectorised)."""
    nS, nC = 2, 1

    def __init__(self, p=PREDPREY_PARAMS):
        self.p = np.asarray(p, dtype=np.float64)

    def F(self, t, y, u):
        p = self.p
        d = y[0] - p[6]
        return np.vstack([y[0] * (p[0] - p[1] * y[1]), y[1] * (p[2] * y[0] - p[3]) - u[0] * y[1],
                          np.exp(-p[7] * t) * (p[4] * u[0] * u[0] + p[5] * d * d)])

    def dFdx_times_vec(self, t, y, u, v):
        p = self.p
        e = np.exp(-p[7] * t)
        g0 = (p[0] - p[1] * y[1]) * v[0] + p[2] * y[1] * v[1] + e * 2 * p[5] * (y[0] - p[6]) * v[2]
        g1 = -p[1] * y[0] * v[0] + (p[2] * y[0] - p[3] - u[0]) * v[1]
        return np.vstack([g0, g1, np.zeros_like(g0)])

    def dFdu_times_vec(self, t, y, u, v):
        p = self.p
        return (-y[1] * v[1] + np.exp(-p[7] * t) * 2 * p[4] * u[0] * v[2])[None, :]


def lq_source(nS, nC):
    """Device source of the build-defined LQ problem (SURVEY BL-5): F = [A x + Bu u ; e^{-rt}(x'Qx + u'Ru)],
    params [r | A (nS x nS col-major) | Bu (nS x nC) | q (nS) | rdiag (nC)] as in oracle/ocs_oracle.c."""
    oA, oB, oq, oR = 1, 1 + nS * nS, 1 + nS * nS + nS * nC, 1 + nS * nS + nS * nC + nS
    return f"""
__device__ void ocs_F(double t, const double* y, const double* u, OCS_PARAMS p, double* f) {{
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < {nS}; ++i) {{
    double a = 0.0;
#pragma unroll
    for (int l = 0; l < {nS}; ++l) a += p[{oA} + i + {nS} * l] * y[l];
#pragma unroll
    for (int l = 0; l < {nC}; ++l) a += p[{oB} + i + {nS} * l] * u[l];
    f[i] = a;
    s += p[{oq} + i] * (y[i] * y[i]);
  }}
#pragma unroll
  for (int l = 0; l < {nC}; ++l) s += p[{oR} + l] * (u[l] * u[l]);
  f[{nS}] = exp(-p[0] * t) * s;
}}
__device__ void ocs_dFdx_times_vec(double t, const double* y, const double* u, OCS_PARAMS p, const double* v, double* g) {{
  const double e = exp(-p[0] * t);
#pragma unroll
  for (int i = 0; i < {nS}; ++i) {{
    double a = 0.0;
#pragma unroll
    for (int l = 0; l < {nS}; ++l) a += p[{oA} + l + {nS} * i] * v[l];
    g[i] = a + 2 * e * p[{oq} + i] * y[i] * v[{nS}];
  }}
}}
__device__ void ocs_dFdu_times_vec(double t, const double* y, const double* u, OCS_PARAMS p, const double* v, double* g) {{
  const double e = exp(-p[0] * t);
#pragma unroll
  for (int l = 0; l < {nC}; ++l) {{
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < {nS}; ++i) a += p[{oB} + i + {nS} * l] * v[i];
    g[l] = a + 2 * e * p[{oR} + l] * u[l] * v[{nS}];
  }}
}}
"""


def lq_matrices(nS, nC, seed=20260405):
    """BL-5 style data: A = -diag(logspace(0, 1.5, nS)) + 0.1 G, seeded (mildly stiff for test step sizes)."""
    rng = np.random.default_rng(seed)
    A = -np.diag(np.logspace(0, 1.5, nS)) + 0.1 * rng.normal(size=(nS, nS))
    Bu = rng.normal(size=(nS, nC))
    q = rng.uniform(0.5, 1.5, nS)
    rdiag = rng.uniform(1.0, 2.0, nC)
    return A, Bu, q, rdiag


def ring6_symbolic(sym):
    """A coupled problem beyond the vector mappings (nS = 6 > 4, nC = 3 > 2), from symbols: six logistic stocks on a ring with
    diffusive exchange, three harvest efforts (effort j works stocks j and j + 3), discounted quadratic objective.
    Returns (objective, stateRHS, params) for optimal-control-solvers_amd/symbolic.py."""
    import sympy as sp
    names = ["r", "kap", "c1", "c2", "c3"] + [f"m{k + 1}" for k in range(6)]
    t, x, lam, u, p = sym.symbols(6, 3, names)
    f = [x[i] * (p[f"m{i + 1}"] - x[i]) + p["kap"] * (x[(i + 1) % 6] - 2 * x[i] + x[(i - 1) % 6]) - u[i % 3] * x[i]
         for i in range(6)]
    g = sp.exp(-p["r"] * t) * (sum((x[i] - 1) ** 2 for i in range(6)) + sum(p[f"c{j + 1}"] * u[j] ** 2 for j in range(3)))
    vals = {"r": 0.05, "kap": 0.3, "c1": 1.5, "c2": 2.0, "c3": 1.2,
            **{f"m{k + 1}": v for k, v in enumerate([3.0, 2.5, 2.0, 2.8, 2.2, 1.8])}}
    return g, f, vals
