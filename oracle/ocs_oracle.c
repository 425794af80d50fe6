/*
 * ocs_oracle.c -- CPU restatement of the reference hot path.  See ocs_oracle.h:
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (no reference goldens exist, MATLAB is
 * not runnable in this pipeline).
 *
 * Compile with -ffp-contract=off: MATLAB evaluates every * and + as a separately
 * rounded IEEE fp64 operation, so the restatement must not fuse them.
 *
 * Layout conventions are MATLAB's: column-major, 1-based indices in the comments
 * that quote the reference, 0-based in the C code.
 */
#include "ocs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* problems                                                                  */
/* ------------------------------------------------------------------------- */
struct ocs_or_problem {
  int id, nS, nC, nAug;
  double *par;
  int npar;
  double *bounds; /* nC x 2 */
};

ocs_or_problem *ocs_or_problem_create(int id, int nS, int nC, const double *params, int nparams,
                                      const double *bounds) {
  int need = -1;
  if (id == OCS_OR_PROBLEM_TEST) {
    if (nS != 1 || nC != 1) return NULL;
    need = 3;
  } else if (id == OCS_OR_PROBLEM_LOGISTIC) {
    if (nS < 1 || nC != 1) return NULL;
    need = 2 + nS;
  } else if (id == OCS_OR_PROBLEM_LQ) {
    if (nS < 1 || nC < 1) return NULL;
    need = 1 + nS * nS + nS * nC + nS + nC;
  }
  if (need < 0 || nparams != need) return NULL;
  ocs_or_problem *p = (ocs_or_problem *)calloc(1, sizeof(*p));
  p->id = id;
  p->nS = nS;
  p->nC = nC;
  p->nAug = nS + 1;
  p->npar = nparams;
  p->par = (double *)malloc(sizeof(double) * (size_t)nparams);
  memcpy(p->par, params, sizeof(double) * (size_t)nparams);
  p->bounds = (double *)malloc(sizeof(double) * 2 * (size_t)nC);
  memcpy(p->bounds, bounds, sizeof(double) * 2 * (size_t)nC);
  return p;
}
void ocs_or_problem_destroy(ocs_or_problem *p) {
  if (!p) return;
  free(p->par);
  free(p->bounds);
  free(p);
}
int ocs_or_problem_nS(const ocs_or_problem *p) { return p->nS; }
int ocs_or_problem_nC(const ocs_or_problem *p) { return p->nC; }

/* LQ parameter block accessors: [r | A nS x nS | Bu nS x nC | q nS | rdiag nC] */
static const double *lq_A(const ocs_or_problem *p) { return p->par + 1; }
static const double *lq_Bu(const ocs_or_problem *p) { return p->par + 1 + p->nS * p->nS; }
static const double *lq_q(const ocs_or_problem *p) { return p->par + 1 + p->nS * p->nS + p->nS * p->nC; }
static const double *lq_R(const ocs_or_problem *p) {
  return p->par + 1 + p->nS * p->nS + p->nS * p->nC + p->nS;
}

/* value = F(obj, t, y, u): tests/TestOCProblem.m:22-26
 *   x = y(1,:);  value = [x.*(m - x) - u ; exp(-r*t).*(x.^2 + c*u.^2)];
 * LogisticK (build-defined): row k  x_k.*(m_k - x_k) - u,
 *   last row exp(-r*t).*(sum_k x_k.^2 + c*u.^2)   (nS = 1 is TestOCProblem).
 * LQ (build-defined, BL-5): [A*x + Bu*u ; exp(-r*t).*(sum q_k x_k^2 + sum R_c u_c^2)]. */
void ocs_or_F(const ocs_or_problem *p, int k, const double *t, const double *y, const double *u,
              double *out) {
  const int nS = p->nS, nC = p->nC, nAug = p->nAug;
  for (int j = 0; j < k; ++j) {
    const double *yj = y + (size_t)j * nAug;
    const double *uj = u + (size_t)j * nC;
    double *oj = out + (size_t)j * nAug;
    if (p->id == OCS_OR_PROBLEM_TEST) {
      const double c = p->par[0], m = p->par[1], r = p->par[2];
      const double x = yj[0];
      oj[0] = x * (m - x) - uj[0];
      oj[1] = exp(-r * t[j]) * (x * x + c * (uj[0] * uj[0]));
    } else if (p->id == OCS_OR_PROBLEM_LOGISTIC) {
      const double c = p->par[0], r = p->par[1];
      const double *m = p->par + 2;
      double s = 0.0;
      for (int i = 0; i < nS; ++i) {
        const double x = yj[i];
        oj[i] = x * (m[i] - x) - uj[0];
        s = (i == 0) ? x * x : s + x * x;
      }
      oj[nS] = exp(-r * t[j]) * (s + c * (uj[0] * uj[0]));
    } else {
      const double r = p->par[0];
      const double *A = lq_A(p), *Bu = lq_Bu(p), *q = lq_q(p), *R = lq_R(p);
      double s = 0.0;
      for (int i = 0; i < nS; ++i) {
        double a = 0.0;
        for (int l = 0; l < nS; ++l) a += A[i + (size_t)l * nS] * yj[l];
        for (int l = 0; l < nC; ++l) a += Bu[i + (size_t)l * nS] * uj[l];
        oj[i] = a;
        s += q[i] * (yj[i] * yj[i]);
      }
      for (int l = 0; l < nC; ++l) s += R[l] * (uj[l] * uj[l]);
      oj[nS] = exp(-r * t[j]) * s;
    }
  }
}

/* value = dFdx_times_vec(obj, t, y, ~, v): tests/TestOCProblem.m:29-33
 *   value = [(m - 2*x).*v(1,:) + 2*exp(-r*t).*x.*v(2,:) ; 0];     (= (dF/dy)' * v) */
void ocs_or_dFdx_times_vec(const ocs_or_problem *p, int k, const double *t, const double *y,
                           const double *u, const double *v, double *out) {
  (void)u;
  const int nS = p->nS, nAug = p->nAug;
  for (int j = 0; j < k; ++j) {
    const double *yj = y + (size_t)j * nAug;
    const double *vj = v + (size_t)j * nAug;
    double *oj = out + (size_t)j * nAug;
    if (p->id == OCS_OR_PROBLEM_TEST) {
      const double m = p->par[1], r = p->par[2];
      const double x = yj[0];
      oj[0] = (m - 2 * x) * vj[0] + 2 * exp(-r * t[j]) * x * vj[1];
      oj[1] = 0.0;
    } else if (p->id == OCS_OR_PROBLEM_LOGISTIC) {
      const double r = p->par[1];
      const double *m = p->par + 2;
      const double e = exp(-r * t[j]);
      for (int i = 0; i < nS; ++i) {
        const double x = yj[i];
        oj[i] = (m[i] - 2 * x) * vj[i] + 2 * e * x * vj[nS];
      }
      oj[nS] = 0.0;
    } else {
      const double r = p->par[0];
      const double *A = lq_A(p), *q = lq_q(p);
      const double e = exp(-r * t[j]);
      for (int i = 0; i < nS; ++i) {
        double a = 0.0;
        for (int l = 0; l < nS; ++l) a += A[l + (size_t)i * nS] * vj[l]; /* A' * v */
        oj[i] = a + 2 * e * q[i] * yj[i] * vj[nS];
      }
      oj[nS] = 0.0;
    }
  }
}

/* value = dFdu_times_vec(obj, t, ~, u, v): tests/TestOCProblem.m:36-38
 *   value = -v(1,:) + 2*c*exp(-r*t).*u.*v(2,:);                   (= (dF/du)' * v) */
void ocs_or_dFdu_times_vec(const ocs_or_problem *p, int k, const double *t, const double *y,
                           const double *u, const double *v, double *out) {
  (void)y;
  const int nS = p->nS, nC = p->nC, nAug = p->nAug;
  for (int j = 0; j < k; ++j) {
    const double *uj = u + (size_t)j * nC;
    const double *vj = v + (size_t)j * nAug;
    double *oj = out + (size_t)j * nC;
    if (p->id == OCS_OR_PROBLEM_TEST) {
      const double c = p->par[0], r = p->par[2];
      oj[0] = -vj[0] + 2 * c * exp(-r * t[j]) * uj[0] * vj[1];
    } else if (p->id == OCS_OR_PROBLEM_LOGISTIC) {
      const double c = p->par[0], r = p->par[1];
      double s = 0.0;
      for (int i = 0; i < nS; ++i) s = (i == 0) ? -vj[0] : s - vj[i];
      oj[0] = s + 2 * c * exp(-r * t[j]) * uj[0] * vj[nS];
    } else {
      const double r = p->par[0];
      const double *Bu = lq_Bu(p), *R = lq_R(p);
      const double e = exp(-r * t[j]);
      for (int l = 0; l < nC; ++l) {
        double a = 0.0;
        for (int i = 0; i < nS; ++i) a += Bu[i + (size_t)l * nS] * vj[i];
        oj[l] = a + 2 * e * R[l] * uj[l] * vj[nS];
      }
    }
  }
}

/* ---- Gen-2 -> Gen-1 adapter.  H = f + lam.g (make_from_symbolic.m:11),
 * adjointRHS = -grad_x H (:14), dHdu = grad_u H (:17), ControlChar = clamp(argzero dHdu)
 * (:19-23,111).  compute_equilibrium.m:14-20 evaluates the Gen-2 methods with v=[lam;1]
 * for exactly this purpose. */
void ocs_or_stateRHS(const ocs_or_problem *p, int k, const double *t, const double *x, const double *u,
                     double *out) {
  const int nS = p->nS, nAug = p->nAug;
  double *y = (double *)calloc((size_t)nAug * k, sizeof(double));
  double *f = (double *)malloc(sizeof(double) * (size_t)nAug * k);
  for (int j = 0; j < k; ++j) memcpy(y + (size_t)j * nAug, x + (size_t)j * nS, sizeof(double) * nS);
  ocs_or_F(p, k, t, y, u, f);
  for (int j = 0; j < k; ++j) memcpy(out + (size_t)j * nS, f + (size_t)j * nAug, sizeof(double) * nS);
  free(y);
  free(f);
}
void ocs_or_objective(const ocs_or_problem *p, int k, const double *t, const double *x, const double *u,
                      double *out) {
  const int nS = p->nS, nAug = p->nAug;
  double *y = (double *)calloc((size_t)nAug * k, sizeof(double));
  double *f = (double *)malloc(sizeof(double) * (size_t)nAug * k);
  for (int j = 0; j < k; ++j) memcpy(y + (size_t)j * nAug, x + (size_t)j * nS, sizeof(double) * nS);
  ocs_or_F(p, k, t, y, u, f);
  for (int j = 0; j < k; ++j) out[j] = f[(size_t)j * nAug + nS];
  free(y);
  free(f);
}
void ocs_or_adjointRHS(const ocs_or_problem *p, int k, const double *t, const double *x,
                       const double *lam, const double *u, double *out) {
  const int nS = p->nS, nAug = p->nAug;
  double *y = (double *)calloc((size_t)nAug * k, sizeof(double));
  double *v = (double *)calloc((size_t)nAug * k, sizeof(double));
  double *g = (double *)malloc(sizeof(double) * (size_t)nAug * k);
  for (int j = 0; j < k; ++j) {
    memcpy(y + (size_t)j * nAug, x + (size_t)j * nS, sizeof(double) * nS);
    memcpy(v + (size_t)j * nAug, lam + (size_t)j * nS, sizeof(double) * nS);
    v[(size_t)j * nAug + nS] = 1.0;
  }
  ocs_or_dFdx_times_vec(p, k, t, y, u, v, g);
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < nS; ++i) out[(size_t)j * nS + i] = -g[(size_t)j * nAug + i];
  free(y);
  free(v);
  free(g);
}
/* ControlChar(t, x, lam): root of dHdu = dFdu_times_vec(t,[x;0],u,[lam;1]) in u, clamped
 * (make_from_symbolic.m:111  value = min(umax, max(umin, value))).
 *   Test/Logistic: -sum(lam) + 2 c e^{-rt} u = 0  ->  u = sum(lam) e^{rt} / (2c)
 *   LQ:            Bu' lam + 2 e^{-rt} R u = 0    ->  u_c = -(Bu' lam)_c e^{rt} / (2 R_c) */
void ocs_or_ControlChar(const ocs_or_problem *p, int k, const double *t, const double *x,
                        const double *lam, double *out) {
  (void)x;
  const int nS = p->nS, nC = p->nC;
  for (int j = 0; j < k; ++j) {
    const double *lj = lam + (size_t)j * nS;
    double *oj = out + (size_t)j * nC;
    if (p->id == OCS_OR_PROBLEM_TEST || p->id == OCS_OR_PROBLEM_LOGISTIC) {
      const double c = p->par[0];
      const double r = (p->id == OCS_OR_PROBLEM_TEST) ? p->par[2] : p->par[1];
      double s = lj[0];
      for (int i = 1; i < nS; ++i) s += lj[i];
      oj[0] = s * exp(r * t[j]) / (2 * c);
    } else {
      const double r = p->par[0];
      const double *Bu = lq_Bu(p), *R = lq_R(p);
      const double e = exp(r * t[j]);
      for (int l = 0; l < nC; ++l) {
        double a = 0.0;
        for (int i = 0; i < nS; ++i) a += Bu[i + (size_t)l * nS] * lj[i];
        oj[l] = -a * e / (2 * R[l]);
      }
    }
    for (int l = 0; l < nC; ++l) {
      const double lo = p->bounds[l], hi = p->bounds[nC + l];
      oj[l] = fmin(hi, fmax(lo, oj[l]));
    }
  }
}

/* ------------------------------------------------------------------------- */
/* helpers: linspace, griddedInterpolant restatements                        */
/* ------------------------------------------------------------------------- */

/* MATLAB linspace(d1,d2,n): y = d1 + (0:n1).*(d2-d1)./n1 with y(1)=d1, y(end)=d2. */
void ocs_or_linspace(double a, double b, int n, double *out) {
  if (n <= 0) return;
  if (n == 1) {
    out[0] = b;
    return;
  }
  const int n1 = n - 1;
  for (int k = 0; k <= n1; ++k) out[k] = a + ((double)k * (b - a)) / (double)n1;
  out[0] = a;
  out[n1] = b;
}

static int sgn(double v) { return (v > 0) - (v < 0); }

/* Fritsch-Carlson shape-preserving slopes as in MATLAB pchip (Moler, NCM pchiptx):
 * interior: 0 if neighbouring secants differ in sign or vanish, else the weighted
 * harmonic mean  dmin / (w1*del(k)/dmax + w2*del(k+1)/dmax); ends: three-point formula
 * with the two shape-preserving corrections. */
void ocs_or_pchip_slopes(int n, const double *x, const double *y, double *d) {
  if (n < 2) {
    if (n == 1) d[0] = 0.0;
    return;
  }
  double *h = (double *)malloc(sizeof(double) * (size_t)(n - 1));
  double *del = (double *)malloc(sizeof(double) * (size_t)(n - 1));
  for (int i = 0; i < n - 1; ++i) {
    h[i] = x[i + 1] - x[i];
    del[i] = (y[i + 1] - y[i]) / h[i];
  }
  if (n == 2) {
    d[0] = d[1] = del[0];
    free(h);
    free(del);
    return;
  }
  for (int k = 0; k < n - 2; ++k) {
    if (sgn(del[k]) * sgn(del[k + 1]) > 0) {
      const double hs = h[k] + h[k + 1];
      const double w1 = (h[k] + hs) / (3 * hs);
      const double w2 = (hs + h[k + 1]) / (3 * hs);
      const double a0 = fabs(del[k]), a1 = fabs(del[k + 1]);
      const double dmax = fmax(a0, a1), dmin = fmin(a0, a1);
      d[k + 1] = dmin / (w1 * (del[k] / dmax) + w2 * (del[k + 1] / dmax));
    } else {
      d[k + 1] = 0.0;
    }
  }
  d[0] = ((2 * h[0] + h[1]) * del[0] - h[0] * del[1]) / (h[0] + h[1]);
  if (sgn(d[0]) != sgn(del[0]))
    d[0] = 0.0;
  else if (sgn(del[0]) != sgn(del[1]) && fabs(d[0]) > fabs(3 * del[0]))
    d[0] = 3 * del[0];
  d[n - 1] = ((2 * h[n - 2] + h[n - 3]) * del[n - 2] - h[n - 2] * del[n - 3]) / (h[n - 2] + h[n - 3]);
  if (sgn(d[n - 1]) != sgn(del[n - 2]))
    d[n - 1] = 0.0;
  else if (sgn(del[n - 2]) != sgn(del[n - 3]) && fabs(d[n - 1]) > fabs(3 * del[n - 2]))
    d[n - 1] = 3 * del[n - 2];
  free(h);
  free(del);
}

/* interval index k with x[k] <= xq < x[k+1], clamped to [0, n-2] */
static int find_interval(int n, const double *x, double xq) {
  int lo = 0, hi = n - 1;
  if (xq <= x[0]) return 0;
  if (xq >= x[n - 1]) return n - 2;
  while (hi - lo > 1) {
    const int mid = (lo + hi) / 2;
    if (x[mid] <= xq)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

void ocs_or_interp1(int n, const double *x, const double *v, int method, int nq, const double *xq,
                    double *out) {
  double *d = NULL;
  if (method == 3) {
    d = (double *)malloc(sizeof(double) * (size_t)n);
    ocs_or_pchip_slopes(n, x, v, d);
  }
  for (int j = 0; j < nq; ++j) {
    const double q = xq[j];
    if (method == 1 && (q < x[0] || q > x[n - 1])) { /* 'nearest' extrapolation */
      out[j] = (q < x[0]) ? v[0] : v[n - 1];
      continue;
    }
    if (method == 2) { /* 'previous': value at the largest grid point <= q */
      if (q < x[0]) {
        out[j] = NAN; /* 'previous' cannot extrapolate to the left */
        continue;
      }
      int k = n - 1;
      if (q < x[n - 1]) k = find_interval(n, x, q);
      out[j] = v[k];
      continue;
    }
    const int k = find_interval(n, x, q);
    if (method == 0 || method == 1) {
      out[j] = v[k] + (v[k + 1] - v[k]) * ((q - x[k]) / (x[k + 1] - x[k]));
    } else { /* pchip: Hermite cubic in the local variable s = q - x_k (pwch coefficients) */
      const double h = x[k + 1] - x[k];
      const double del = (v[k + 1] - v[k]) / h;
      const double dzzdx = (del - d[k]) / h;
      const double dzdxdx = (d[k + 1] - del) / h;
      const double c3 = (dzdxdx - dzzdx) / h;
      const double c2 = 2 * dzzdx - dzdxdx;
      const double s = q - x[k];
      out[j] = v[k] + s * (d[k] + s * (c2 + s * c3));
    }
  }
  free(d);
}

/* functions/vectorInterpolant.m:1-12: one griddedInterpolant per component row */
void ocs_or_vector_interp(int nComp, int n, const double *x, const double *v, int method, int nq,
                          const double *tq, double *out) {
  double *row = (double *)malloc(sizeof(double) * (size_t)n);
  double *res = (double *)malloc(sizeof(double) * (size_t)nq);
  for (int c = 0; c < nComp; ++c) {
    for (int i = 0; i < n; ++i) row[i] = v[c + (size_t)i * nComp];
    ocs_or_interp1(n, x, row, method, nq, tq, res);
    for (int j = 0; j < nq; ++j) out[c + (size_t)j * nComp] = res[j];
  }
  free(row);
  free(res);
}

/* ------------------------------------------------------------------------- */
/* Integrator/RK4Integrator.m                                                */
/* ------------------------------------------------------------------------- */
struct ocs_or_rk4 {
  int nSTEPS;
  double *t;  /* 1 x (2N+1) */
  double *h;  /* 1 x N */
  int nAug;   /* rows of xK (set by compute_states) */
  double *xK; /* nAug x (N+1) x 4 */
  double *dJdk; /* nAug x N x 4 scratch of compute_adjoints, kept between calls */
  size_t xK_cap, dJdk_cap;
};

/* RK4Integrator.m:16-25 */
ocs_or_rk4 *ocs_or_rk4_create(const double *tspan, int npts) {
  if (npts < 2) return NULL;
  ocs_or_rk4 *g = (ocs_or_rk4 *)calloc(1, sizeof(*g));
  const int N = npts - 1;
  g->nSTEPS = N;
  g->h = (double *)malloc(sizeof(double) * (size_t)N);
  g->t = (double *)malloc(sizeof(double) * (size_t)(2 * N + 1));
  for (int i = 0; i < N; ++i) g->h[i] = tspan[i + 1] - tspan[i];          /* h = diff(tspan)      :17 */
  for (int i = 0; i <= N; ++i) g->t[2 * i] = tspan[i];                    /* t(1:2:end) = tspan   :22 */
  for (int i = 0; i < N; ++i) g->t[2 * i + 1] = (tspan[i] + tspan[i + 1]) / 2; /* midpoints       :23 */
  return g;
}
void ocs_or_rk4_destroy(ocs_or_rk4 *g) {
  if (!g) return;
  free(g->t);
  free(g->h);
  free(g->xK);
  free(g->dJdk);
  free(g);
}
int ocs_or_rk4_nsteps(const ocs_or_rk4 *g) { return g->nSTEPS; }
const double *ocs_or_rk4_t(const ocs_or_rk4 *g) { return g->t; }
const double *ocs_or_rk4_h(const ocs_or_rk4 *g) { return g->h; }
const double *ocs_or_rk4_xK(const ocs_or_rk4 *g) { return g->xK; }

#define XK(g, row, col, slot) ((g)->xK[(row) + (size_t)(g)->nAug * ((col) + (size_t)((g)->nSTEPS + 1) * (slot))])

/* RK4Integrator.m:28-56 */
void ocs_or_rk4_compute_states(ocs_or_rk4 *g, const ocs_or_problem *p, const double *x0,
                               const double *u, double *x, double *J) {
  const int nSTATES = p->nS + 1; /* :29 */
  const int N = g->nSTEPS, nC = p->nC;
  g->nAug = nSTATES;
  const size_t tot = (size_t)nSTATES * (N + 1) * 4;
  if (tot > g->xK_cap) { /* storage is reused between calls; the NaN fill below is the reference's :32 */
    free(g->xK);
    g->xK = (double *)malloc(sizeof(double) * tot);
    g->xK_cap = tot;
  }
  for (size_t i = 0; i < tot; ++i) g->xK[i] = NAN; /* :32 */
  for (int r = 0; r < p->nS; ++r) XK(g, r, 0, 0) = x0[r];
  XK(g, nSTATES - 1, 0, 0) = 0.0; /* :33 */

  double *F1 = (double *)malloc(sizeof(double) * 4 * (size_t)nSTATES);
  double *F2 = F1 + nSTATES, *F3 = F2 + nSTATES, *F4 = F3 + nSTATES;
  for (int i = 0; i < N; ++i) { /* MATLAB i = i+1 */
    const double h = g->h[i];
    const double *tA = g->t + 2 * i, *tM = g->t + 2 * i + 1, *tB = g->t + 2 * i + 2;
    const double *uA = u + (size_t)nC * (2 * i), *uM = uA + nC, *uB = uM + nC;
    ocs_or_F(p, 1, tA, &XK(g, 0, i, 0), uA, F1);                                     /* :39 */
    for (int r = 0; r < nSTATES; ++r) XK(g, r, i, 1) = XK(g, r, i, 0) + h / 2 * F1[r]; /* :40 */
    ocs_or_F(p, 1, tM, &XK(g, 0, i, 1), uM, F2);                                     /* :42 */
    for (int r = 0; r < nSTATES; ++r) XK(g, r, i, 2) = XK(g, r, i, 0) + h / 2 * F2[r]; /* :43 */
    ocs_or_F(p, 1, tM, &XK(g, 0, i, 2), uM, F3);                                     /* :45 */
    for (int r = 0; r < nSTATES; ++r) XK(g, r, i, 3) = XK(g, r, i, 0) + h * F3[r];     /* :46 */
    ocs_or_F(p, 1, tB, &XK(g, 0, i, 3), uB, F4);                                     /* :48 */
    for (int r = 0; r < nSTATES; ++r)                                                /* :50-51 */
      XK(g, r, i + 1, 0) = XK(g, r, i, 0) + h / 6 * (F1[r] + 2 * F2[r] + 2 * F3[r] + F4[r]);
  }
  free(F1);
  if (x)
    for (int c = 0; c <= N; ++c)
      for (int r = 0; r < nSTATES; ++r) x[r + (size_t)c * nSTATES] = XK(g, r, c, 0); /* :54 */
  if (J) *J = XK(g, nSTATES - 1, N, 0);                                              /* :55 */
}

/* RK4Integrator.m:59-94 and compute_dJdu :97-121 */
void ocs_or_rk4_compute_adjoints(ocs_or_rk4 *g, const ocs_or_problem *p, const double *u,
                                 const double *lamT, double *lam, double *dJdu) {
  const int nSTATES = g->nAug; /* size(obj.xK,1) :61 */
  const int N = g->nSTEPS, nC = p->nC;
  const size_t LD = (size_t)nSTATES;
#define LAM(r, c) lam[(r) + LD * (c)]
#define DJDK(r, c, s) dJdk[(r) + LD * ((c) + (size_t)N * (s))]
  for (int r = 0; r < nSTATES; ++r)
    LAM(r, N) = lamT ? lamT[r] : (r == nSTATES - 1 ? 1.0 : 0.0); /* :63-69 */
  if (LD * N * 4 > g->dJdk_cap) {
    free(g->dJdk);
    g->dJdk = (double *)malloc(sizeof(double) * LD * N * 4);
    g->dJdk_cap = LD * N * 4;
  }
  double *dJdk = g->dJdk; /* :70 */
  double *dJdx1 = (double *)malloc(sizeof(double) * 4 * LD);
  double *dJdx2 = dJdx1 + LD, *dJdx3 = dJdx2 + LD, *dJdx0 = dJdx3 + LD;

  for (int i = N - 1; i >= 0; --i) { /* MATLAB i = i+1 */
    const double h = g->h[i];
    const double *tA = g->t + 2 * i, *tM = tA + 1, *tB = tA + 2;
    const double *uA = u + (size_t)nC * (2 * i), *uM = uA + nC, *uB = uM + nC;
    for (int r = 0; r < nSTATES; ++r) DJDK(r, i, 3) = h / 6 * LAM(r, i + 1);           /* :73 */
    ocs_or_dFdx_times_vec(p, 1, tB, &XK(g, 0, i, 3), uB, &DJDK(0, i, 3), dJdx3);       /* :74-75 */
    for (int r = 0; r < nSTATES; ++r) DJDK(r, i, 2) = h / 3 * LAM(r, i + 1) + h * dJdx3[r]; /* :77 */
    ocs_or_dFdx_times_vec(p, 1, tM, &XK(g, 0, i, 2), uM, &DJDK(0, i, 2), dJdx2);       /* :78-79 */
    for (int r = 0; r < nSTATES; ++r) DJDK(r, i, 1) = h / 3 * LAM(r, i + 1) + h / 2 * dJdx2[r]; /* :81 */
    ocs_or_dFdx_times_vec(p, 1, tM, &XK(g, 0, i, 1), uM, &DJDK(0, i, 1), dJdx1);       /* :82-83 */
    for (int r = 0; r < nSTATES; ++r) DJDK(r, i, 0) = h / 6 * LAM(r, i + 1) + h / 2 * dJdx1[r]; /* :85 */
    ocs_or_dFdx_times_vec(p, 1, tA, &XK(g, 0, i, 0), uA, &DJDK(0, i, 0), dJdx0);       /* :87-88 */
    for (int r = 0; r < nSTATES; ++r)
      LAM(r, i) = LAM(r, i + 1) + dJdx1[r] + dJdx2[r] + dJdx3[r] + dJdx0[r];            /* :86-88 */
  }

  if (dJdu) { /* nargout > 1  :91-93 -> compute_dJdu :97-121 */
    const int nT = 2 * N + 1;
    double *tmpA = (double *)malloc(sizeof(double) * (size_t)nC * N * 2);
    double *tmpB = tmpA + (size_t)nC * N;
    double *tt = (double *)malloc(sizeof(double) * (size_t)N);
    double *uu = (double *)malloc(sizeof(double) * (size_t)nC * N);
    for (size_t i = 0; i < (size_t)nC * nT; ++i) dJdu[i] = 0.0; /* :98 */
    /* Left end point :101-102 */
    ocs_or_dFdu_times_vec(p, 1, g->t, &XK(g, 0, 0, 0), u, &DJDK(0, 0, 0), dJdu);
    /* RK step interval mid points :105-109: columns 2:2:end-1 (0-based 1,3,..,2N-1) */
    for (int i = 0; i < N; ++i) {
      tt[i] = g->t[2 * i + 1];
      memcpy(uu + (size_t)nC * i, u + (size_t)nC * (2 * i + 1), sizeof(double) * nC);
    }
    ocs_or_dFdu_times_vec(p, N, tt, &XK(g, 0, 0, 1), uu, &DJDK(0, 0, 1), tmpA);
    ocs_or_dFdu_times_vec(p, N, tt, &XK(g, 0, 0, 2), uu, &DJDK(0, 0, 2), tmpB);
    for (int i = 0; i < N; ++i)
      for (int c = 0; c < nC; ++c)
        dJdu[c + (size_t)nC * (2 * i + 1)] = tmpA[c + (size_t)nC * i] + tmpB[c + (size_t)nC * i];
    /* interior end points :112-116: columns 3:2:end-2 (0-based 2,4,..,2N-2), N-1 of them */
    if (N > 1) {
      for (int i = 0; i < N - 1; ++i) {
        tt[i] = g->t[2 * i + 2];
        memcpy(uu + (size_t)nC * i, u + (size_t)nC * (2 * i + 2), sizeof(double) * nC);
      }
      ocs_or_dFdu_times_vec(p, N - 1, tt, &XK(g, 0, 1, 0), uu, &DJDK(0, 1, 0), tmpA); /* xK(:,2:end-1,1), dJdk(:,2:end,1) */
      ocs_or_dFdu_times_vec(p, N - 1, tt, &XK(g, 0, 0, 3), uu, &DJDK(0, 0, 3), tmpB); /* xK(:,1:end-2,4), dJdk(:,1:end-1,4) */
      for (int i = 0; i < N - 1; ++i)
        for (int c = 0; c < nC; ++c)
          dJdu[c + (size_t)nC * (2 * i + 2)] = tmpA[c + (size_t)nC * i] + tmpB[c + (size_t)nC * i];
    }
    /* Right end point :119-120 */
    ocs_or_dFdu_times_vec(p, 1, g->t + 2 * N, &XK(g, 0, N - 1, 3), u + (size_t)nC * (2 * N),
                          &DJDK(0, N - 1, 3), dJdu + (size_t)nC * (2 * N));
    free(tmpA);
    free(tt);
    free(uu);
  }
  free(dJdx1);
#undef LAM
#undef DJDK
}

/* ------------------------------------------------------------------------- */
/* Integrator/RK4InfiniteIntegrator.m                                        */
/* ------------------------------------------------------------------------- */
struct ocs_or_rk4inf {
  ocs_or_rk4 *integrator1, *integrator2;
  double *uStar; /* nC x (2*N2+1) */
  int nC;
};
/* :12-17 */
ocs_or_rk4inf *ocs_or_rk4inf_create(const double *tspan, int npts, const double *tspanExtra,
                                    int nptsExtra, const double *uStar, int nC) {
  ocs_or_rk4inf *g = (ocs_or_rk4inf *)calloc(1, sizeof(*g));
  g->integrator1 = ocs_or_rk4_create(tspan, npts);
  g->integrator2 = ocs_or_rk4_create(tspanExtra, nptsExtra);
  g->nC = nC;
  const int nT2 = 2 * g->integrator2->nSTEPS + 1;
  g->uStar = (double *)malloc(sizeof(double) * (size_t)nC * nT2);
  for (int j = 0; j < nT2; ++j)
    for (int c = 0; c < nC; ++c) g->uStar[c + (size_t)nC * j] = uStar[c] * 1.0; /* uStar*ones(size(t)) :15 */
  return g;
}
void ocs_or_rk4inf_destroy(ocs_or_rk4inf *g) {
  if (!g) return;
  ocs_or_rk4_destroy(g->integrator1);
  ocs_or_rk4_destroy(g->integrator2);
  free(g->uStar);
  free(g);
}
const double *ocs_or_rk4inf_t(const ocs_or_rk4inf *g) { return g->integrator1->t; } /* :16 */
int ocs_or_rk4inf_nsteps(const ocs_or_rk4inf *g) { return g->integrator1->nSTEPS; }
/* :20-24 */
void ocs_or_rk4inf_compute_states(ocs_or_rk4inf *g, const ocs_or_problem *p, const double *x0,
                                  const double *u, double *x, double *J) {
  const int nAug = p->nS + 1, N1 = g->integrator1->nSTEPS;
  double J1, J2;
  double *xloc = x ? x : (double *)malloc(sizeof(double) * (size_t)nAug * (N1 + 1));
  ocs_or_rk4_compute_states(g->integrator1, p, x0, u, xloc, &J1);
  ocs_or_rk4_compute_states(g->integrator2, p, xloc + (size_t)nAug * N1, g->uStar, NULL, &J2); /* x(1:end-1,end) */
  if (J) *J = J1 + J2;
  if (!x) free(xloc);
}
/* :27-30 */
void ocs_or_rk4inf_compute_adjoints(ocs_or_rk4inf *g, const ocs_or_problem *p, const double *u,
                                    double *lam, double *dJdu) {
  const int nAug = p->nS + 1, N2 = g->integrator2->nSTEPS;
  double *lam2 = (double *)malloc(sizeof(double) * (size_t)nAug * (N2 + 1));
  ocs_or_rk4_compute_adjoints(g->integrator2, p, g->uStar, NULL, lam2, NULL);
  ocs_or_rk4_compute_adjoints(g->integrator1, p, u, lam2 /* lam2(:,1) */, lam, dJdu);
  free(lam2);
}

/* ------------------------------------------------------------------------- */
/* Control classes                                                           */
/* ------------------------------------------------------------------------- */
struct ocs_or_control {
  int kind, nBasis, nControls, nt;
  double *pts; /* controlPts (PWLinear, Chebyshev: nBasis) or intervalStarts (PWConstant: nBasis) */
  double *B;   /* nBasis x nt */
  double t0, t1;
};

ocs_or_control *ocs_or_control_create(int kind, const double *t, int nt, int nBasis, int nControls) {
  ocs_or_control *c = (ocs_or_control *)calloc(1, sizeof(*c));
  c->kind = kind;
  c->nBasis = nBasis;
  c->nControls = nControls;
  c->nt = nt;
  c->t0 = t[0];
  c->t1 = t[nt - 1];
  c->B = (double *)calloc((size_t)nBasis * nt, sizeof(double));
#define BM(i, j) c->B[(i) + (size_t)nBasis * (j)]
  if (kind == OCS_OR_CONTROL_PWLINEAR) {
    /* PWLinearControl.m:16 controlPts = linspace(t(1), t(end), nControlPts); :31-50 tents */
    c->pts = (double *)malloc(sizeof(double) * (size_t)nBasis);
    ocs_or_linspace(t[0], t[nt - 1], nBasis, c->pts);
    double *row = (double *)malloc(sizeof(double) * (size_t)nt);
    const double v10[2] = {1, 0}, v010[3] = {0, 1, 0}, v01[2] = {0, 1};
    ocs_or_interp1(2, c->pts, v10, 1, nt, t, row); /* :35-37 */
    for (int j = 0; j < nt; ++j) BM(0, j) = row[j];
    for (int i = 1; i < nBasis - 1; ++i) { /* :40-44 */
      ocs_or_interp1(3, c->pts + i - 1, v010, 1, nt, t, row);
      for (int j = 0; j < nt; ++j) BM(i, j) = row[j];
    }
    ocs_or_interp1(2, c->pts + nBasis - 2, v01, 1, nt, t, row); /* :47-49 */
    for (int j = 0; j < nt; ++j) BM(nBasis - 1, j) = row[j];
    free(row);
  } else if (kind == OCS_OR_CONTROL_PWCONSTANT) {
    /* PWConstantControl.m:14-15 intervalStarts = linspace(t(1),t(end),n+1)(1:end-1); :41-50 */
    double *ls = (double *)malloc(sizeof(double) * (size_t)(nBasis + 1));
    ocs_or_linspace(t[0], t[nt - 1], nBasis + 1, ls);
    c->pts = (double *)malloc(sizeof(double) * (size_t)nBasis);
    memcpy(c->pts, ls, sizeof(double) * (size_t)nBasis);
    free(ls);
    for (int i = 0; i < nBasis - 1; ++i)
      for (int j = 0; j < nt; ++j) BM(i, j) = (t[j] >= c->pts[i] && t[j] < c->pts[i + 1]) ? 1.0 : 0.0;
    for (int j = 0; j < nt; ++j) BM(nBasis - 1, j) = (t[j] >= c->pts[nBasis - 1]) ? 1.0 : 0.0;
  } else {
    /* ChebyshevControl.m:16 controlPts (unused by the reference); :21-31 recurrence */
    c->pts = (double *)malloc(sizeof(double) * (size_t)nBasis);
    ocs_or_linspace(t[0], t[nt - 1], nBasis, c->pts);
    for (int j = 0; j < nt; ++j) {
      const double tT = 2 * (t[j] - t[0]) / (t[nt - 1] - t[0]) - 1; /* :23 */
      BM(0, j) = 1.0;
      if (nBasis > 1) BM(1, j) = tT;
      for (int i = 2; i < nBasis; ++i) BM(i, j) = 2 * tT * BM(i - 1, j) - BM(i - 2, j); /* :28-30 */
    }
  }
#undef BM
  return c;
}
void ocs_or_control_destroy(ocs_or_control *c) {
  if (!c) return;
  free(c->pts);
  free(c->B);
  free(c);
}
int ocs_or_control_nbasis(const ocs_or_control *c) { return c->nBasis; }
const double *ocs_or_control_B(const ocs_or_control *c) { return c->B; }
const double *ocs_or_control_pts(const ocs_or_control *c) { return c->pts; }

/* u = reshape(v, nControls, []) * B   (PWLinearControl.m:59-62 and twins): dense product */
void ocs_or_control_compute_u(const ocs_or_control *c, const double *v, double *u) {
  const int nC = c->nControls, nB = c->nBasis;
  for (int j = 0; j < c->nt; ++j)
    for (int r = 0; r < nC; ++r) {
      double a = 0.0;
      for (int i = 0; i < nB; ++i) a += v[r + (size_t)nC * i] * c->B[i + (size_t)nB * j];
      u[r + (size_t)nC * j] = a;
    }
}
/* dJdv = reshape(dJdu * B', [], 1)    (PWLinearControl.m:53-56 and twins) */
void ocs_or_control_compute_dJdv(const ocs_or_control *c, const double *dJdu, double *dJdv) {
  const int nC = c->nControls, nB = c->nBasis;
  for (int i = 0; i < nB; ++i)
    for (int r = 0; r < nC; ++r) {
      double a = 0.0;
      for (int j = 0; j < c->nt; ++j) a += dJdu[r + (size_t)nC * j] * c->B[i + (size_t)nB * j];
      dJdv[r + (size_t)nC * i] = a;
    }
}
/* compute_initial_v: PWLinearControl.m:65-71 (length(u0)==1 -> repmat; ==nControlPts -> reshape;
 * build stance SURVEY App. A: nC x 1 u0 is repmat'ed as well), PWConstantControl.m:53-55,
 * ChebyshevControl.m:46-48.  Returns 0 on success, -1 when the reference would leave v unset. */
int ocs_or_control_compute_initial_v(const ocs_or_control *c, const double *u0, int len_u0, double *v) {
  const int nC = c->nControls, nB = c->nBasis;
  if (c->kind == OCS_OR_CONTROL_CHEBYSHEV) {
    if (len_u0 != nC) return -1;
    for (int i = 0; i < nC * nB; ++i) v[i] = 0.0;
    for (int r = 0; r < nC; ++r) v[r] = u0[r];
    return 0;
  }
  if (len_u0 == nC) {
    for (int i = 0; i < nB; ++i)
      for (int r = 0; r < nC; ++r) v[r + (size_t)nC * i] = u0[r];
    return 0;
  }
  if (c->kind == OCS_OR_CONTROL_PWLINEAR && len_u0 == nC * nB) {
    memcpy(v, u0, sizeof(double) * (size_t)len_u0);
    return 0;
  }
  return -1;
}
/* compute_nlp_bounds: PWLinearControl.m:21-28, PWConstantControl.m:20-27 */
void ocs_or_control_compute_nlp_bounds(const ocs_or_control *c, const double *bounds, double *Lb,
                                       double *Ub) {
  const int nC = c->nControls, nB = c->nBasis;
  for (int i = 0; i < nB; ++i)
    for (int r = 0; r < nC; ++r) {
      Lb[r + (size_t)nC * i] = bounds[r] * 1.0;
      Ub[r + (size_t)nC * i] = bounds[nC + r] * 1.0;
    }
}
void ocs_or_control_eval_uFunc(const ocs_or_control *c, const double *v, int nq, const double *tq,
                               double *out) {
  const int nC = c->nControls, nB = c->nBasis;
  if (c->kind == OCS_OR_CONTROL_PWLINEAR) {
    ocs_or_vector_interp(nC, nB, c->pts, v, 0, nq, tq, out); /* PWLinearControl.m:74-77 'linear' */
  } else if (c->kind == OCS_OR_CONTROL_PWCONSTANT) {
    ocs_or_vector_interp(nC, nB, c->pts, v, 2, nq, tq, out); /* PWConstantControl.m:58-61 'previous' */
  } else { /* Chebyshev: the reference defines no compute_uFunc; evaluate sum_k v_k T_k(tau) */
    for (int j = 0; j < nq; ++j) {
      const double tT = 2 * (tq[j] - c->t0) / (c->t1 - c->t0) - 1;
      for (int r = 0; r < nC; ++r) {
        double b0 = 1.0, b1 = tT, a = v[r] * 1.0;
        if (nB > 1) a += v[r + (size_t)nC] * tT;
        for (int i = 2; i < nB; ++i) {
          const double b2 = 2 * tT * b1 - b0;
          a += v[r + (size_t)nC * i] * b2;
          b0 = b1;
          b1 = b2;
        }
        out[r + (size_t)nC * j] = a;
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* functions/single_shooting.m:137-150 nlpObjective                          */
/* ------------------------------------------------------------------------- */
void ocs_or_nlp_objective(int integ_kind, void *gv, const ocs_or_problem *p, const ocs_or_control *c,
                          double *x0, const double *v, int nFree, const int *FreeInitStates, double *J,
                          double *dJdv) {
  const int nC = p->nC, nAug = p->nS + 1;
  const int nV = c->nControls * c->nBasis;
  const int N = integ_kind ? ocs_or_rk4inf_nsteps((ocs_or_rk4inf *)gv) : ocs_or_rk4_nsteps((ocs_or_rk4 *)gv);
  double *u = (double *)malloc(sizeof(double) * (size_t)nC * (2 * N + 1));
  double *dJdu = (double *)malloc(sizeof(double) * (size_t)nC * (2 * N + 1));
  double *lam = (double *)malloc(sizeof(double) * (size_t)nAug * (N + 1));
  ocs_or_control_compute_u(c, v, u);                                         /* :139 / :145 */
  for (int f = 0; f < nFree; ++f) x0[FreeInitStates[f] - 1] = v[nV + f];     /* :146 */
  if (integ_kind) {
    ocs_or_rk4inf_compute_states((ocs_or_rk4inf *)gv, p, x0, u, NULL, J);
    ocs_or_rk4inf_compute_adjoints((ocs_or_rk4inf *)gv, p, u, lam, dJdu);
  } else {
    ocs_or_rk4_compute_states((ocs_or_rk4 *)gv, p, x0, u, NULL, J);          /* :140 / :147 */
    ocs_or_rk4_compute_adjoints((ocs_or_rk4 *)gv, p, u, NULL, lam, dJdu);    /* :141 / :148 */
  }
  ocs_or_control_compute_dJdv(c, dJdu, dJdv);                                /* :142 / :149 */
  for (int f = 0; f < nFree; ++f) dJdv[nV + f] = lam[FreeInitStates[f] - 1]; /* lam(FreeInitStates,1) :149 */
  free(u);
  free(dJdu);
  free(lam);
}

/* ------------------------------------------------------------------------- */
/* functions/compute_x_lam(_J).m and fb_sweep.m on the grid                  */
/* ------------------------------------------------------------------------- */
void ocs_or_fbs_default_options(ocs_or_fbs_options *o) {
  o->uRelTol = 1e-7;    /* fb_sweep.m:16 */
  o->uAbsTol = 1e-7;    /* :17 */
  o->nSWEEPS = 50;      /* :20 */
  o->nERROR_PTS = 1001; /* :21 */
  o->nINTERP_PTS = 1001; /* :22 */
  o->uRelax = 0.0;
}

/* compute_x_lam.m:1-19 / compute_x_lam_J.m:1-21 with odevr7 -> grid RK4.
 * forward  (compute_x_lam_J.m:6-15): [x;J]' = [stateRHS ; objective](t, x, u(t)), [x0;0]
 * backward (compute_x_lam.m:11-14):  lam' = adjointRHS(t, x(t), lam, u(t)), lam(TF) = 0*x0,
 *          x(t) = pchip of the forward node values (compute_x_lam.m:9). */
void ocs_or_compute_x_lam(const ocs_or_rk4 *g, const ocs_or_problem *p, const double *x0,
                          const double *ugrid, double *x, double *lam, double *J) {
  const int nS = p->nS, nC = p->nC, N = g->nSTEPS;
  double *y = (double *)malloc(sizeof(double) * (size_t)(nS + 1) * 6);
  double *Y = y + (nS + 1), *k1 = Y + (nS + 1), *k2 = k1 + (nS + 1), *k3 = k2 + (nS + 1), *k4 = k3 + (nS + 1);
  /* forward: classical RK4 on the augmented system [x ; Jacc] */
  for (int r = 0; r < nS; ++r) y[r] = x0[r];
  y[nS] = 0.0;
  for (int r = 0; r < nS; ++r) x[r] = y[r];
  for (int i = 0; i < N; ++i) {
    const double h = g->h[i];
    const double *tA = g->t + 2 * i, *tM = tA + 1, *tB = tA + 2;
    const double *uA = ugrid + (size_t)nC * (2 * i), *uM = uA + nC, *uB = uM + nC;
    ocs_or_stateRHS(p, 1, tA, y, uA, k1);
    ocs_or_objective(p, 1, tA, y, uA, k1 + nS);
    for (int r = 0; r < nS; ++r) Y[r] = y[r] + h / 2 * k1[r];
    ocs_or_stateRHS(p, 1, tM, Y, uM, k2);
    ocs_or_objective(p, 1, tM, Y, uM, k2 + nS);
    for (int r = 0; r < nS; ++r) Y[r] = y[r] + h / 2 * k2[r];
    ocs_or_stateRHS(p, 1, tM, Y, uM, k3);
    ocs_or_objective(p, 1, tM, Y, uM, k3 + nS);
    for (int r = 0; r < nS; ++r) Y[r] = y[r] + h * k3[r];
    ocs_or_stateRHS(p, 1, tB, Y, uB, k4);
    ocs_or_objective(p, 1, tB, Y, uB, k4 + nS);
    for (int r = 0; r <= nS; ++r) y[r] = y[r] + h / 6 * (k1[r] + 2 * k2[r] + 2 * k3[r] + k4[r]);
    for (int r = 0; r < nS; ++r) x[r + (size_t)nS * (i + 1)] = y[r];
  }
  if (J) *J = y[nS]; /* compute_x_lam_J.m:15 J = xAugout(end,end) */

  /* x(t) at the interval midpoints: pchip of the node values */
  double *tn = (double *)malloc(sizeof(double) * (size_t)(N + 1));
  double *tm = (double *)malloc(sizeof(double) * (size_t)N);
  double *xm = (double *)malloc(sizeof(double) * (size_t)nS * N);
  for (int i = 0; i <= N; ++i) tn[i] = g->t[2 * i];
  for (int i = 0; i < N; ++i) tm[i] = g->t[2 * i + 1];
  ocs_or_vector_interp(nS, N + 1, tn, x, 3, N, tm, xm);

  /* backward from TF to T0 with step -h, lam0 = 0*x0 (compute_x_lam.m:4,12) */
  double *l = y, *L = Y;
  for (int r = 0; r < nS; ++r) l[r] = 0.0 * x0[r];
  for (int r = 0; r < nS; ++r) lam[r + (size_t)nS * N] = l[r];
  for (int i = N - 1; i >= 0; --i) {
    const double h = -g->h[i];
    const double *tA = g->t + 2 * i, *tM = tA + 1, *tB = tA + 2;
    const double *uA = ugrid + (size_t)nC * (2 * i), *uM = uA + nC, *uB = uM + nC;
    const double *xA = x + (size_t)nS * i, *xB = x + (size_t)nS * (i + 1), *xM = xm + (size_t)nS * i;
    ocs_or_adjointRHS(p, 1, tB, xB, l, uB, k1);
    for (int r = 0; r < nS; ++r) L[r] = l[r] + h / 2 * k1[r];
    ocs_or_adjointRHS(p, 1, tM, xM, L, uM, k2);
    for (int r = 0; r < nS; ++r) L[r] = l[r] + h / 2 * k2[r];
    ocs_or_adjointRHS(p, 1, tM, xM, L, uM, k3);
    for (int r = 0; r < nS; ++r) L[r] = l[r] + h * k3[r];
    ocs_or_adjointRHS(p, 1, tA, xA, L, uA, k4);
    for (int r = 0; r < nS; ++r) l[r] = l[r] + h / 6 * (k1[r] + 2 * k2[r] + 2 * k3[r] + k4[r]);
    for (int r = 0; r < nS; ++r) lam[r + (size_t)nS * i] = l[r];
  }
  free(y);
  free(tn);
  free(tm);
  free(xm);
}

/* uNew = @(t) prob.ControlChar(t, x(t), lam(t))   fb_sweep.m:96, x/lam = pchip of node values */
void ocs_or_control_from_x_lam(const ocs_or_rk4 *g, const ocs_or_problem *p, const double *x,
                               const double *lam, int nq, const double *tq, double *out) {
  const int nS = p->nS, N = g->nSTEPS;
  double *tn = (double *)malloc(sizeof(double) * (size_t)(N + 1));
  double *xq = (double *)malloc(sizeof(double) * (size_t)nS * nq);
  double *lq = (double *)malloc(sizeof(double) * (size_t)nS * nq);
  for (int i = 0; i <= N; ++i) tn[i] = g->t[2 * i];
  ocs_or_vector_interp(nS, N + 1, tn, x, 3, nq, tq, xq);
  ocs_or_vector_interp(nS, N + 1, tn, lam, 3, nq, tq, lq);
  ocs_or_ControlChar(p, nq, tq, xq, lq, out);
  free(tn);
  free(xq);
  free(lq);
}

/* fb_sweep.m:79-125 */
int ocs_or_fb_sweep(const ocs_or_rk4 *g, const ocs_or_problem *p, const double *x0,
                    const ocs_or_fbs_options *o, const double *u0grid, const double *u0err, double *x,
                    double *lam, double *uInterp, double *J, double *maxChange) {
  const int nC = p->nC, N = g->nSTEPS, nT = 2 * N + 1;
  const double T0 = g->t[0], TF = g->t[2 * N];
  double *errorPts = (double *)malloc(sizeof(double) * (size_t)o->nERROR_PTS);
  double *interpPts = (double *)malloc(sizeof(double) * (size_t)o->nINTERP_PTS);
  ocs_or_linspace(T0, TF, o->nERROR_PTS, errorPts);   /* :69 */
  ocs_or_linspace(T0, TF, o->nINTERP_PTS, interpPts); /* :70 */
  double *u = (double *)malloc(sizeof(double) * (size_t)nC * nT);
  double *uNew = (double *)malloc(sizeof(double) * (size_t)nC * nT);
  double *uErr = (double *)malloc(sizeof(double) * (size_t)nC * o->nERROR_PTS);
  double *uErrNew = (double *)malloc(sizeof(double) * (size_t)nC * o->nERROR_PTS);
  memcpy(u, u0grid, sizeof(double) * (size_t)nC * nT); /* u = u0 :76 */
  memcpy(uErr, u0err, sizeof(double) * (size_t)nC * o->nERROR_PTS);
  int converged_at = 0;
  for (int sweepIdx = 1; sweepIdx <= o->nSWEEPS; ++sweepIdx) { /* :79 */
    /* uNew = sweep(u) :80, :94-97 */
    ocs_or_compute_x_lam(g, p, x0, u, x, lam, NULL);
    ocs_or_control_from_x_lam(g, p, x, lam, nT, g->t, uNew);
    ocs_or_control_from_x_lam(g, p, x, lam, o->nERROR_PTS, errorPts, uErrNew);
    /* check_convergence(uNew, u) :99-115 */
    double mx = NAN; /* MATLAB max() skips NaN and returns NaN only when every entry is NaN */
    for (size_t i = 0; i < (size_t)nC * o->nERROR_PTS; ++i) {
      const double w = fabs(uErrNew[i] - uErr[i]) / (o->uRelTol * fabs(uErr[i]) + o->uAbsTol); /* :107 */
      if (w == w && (mx != mx || w > mx)) mx = w;                                                 /* :108 */
    }
    if (maxChange) maxChange[sweepIdx - 1] = mx;
    if (mx <= 1) { /* :110 */
      /* final_sweep(u) with the OLD u :82, :117-125 */
      ocs_or_compute_x_lam(g, p, x0, u, x, lam, J);
      ocs_or_control_from_x_lam(g, p, x, lam, o->nINTERP_PTS, interpPts, uInterp); /* :123 */
      converged_at = sweepIdx;
      break;
    }
    if (o->uRelax > 0.0 && o->uRelax < 1.0) { /* damped update (extension): u = u + w (uNew - u) */
      for (size_t i = 0; i < (size_t)nC * nT; ++i) u[i] = fma(o->uRelax, uNew[i] - u[i], u[i]);
      for (size_t i = 0; i < (size_t)nC * o->nERROR_PTS; ++i) uErr[i] = fma(o->uRelax, uErrNew[i] - uErr[i], uErr[i]);
    } else {
      memcpy(u, uNew, sizeof(double) * (size_t)nC * nT); /* u = uNew :85 */
      memcpy(uErr, uErrNew, sizeof(double) * (size_t)nC * o->nERROR_PTS);
    }
  }
  free(errorPts);
  free(interpPts);
  free(u);
  free(uNew);
  free(uErr);
  free(uErrNew);
  return converged_at;
}

/* ------------------------------------------------------------------------- */
/* batch driver for the cpu_baseline timing                                   */
/* ------------------------------------------------------------------------- */
int ocs_or_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void ocs_or_batch_states_adjoints(int id, int nS, int nC, const double *params, int nparams,
                                  const double *bounds, const double *tspan, int npts, int batch,
                                  const double *x0, const double *u, double *x, double *J, double *lam,
                                  double *dJdu, int nthreads) {
  const int N = npts - 1, nAug = nS + 1;
  const size_t su = (size_t)nC * (2 * N + 1), sx = (size_t)nAug * (N + 1);
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    ocs_or_problem *p = ocs_or_problem_create(id, nS, nC, params, nparams, bounds);
    ocs_or_rk4 *g = ocs_or_rk4_create(tspan, npts);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int b = 0; b < batch; ++b) {
      ocs_or_rk4_compute_states(g, p, x0 + (size_t)nS * b, u + su * b, x ? x + sx * b : NULL,
                                J ? J + b : NULL);
      double *lb = lam ? lam + sx * b : (double *)malloc(sizeof(double) * sx);
      ocs_or_rk4_compute_adjoints(g, p, u + su * b, NULL, lb, dJdu ? dJdu + su * b : NULL);
      if (!lam) free(lb);
    }
    ocs_or_rk4_destroy(g);
    ocs_or_problem_destroy(p);
  }
  (void)nthreads;
}
