// ocs_problems.hpp -- device functors for the OCProblem plugin surface.
//
// The reference's plugin is three MATLAB methods (OCProblem/OCProblem.m:8-21):
//   F(t,y,u)                 nAug x 1   (last row = objective integrand; y(end) is never read)
//   dFdx_times_vec(t,y,u,v)  (dF/dy)' v (last row identically 0, OCProblem.m:14-15)
//   dFdu_times_vec(t,y,u,v)  (dF/du)' v
// A kernel cannot call back into MATLAB, so each registered problem is a struct of
// __device__ functions that the RK4 kernels are instantiated on.  Everything that
// depends on t only (exp(-r t) in tests/TestOCProblem.m:25,31,37) is hoisted into a
// per-grid-point table TC filled once per (integrator, problem) pair by tcoef(): it is
// wave-uniform and reaches the kernels through the per-step record table, which keeps
// transcendental calls off the serial RK4 recursion.
#pragma once
#include <hip/hip_runtime.h>

namespace ocs {

// LogisticK<NS>: x_k' = x_k (m_k - x_k) - u,  cost' = e^{-rt} (sum_k x_k^2 + c u^2).
// NS = 1 with params [c r m] is tests/TestOCProblem.m (params [c m r] are permuted on the host).
//   parameter block: [c, r, m_1 .. m_NS]
template <int NS_>
struct LogisticK {
  static constexpr int NS = NS_;
  static constexpr int NC = 1;
  static constexpr int NAUG = NS_ + 1;
  static constexpr int NPAR = 2 + NS_;
  static constexpr int NTC = 1;                  // F-side time coefficients  [e^{-rt}]
  static constexpr int NTU = 1;                  // ControlChar-side coefficients [e^{+rt}]
  static constexpr unsigned TC_PARAM_MASK = 2u;  // r (index 1) feeds tcoef: must stay batch-uniform

  struct Par {
    double c;
    double m[NS];
  };

  template <class Get>
  __device__ static inline Par load(Get get) {
    Par p;
    p.c = get(0);
#pragma unroll
    for (int k = 0; k < NS; ++k) p.m[k] = get(2 + k);
    return p;
  }

  __device__ static inline void tcoef(double t, const double* ps, double* tc, double* tu) {
    const double r = ps[1];
    tc[0] = exp(-r * t);
    tu[0] = exp(r * t);
  }

  // tests/TestOCProblem.m:22-26.  f has NAUG entries; y has NS entries (y(end) is never read).
  __device__ static inline void F(const double* tc, const double* y, const double* u, const Par& p,
                                  double* f) {
    double s = y[0] * y[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) s = __builtin_fma(y[k], y[k], s);
#pragma unroll
    for (int k = 0; k < NS; ++k) f[k] = __builtin_fma(y[k], p.m[k] - y[k], -u[0]);
    f[NS] = tc[0] * __builtin_fma(p.c, u[0] * u[0], s);
  }
  // states only (stage recomputation in the adjoint pass does not need the cost row)
  __device__ static inline void Fx(const double* tc, const double* y, const double* u, const Par& p,
                                   double* f) {
    (void)tc;
#pragma unroll
    for (int k = 0; k < NS; ++k) f[k] = __builtin_fma(y[k], p.m[k] - y[k], -u[0]);
  }
  // tests/TestOCProblem.m:29-33.  v has NAUG entries, g gets the NS non-trivial rows.
  __device__ static inline void dFdxT(const double* tc, const double* y, const double* u, const Par& p,
                                      const double* v, double* g) {
    (void)u;
    const double ev = 2.0 * tc[0] * v[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k)
      g[k] = __builtin_fma(__builtin_fma(-2.0, y[k], p.m[k]), v[k], ev * y[k]);
  }
  // tests/TestOCProblem.m:36-38.  g gets NC entries.
  __device__ static inline void dFduT(const double* tc, const double* y, const double* u, const Par& p,
                                      const double* v, double* g) {
    (void)y;
    double s = -v[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) s -= v[k];
    g[0] = __builtin_fma(2.0 * p.c * tc[0] * u[0], v[NS], s);
  }
  // Per-step constants that depend on the grid only (uniform over the batch); computed once into the
  // record table so that the wave-specialised kernels need not recompute them per lane and step.
  static constexpr int NSC = 8;
  __device__ static inline void step_consts(double h6, double h3, const double* tcA, const double* tcM,
                                            const double* tcB, double* sc) {
    sc[0] = (2.0 * tcB[0]) * h6;  // E4:  2 e^{-r t} * (h/6): cost-row factor of dFdx_times_vec at stage 4
    sc[1] = (2.0 * tcM[0]) * h3;  // E3 (= E2)
    sc[2] = (2.0 * tcA[0]) * h6;  // E1
    sc[3] = h6 * tcA[0];          // W_A: RK4 quadrature weights of the objective integrand
    sc[4] = 2.0 * (h6 * tcM[0]);  // W_M
    sc[5] = h6 * tcB[0];          // W_B
    sc[6] = 2.0 * tcA[0];         // cost-row factor of the continuous adjoint right-hand side at the left node
    sc[7] = 2.0 * tcM[0];         //   ... and at the midpoint (costate pass of the sweep; the right node is the next step's left)
  }

  // ---- row-separable form (row-split mapping, ocs_rowsplit_kernels.hip) --------------------
  // Row k of F depends on y_k and u only, and the objective integrand is a sum of per-row terms
  // q_k(y_k, u) (the control cost c u^2 is charged to row 0), so a group of NS lanes can own one
  // trajectory, one state row per lane, and meet only in the running-cost / dJdu reductions.
  static constexpr bool ROW_SEPARABLE = true;
  struct RowPar {
    double m;   // m_r
    double cw;  // c for row 0, 0 for the other rows
  };
  template <class Get>
  __device__ static inline RowPar load_row(Get get, int r) {
    RowPar rp;
    rp.m = get(2);
#pragma unroll
    for (int k = 1; k < NS; ++k) rp.m = (r == k) ? get(2 + k) : rp.m;
    rp.cw = (r == 0) ? get(0) : 0.0;
    return rp;
  }
  // row r of F (states):  y (m_r - y) - u
  __device__ static inline double row_f(double y, double u, const RowPar& rp) {
    return __builtin_fma(y, rp.m - y, -u);
  }
  // The same row about its vertex: with z = y - m_r/2,  F_r = (m_r^2/4 - u) - z^2, and y + a F becomes z + a F.
  // A marching wave that carries z instead of y has two dependent operations per RK4 stage instead of three
  // (the constant m_r^2/4 - u does not depend on the state); results differ from row_f by round-off only.
  __device__ static inline double row_shift(const RowPar& rp) { return 0.5 * rp.m; }
  __device__ static inline double row_vertex(double mh, double u) { return __builtin_fma(mh, mh, -u); }
  __device__ static inline double row_f_shifted(double z, double cv) { return __builtin_fma(-z, z, cv); }
  // this row's share of the objective integrand, without the e^{-rt} factor:  y^2 + cw u^2
  __device__ static inline double row_q(double y, double u2, const RowPar& rp) {
    return __builtin_fma(rp.cw, u2, y * y);
  }
  // the same integrand for a lane that holds all rows of one stage state: control_q(u^2) + sum_r state_q(y_r)
  // (equal to sum_r row_q(y_r, u^2, rp_r): the control cost is charged to row 0 only)
  __device__ static inline double control_q(double u2, const RowPar& rp0) { return rp0.cw * u2; }
  __device__ static inline double state_q_acc(double y, double acc) { return __builtin_fma(y, y, acc); }
  // row r of (dF/dy)' v:  (m_r - 2 y) v_r + ev y,   ev = 2 e^{-rt} v_cost
  __device__ static inline double row_dfdx(double y, double v, double ev, const RowPar& rp) {
    return __builtin_fma(__builtin_fma(-2.0, y, rp.m), v, ev * y);
  }
  // the same row in affine form g = a v_r + b (a, b do not depend on v_r, so a wave that runs ahead of
  // the adjoint recursion can prepare them):  a = m_r - 2 y,  b = ev y
  __device__ static inline void row_dfdx_pre(double y, double ev, const RowPar& rp, double& a, double& b) {
    a = __builtin_fma(-2.0, y, rp.m);
    b = ev * y;
  }
  // the costate equation's row (compute_x_lam.m:11-14, lam_r' = -(a lam_r + b)) from the time coefficient e^{-r t}
  __device__ static inline void costate_row_pre(double y, double tc, const RowPar& rp, double& a, double& b) {
    row_dfdx_pre(y, 2.0 * tc, rp, a, b);
  }
  __device__ static inline void costate_row_pre_u(double y, double u, double tc, const RowPar& rp, double& a, double& b) {
    (void)u;
    row_dfdx_pre(y, 2.0 * tc, rp, a, b);
  }
  // this row's share of (dF/du)' v:  -v_r + cw u ev,   cu = cw u
  __device__ static inline double row_dfdu(double cu, double v, double ev) {
    return __builtin_fma(cu, ev, -v);
  }

  // ---- generic row-separable interface (the names the scan / pipeline kernels call; a user problem given as row
  //      functions implements the same names in ocs_user_functor.hpp) ------------------------------------------------
  // Stage: what a stage evaluation of the adjoint needs besides y, u: built from the step record (sc = the problem's
  // step constant of that stage, w = h/6 or h/3, tc = the time coefficient of the stage's grid point) and lam(end).
  static constexpr bool DFDU_READS_Y = false;   // (dF/du)'v of these rows does not depend on y
  static constexpr bool HAS_SHIFT = true;       // the state recursion may run on z = y - row_shift (row_f_shifted)
  struct Stage { double ev; };
  template <bool LT>
  __device__ static inline Stage stage(double sc, double w, double tc, double lamc) {
    (void)w; (void)tc;
    return Stage{LT ? sc * lamc : sc};
  }
  __device__ static inline double g_row_f(double y, double u, double tc, const RowPar& rp) { (void)tc; return row_f(y, u, rp); }
  // this row's share of the objective integrand at a stage (time coefficient included)
  __device__ static inline double g_row_q(double y, double u, double tc, const RowPar& rp) { return tc * row_q(y, u * u, rp); }
  __device__ static inline void g_row_dfdx_pre(double y, double u, const Stage& st, const RowPar& rp, double& a, double& b) {
    (void)u;
    row_dfdx_pre(y, st.ev, rp, a, b);
  }
  __device__ static inline double g_row_dfdx(double y, double u, double v, const Stage& st, const RowPar& rp) {
    (void)u;
    return row_dfdx(y, v, st.ev, rp);
  }
  __device__ static inline double g_row_dfdu(double y, double u, double v, const Stage& st, const RowPar& rp) {
    (void)y;
    return row_dfdu(rp.cw * u, v, st.ev);
  }

  // Gen-1 ControlChar through the A9 adapter (make_from_symbolic.m:19-23,111):
  //   dHdu = -sum(lam) + 2 c e^{-rt} u = 0  ->  u = sum(lam) e^{rt} / (2c), clamped to the bounds.
  __device__ static inline void control_char(const double* tu, const double* x, const double* lam,
                                             const Par& p, const double* lb, const double* ub, double* u) {
    (void)x;
    double s = lam[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) s += lam[k];
    const double v = s * tu[0] / (2.0 * p.c);
    u[0] = fmin(ub[0], fmax(lb[0], v));
  }
  // the same with the division hoisted out of a loop over grid points (kernels that evaluate ControlChar per step on a
  // wave with an instruction budget: k_forward_cc, k_costate_plx): one rounding more than control_char
  static constexpr bool CC_READS_X = false;
  struct CCPre {
    double inv2c;
  };
  __device__ static inline CCPre cc_pre(const Par& p) { return CCPre{1.0 / (2.0 * p.c)}; }
  __device__ static inline double control_char_pre(double tu, const double* lam, const CCPre& c, double lb, double ub) {
    double s = lam[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) s += lam[k];
    return fmin(ub, fmax(lb, s * tu * c.inv2c));
  }
};

}  // namespace ocs
