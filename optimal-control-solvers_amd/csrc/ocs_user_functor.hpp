// ocs_user_functor.hpp -- wraps user-supplied plugin functions into the functor concept the kernels
// are instantiated on.  This header is compiled ONLY by hipRTC (ocs_jit.cpp), after the user's source.
//
// The user's translation unit defines, in plain device C++ (the three methods of OCProblem/OCProblem.m:8-21,
// same argument order as the MATLAB methods, t a scalar, one column at a time):
//
//   __device__ void ocs_F(double t, const double* y, const double* u, OCS_PARAMS p, double* f);
//        f[0..NS-1] = state right-hand side, f[NS] = objective integrand; y[NS] is not available (never read)
//   __device__ void ocs_dFdx_times_vec(double t, const double* y, const double* u, OCS_PARAMS p,
//                                      const double* v, double* g);      g[0..NS-1] = (dF/dy)' v, v has NS+1 entries
//   __device__ void ocs_dFdu_times_vec(double t, const double* y, const double* u, OCS_PARAMS p,
//                                      const double* v, double* g);      g[0..NC-1] = (dF/du)' v
//   (optional, for fb_sweep; announce with has_control_char)
//   __device__ void ocs_ControlChar(double t, const double* x, const double* lam, OCS_PARAMS p,
//                                   const double* lb, const double* ub, double* u);
//
//   (optional: #define OCS_USER_TCOEF in the source and give
//   __device__ double ocs_tcoef(double t, OCS_PARAMS p);
//    -- the three methods then receive ocs_tcoef(t, p) in the place of t, evaluated once per grid point into the integrator's
//    step records instead of once per call: return exp(-r t) for a discounted objective whose dynamics do not read the time,
//    which is what keeps the transcendental off the per-stage path (the registry problems hoist tests/TestOCProblem.m:25,31,37
//    the same way).  The parameters it reads must be the same for all trajectories.)
//   (optional: #define OCS_USER_CC_TCOEF in the source and give
//   __device__ double ocs_cc_tcoef(double t, OCS_PARAMS p);
//    -- ocs_ControlChar then receives ocs_cc_tcoef(t, p) in the place of t, evaluated once per grid point into the
//    integrator's tables instead of once per call: e.g. return exp(r t) for the current-value costate of a discounted
//    problem, as the registry problems tabulate it.  The parameters it reads must be the same for all trajectories.)
//
// NS, NC, NPAR are available as constants.  OCS_PARAMS is `const double*` (a per-trajectory register copy,
// so per-trajectory parameter overrides work) when NPAR <= 16, and a pointer to the shared parameter block
// in constant address space (scalar loads) otherwise, and always for problems given as row functions.
#pragma once

// ---- row-separable user problems (OCS_USER_ROWSEP) --------------------------------------------------------------
// A problem whose state rows are uncoupled -- row r of F reads y_r, u and t only, and the objective integrand is a sum
// of per-row shares -- may be given as ROW FUNCTIONS instead of the three full-vector methods:
//   __device__ double ocs_row_tcoef(double t, OCS_PARAMS p);                              time coefficient tc(t)
//   __device__ double ocs_row_F(double tc, double y, double u, OCS_PARAMS p, int r);      F_r(t, y_r, u)
//   __device__ double ocs_row_q(double tc, double y, double u, OCS_PARAMS p, int r);      share of F(end): sum_r = integrand
//   __device__ void   ocs_row_dFdy(double tc, double y, double u, OCS_PARAMS p, int r, double* dF, double* dq);
//                                                                  dF_r/dy_r and dq_r/dy_r
//   __device__ void   ocs_row_dFdu(double tc, double y, double u, OCS_PARAMS p, int r, double* dF, double* dq);
//                                                                  dF_r/du and dq_r/du
// Time enters the row functions through ONE coefficient tc = ocs_row_tcoef(t, p), evaluated once per grid point into
// the step records (return t itself if the rows need the time; return e.g. exp(-r t) for a discounted objective, which
// keeps the transcendental off the per-stage path as tests/TestOCProblem.m:25,31,37 are hoisted for the registry
// problems); the parameters it reads must be the same for all trajectories.  ocs_ControlChar still receives t.
// (NC = 1.)  The full-vector plugin methods of OCProblem.m:8-21 are derived from them below, so every kernel works; in
// addition the wave-specialised state pass and the scan adjoint pass (the mappings of the registry problems) are
// instantiated for the problem.
//
// OCS_USER_CC_NOX (flag bit 2 of ocs_problem_create_from_source, with row functions and ocs_ControlChar): the problem
// declares that ocs_ControlChar does not read x and that ocs_row_dFdy does not read u -- the minimum principle gives the
// control from the costate alone and the costate equation does not see the control, as for a Hamiltonian that is
// separable in (x, u).  fb_sweep then runs its two-kernel sweep (ocs_fold_kernel.hpp: the state pass forms its control
// from the costate of the sweep before; ocs_costate_scan_kernel.hpp: the costate pass as a scan over time with the
// convergence test inside).  ocs_ControlChar receives x = zeros and ocs_row_dFdy u = 0 there.
#ifdef OCS_USER_ROWSEP
__device__ static inline void ocs_F(double t, const double* y, const double* u, OCS_PARAMS p, double* f) {
  double s = 0.0;
  for (int k = 0; k < OCS_USER_NS; ++k) {
    f[k] = ocs_row_F(t, y[k], u[0], p, k);
    s += ocs_row_q(t, y[k], u[0], p, k);
  }
  f[OCS_USER_NS] = s;
}
__device__ static inline void ocs_dFdx_times_vec(double t, const double* y, const double* u, OCS_PARAMS p,
                                                 const double* v, double* g) {
  for (int k = 0; k < OCS_USER_NS; ++k) {
    double dF, dq;
    ocs_row_dFdy(t, y[k], u[0], p, k, &dF, &dq);
    g[k] = __builtin_fma(dF, v[k], dq * v[OCS_USER_NS]);
  }
}
__device__ static inline void ocs_dFdu_times_vec(double t, const double* y, const double* u, OCS_PARAMS p,
                                                 const double* v, double* g) {
  double s = 0.0;
  for (int k = 0; k < OCS_USER_NS; ++k) {
    double dF, dq;
    ocs_row_dFdu(t, y[k], u[0], p, k, &dF, &dq);
    s += __builtin_fma(dF, v[k], dq * v[OCS_USER_NS]);
  }
  g[0] = s;
}
#endif

namespace ocs {

struct UserP {
#if defined(OCS_USER_CC_NOX) && defined(OCS_USER_ROWSEP) && defined(OCS_USER_HAS_CONTROLCHAR)
  static constexpr bool CC_READS_X = false;  // declared by the problem
#else
  static constexpr bool CC_READS_X = true;   // unknown: ocs_ControlChar may read x
#endif
  static constexpr int NS = OCS_USER_NS;
  static constexpr int NC = OCS_USER_NC;
  static constexpr int NAUG = OCS_USER_NS + 1;
  static constexpr int NPAR = OCS_USER_NPAR;
  static constexpr int NTC = 1;   // tc[0] = t: user code receives the time itself
  static constexpr int NTU = 1;
  static constexpr int NSC = 0;
  static constexpr unsigned TC_PARAM_MASK = 0u;

#if OCS_USER_NPAR <= 16 && !defined(OCS_USER_ROWSEP)
  struct Par {
    double p[OCS_USER_NPAR > 0 ? OCS_USER_NPAR : 1];
  };
  __device__ static inline Par load(const ParamSrc& g) {
    Par q;
#pragma unroll
    for (int k = 0; k < NPAR; ++k) q.p[k] = g(k);
    return q;
  }
  __device__ static inline const double* par(const Par& q) { return q.p; }
#else
  struct Par {
    uniform_ptr p;
  };
  __device__ static inline Par load(const ParamSrc& g) { return Par{g.ps}; }
  __device__ static inline uniform_ptr par(const Par& q) { return q.p; }
#endif

  __device__ static inline void tcoef(double t, const double* ps, double* tc, double* tu) {
#if defined(OCS_USER_ROWSEP)
    tc[0] = ocs_row_tcoef(t, (OCS_PARAMS)ps);   // the user's time coefficient (shared parameters)
#elif defined(OCS_USER_TCOEF)
    tc[0] = ocs_tcoef(t, (OCS_PARAMS)ps);       // ... of a problem given as full-vector methods
#else
    (void)ps;
    tc[0] = t;
#endif
#ifdef OCS_USER_CC_TCOEF
    tu[0] = ocs_cc_tcoef(t, (OCS_PARAMS)ps);    // the user's ControlChar-side time coefficient (shared parameters)
#else
    tu[0] = t;
#endif
  }
  __device__ static inline void step_consts(double, double, const double*, const double*, const double*, double*) {}

  __device__ static inline void F(const double* tc, const double* y, const double* u, const Par& p, double* f) {
    ocs_F(tc[0], y, u, par(p), f);
  }
  __device__ static inline void Fx(const double* tc, const double* y, const double* u, const Par& p, double* f) {
    double full[NAUG];
    ocs_F(tc[0], y, u, par(p), full);
#pragma unroll
    for (int k = 0; k < NS; ++k) f[k] = full[k];
  }
  __device__ static inline void dFdxT(const double* tc, const double* y, const double* u, const Par& p,
                                      const double* v, double* g) {
    ocs_dFdx_times_vec(tc[0], y, u, par(p), v, g);
  }
  __device__ static inline void dFduT(const double* tc, const double* y, const double* u, const Par& p,
                                      const double* v, double* g) {
    ocs_dFdu_times_vec(tc[0], y, u, par(p), v, g);
  }
#ifdef OCS_USER_ROWSEP
  // ---- generic row-separable interface (ocs_problems.hpp) over the user's row functions ----
  static constexpr bool ROW_SEPARABLE = true;
  static constexpr bool DFDU_READS_Y = true;    // unknown: assume (dF/du)'v may read y
  static constexpr bool HAS_SHIFT = false;
  struct RowPar {
    Par p;
    int r;
  };
  // (row functions index the parameters by the row number, a per-lane value: a register copy of the block would
  //  live in scratch memory.  They read the shared block in constant address space instead -- invariant loads the
  //  compiler hoists out of the time loop -- so per-trajectory parameter overrides do not apply to these problems.)
  __device__ static inline RowPar load_row(const ParamSrc& g, int r) { return RowPar{Par{g.ps}, r}; }
  struct Stage { double t, kc; };   // time of the stage's grid point, cost-row entry of dJdk (h/6 or h/3 times lam(end))
  template <bool LT>
  __device__ static inline Stage stage(double sc, double w, double tc, double lamc) {
    (void)sc;
    return Stage{tc, LT ? w * lamc : w};
  }
  __device__ static inline double g_row_f(double y, double u, double tc, const RowPar& rp) {
    return ocs_row_F(tc, y, u, par(rp.p), rp.r);
  }
  __device__ static inline double g_row_q(double y, double u, double tc, const RowPar& rp) {
    return ocs_row_q(tc, y, u, par(rp.p), rp.r);
  }
  __device__ static inline void g_row_dfdx_pre(double y, double u, const Stage& st, const RowPar& rp, double& a, double& b) {
    double dF, dq;
    ocs_row_dFdy(st.t, y, u, par(rp.p), rp.r, &dF, &dq);
    a = dF;
    b = dq * st.kc;
  }
  __device__ static inline double g_row_dfdx(double y, double u, double v, const Stage& st, const RowPar& rp) {
    double dF, dq;
    ocs_row_dFdy(st.t, y, u, par(rp.p), rp.r, &dF, &dq);
    return __builtin_fma(dF, v, dq * st.kc);
  }
  __device__ static inline double g_row_dfdu(double y, double u, double v, const Stage& st, const RowPar& rp) {
    double dF, dq;
    ocs_row_dFdu(st.t, y, u, par(rp.p), rp.r, &dF, &dq);
    return __builtin_fma(dF, v, dq * st.kc);
  }
  // the costate equation's row: lam_r' = -(a lam_r + b), a = dF_r/dy_r, b = dq_r/dy_r (the cost row of [lam; 1]);
  // without u only for problems that declare OCS_USER_CC_NOX (u is not read)
  __device__ static inline void costate_row_pre(double y, double tc, const RowPar& rp, double& a, double& b) {
    ocs_row_dFdy(tc, y, 0.0, par(rp.p), rp.r, &a, &b);
  }
  __device__ static inline void costate_row_pre_u(double y, double u, double tc, const RowPar& rp, double& a, double& b) {
    ocs_row_dFdy(tc, y, u, par(rp.p), rp.r, &a, &b);
  }
  // (names of the shifted form of the registry problems: never called, HAS_SHIFT is false)
  __device__ static inline double row_shift(const RowPar&) { return 0.0; }
  __device__ static inline double row_vertex(double, double u) { return u; }
  __device__ static inline double row_f_shifted(double z, double) { return z; }
  __device__ static inline double control_q(double, const RowPar&) { return 0.0; }
  __device__ static inline double state_q_acc(double, double acc) { return acc; }
#endif

#if defined(OCS_USER_ROWSEP)
  // ControlChar from the costate alone (the names LogisticK gives the same thing, ocs_problems.hpp); only called by the
  // kernels of the two-kernel sweep, which are instantiated for problems that declare OCS_USER_CC_NOX
  struct CCPre {
    Par p;
  };
  __device__ static inline CCPre cc_pre(const Par& p) { return CCPre{p}; }
  __device__ static inline double control_char_pre(double tu, const double* lam, const CCPre& c, double lb, double ub) {
    double x0[NS], u;
#pragma unroll
    for (int k = 0; k < NS; ++k) x0[k] = 0.0;
#ifdef OCS_USER_HAS_CONTROLCHAR
    ocs_ControlChar(tu, x0, lam, par(c.p), &lb, &ub, &u);
#else
    (void)tu; (void)lam; (void)c; (void)ub;
    u = lb;
#endif
    return u;
  }
#endif

  __device__ static inline void control_char(const double* tu, const double* x, const double* lam, const Par& p,
                                             const double* lb, const double* ub, double* u) {
#ifdef OCS_USER_HAS_CONTROLCHAR
    ocs_ControlChar(tu[0], x, lam, par(p), lb, ub, u);
#else
#pragma unroll
    for (int c = 0; c < NC; ++c) u[c] = lb[c];
#endif
  }
};

}  // namespace ocs
