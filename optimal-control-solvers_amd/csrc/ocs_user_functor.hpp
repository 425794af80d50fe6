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
// NS, NC, NPAR are available as constants.  OCS_PARAMS is `const double*` (a per-trajectory register copy,
// so per-trajectory parameter overrides work) when NPAR <= 16, and a pointer to the shared parameter block
// in constant address space (scalar loads) otherwise.
#pragma once

namespace ocs {

struct UserP {
  static constexpr int NS = OCS_USER_NS;
  static constexpr int NC = OCS_USER_NC;
  static constexpr int NAUG = OCS_USER_NS + 1;
  static constexpr int NPAR = OCS_USER_NPAR;
  static constexpr int NTC = 1;   // tc[0] = t: user code receives the time itself
  static constexpr int NTU = 1;
  static constexpr int NSC = 0;
  static constexpr unsigned TC_PARAM_MASK = 0u;

#if OCS_USER_NPAR <= 16
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

  __device__ static inline void tcoef(double t, const double*, double* tc, double* tu) {
    tc[0] = t;
    tu[0] = t;
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
