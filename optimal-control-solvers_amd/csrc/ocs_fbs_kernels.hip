// ocs_fbs_kernels.hip -- forward-backward sweep on the integrator grid (functions/fb_sweep.m,
// compute_x_lam.m, compute_x_lam_J.m) through the Gen-2 -> Gen-1 adapter.
//
// The reference integrates with the external adaptive solver odevr7 and couples the passes by
// pchip interpolants of the solver output; here the solver is classical RK4 on the node grid
// (the forward pass is the k_forward kernel), x(t) / lam(t) are pchip interpolants of the node
// values exactly as compute_x_lam.m:9,14, and
//   adjointRHS  = -dFdx_times_vec(t, [x;0], u, [lam;1])(1:nS)       (make_from_symbolic.m:14)
//   ControlChar = clamp(argzero_u dFdu_times_vec(t, [x;0], u, [lam;1]))   (:19-23, :111)
// One sweep = forward RK4 (serial in t) -> pchip midpoints of x (parallel in t) -> costate RK4
// (serial) -> pchip midpoints of lam -> new control on the 2N+1 grid and on the error points
// (parallel) -> weighted max-norm change per instance and convergence bookkeeping.
#include "ocs_device_common.hpp"
#include "ocs_fbs_device.hpp"
#include "ocs_internal.hpp"
#include "ocs_jit.hpp"
#include "ocs_problems.hpp"

namespace ocs {

static inline int hip_rc3(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
#define OCS_DISPATCH_LOGISTIC2(NSV, CALL) \
  switch (NSV) {                          \
    case 1: { using P = LogisticK<1>; CALL; } break; \
    case 2: { using P = LogisticK<2>; CALL; } break; \
    case 3: { using P = LogisticK<3>; CALL; } break; \
    case 4: { using P = LogisticK<4>; CALL; } break; \
    default: return -1;                   \
  }

static PchipTab make_tab(const FbsTables& t) { return PchipTab{t.n, t.TN, t.HN, t.W1, t.W2, t.IH}; }

int launch_pchip_mid(const FbsTables& t, int nrows, int ld, int batch, const double* V, double* out, hipStream_t s,
                     int ldb, const int* gate) {
  k_pchip_mid<<<dim3((batch + 255) / 256, (t.n - 1 + kPchipRun - 1) / kPchipRun), dim3(256), 0, s>>>(
      make_tab(t), nrows, ld, batch, t.TM, V, out, ldb, gate);
  return hip_rc3(hipGetLastError());
}

// whether launch_costate takes xmid == NULL (with PR) and forms the pchip midpoints of x itself
bool costate_forms_midpoints(const ProblemDesc& p, int N, int batch) {
  if (p.functor == Functor::User) {
    if (user_vector(p.user)) {   // full-vector methods, nS <= 4, nC <= 2: dense step maps
      // at every batch: unlike the adjoint pass of the integrator (ocs_kernels.hip) the sweep's costate scan also spares the
      // midpoint kernel -- two states, us per sweep with / without it: batch 8192 479 / 627, 32768 1168 / 1276, 65536 2206 / 2338
      // (profiles/r04_user_vector_pair_by_batch_mapping.log; OCS_COSTATE_VSCAN_MAX sets a limit for A/B runs)
      static const int vmax = [] { const char* e = getenv("OCS_COSTATE_VSCAN_MAX"); return e ? atoi(e) : 0; }();
      return N >= 8 && N % 8 == 0 && (vmax <= 0 || batch <= vmax);
    }
    // problems given as row functions: the scan that reads the control samples
    return user_rowsep(p.user) && p.nC == 1 && (p.nS == 1 || p.nS == 2 || p.nS == 4) && N >= 8 && N % 8 == 0 &&
           tile_ok(batch, 64 / p.nS);
  }
  return costate_pl_ok(p.functor, p.nS, p.nC, N, batch) && batch / (64 / p.nS) <= fold_wg_limit();
}
template <class P>
static void run_costate(const CostateArgs& a, hipStream_t s) {
  k_costate<P, 4><<<dim3((a.batch + 63) / 64), dim3(64), 0, s>>>(a);
}
int launch_costate(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* xmid,
                   const double* u, const int* frozen, double* dump, double* lam, hipStream_t s, int ldb,
                   const double* PR, const int* gate) {
  if (frozen && !dump) return -1;
  if (p.functor == Functor::User && !xmid) {
    if (ldb != 0 && ldb != batch) return -1;
    return user_vector(p.user) ? launch_costate_vscan(p, g, batch, x, ldx, PR, u, frozen, lam, s, gate)
                               : launch_costate_scan_u(p, g, batch, x, ldx, PR, u, frozen, lam, s, gate);
  }
  // the wave-specialised kernel while its workgroups (one per 64/nS instances) fit on the chip in two rounds
  // (launch_costate_pl has no gate: a gated pass with the midpoints given takes the lane kernel below)
  if (p.functor != Functor::User && costate_forms_midpoints(p, g.N, batch) && !(xmid && gate))
    return xmid ? launch_costate_pl(p, g, batch, x, ldx, xmid, frozen, dump, lam, ldb, s)
                : launch_costate_plx(p, g, batch, x, ldx, PR, frozen, dump, lam, ldb, s, gate);
  if (!xmid) return -1;
  CostateArgs a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, x, ldx, xmid, u, frozen, dump, lam, ldb};
  a.gate = gate;
  if (p.functor == Functor::User) {
    void* args[] = {(void*)&a};
    return jit_launch(p.user, UK_COSTATE, dim3((batch + 63) / 64), dim3(64), args, s);
  }
  if (p.functor != Functor::Logistic) return -1;
  OCS_DISPATCH_LOGISTIC2(p.nS, run_costate<P>(a, s));
  return hip_rc3(hipGetLastError());
}

template <class P>
static void run_control_grid(const ControlGridArgs& a, hipStream_t s) {
  k_control_grid<P><<<dim3((a.batch + 255) / 256, (a.N + kPchipRun - 1) / kPchipRun), dim3(256), 0, s>>>(a);
}
int control_grid_parts(int N) { return (N + kPchipRun - 1) / kPchipRun; }
int launch_control_grid(const ProblemDesc& p, const GridDesc& g, const FbsTables& t, int batch, const double* x, int ldx,
                        const double* xmid, const double* lam, double* u, const int* status, double* metric,
                        double relTol, double absTol, hipStream_t s, int ldb, const int* gate, double relax) {
  const ControlGridArgs a{g.N, batch, g.TU, p.ps, p.pb, p.pmask, p.lb, p.ub, x, ldx, xmid, lam, make_tab(t), t.TM, u,
                          status, metric, relTol, absTol, ldb, gate, relax};
  if (p.functor == Functor::User) {
    void* args[] = {(void*)&a};
    if (p.nS > 4 && !xmid) return -1;   // (no costate kernel of these shapes leaves the midpoints of x to this one: OWNX is not instantiated)
    return jit_launch(p.user, UK_CONTROL_GRID, dim3((batch + 255) / 256, (g.N + kPchipRun - 1) / kPchipRun), dim3(256), args, s);
  }
  if (p.functor != Functor::Logistic) return -1;
  OCS_DISPATCH_LOGISTIC2(p.nS, run_control_grid<P>(a, s));
  return hip_rc3(hipGetLastError());
}

template <class P>
static void run_control_pts(const ControlPtsArgs& a, hipStream_t s) {
  k_control_pts<P><<<dim3((a.batch + 255) / 256, (a.nq + kPtsPerThread - 1) / kPtsPerThread), dim3(256), 0, s>>>(a);
}
int launch_control_pts(const ProblemDesc& p, const FbsTables& t, int nq, const int* KQ, const double* SQ,
                       const double* TUQ, int batch, const double* x, int ldx, const double* lam, double* out,
                       const int* usel, long long odelta, double* metric, int* anyvalid, double relTol,
                       double absTol, hipStream_t s, double relax, const int* gate) {
  ControlPtsArgs a{nq, batch, make_tab(t), KQ, SQ, TUQ, p.ps, p.pb, p.pmask, p.lb, p.ub, x, ldx, lam, out,
                   usel, odelta, metric, anyvalid, relTol, absTol, relax};
  a.gate = gate;
  if (p.functor == Functor::User) {
    void* args[] = {(void*)&a};
    return jit_launch(p.user, UK_CONTROL_PTS, dim3((batch + 255) / 256, (nq + kPtsPerThread - 1) / kPtsPerThread),
                      dim3(256), args, s);
  }
  if (p.functor != Functor::Logistic) return -1;
  OCS_DISPATCH_LOGISTIC2(p.nS, run_control_pts<P>(a, s));
  return hip_rc3(hipGetLastError());
}

template <class P>
static void run_tu_at(int nq, const double* tq, const double* ps, double* TUQ, hipStream_t s) {
  k_tu_at<P><<<dim3((nq + 255) / 256), dim3(256), 0, s>>>(nq, tq, ps, TUQ);
}
int launch_tu_at(const ProblemDesc& p, int nq, const double* tq, double* TUQ, hipStream_t s) {
  if (p.functor == Functor::User) {
    const double* ps = p.ps;
    void* args[] = {&nq, &tq, &ps, &TUQ};
    return jit_launch(p.user, UK_TU_AT, dim3((nq + 255) / 256), dim3(256), args, s);
  }
  if (p.functor != Functor::Logistic) return -1;
  OCS_DISPATCH_LOGISTIC2(p.nS, run_tu_at<P>(nq, tq, p.ps, TUQ, s));
  return hip_rc3(hipGetLastError());
}

int launch_interp(int method, const FbsTables& t, int nComp, int nq, const int* KQ, const double* SQ, int batch,
                  const double* V, double* out, hipStream_t s) {
  k_interp<<<dim3((batch + 255) / 256, (nq + kInterpPts - 1) / kInterpPts, nComp), dim3(256), 0, s>>>(
      method, make_tab(t), nComp, nq, batch, KQ, SQ, V, out);
  return hip_rc3(hipGetLastError());
}

int launch_interp_pchip_sorted(const FbsTables& t, int nComp, const int* QS, const int* QI, const double* SS, int batch,
                               const double* V, double* out, hipStream_t s) {
  k_interp_pchip_sorted<<<dim3((batch + 255) / 256, (t.n - 1 + kInterpRun - 1) / kInterpRun, nComp), dim3(256), 0, s>>>(
      make_tab(t), nComp, batch, QS, QI, SS, V, out);
  return hip_rc3(hipGetLastError());
}

// start of a solve: usel = 0, status = 0, maxChange = NaN (all-ones) in one launch instead of three memsets
__global__ void k_fbs_init(int batch, long long nmc, int* __restrict__ usel, int* __restrict__ status,
                           unsigned long long* __restrict__ mc) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch) {
    usel[i] = 0;
    status[i] = 0;
  }
  if (i < nmc) mc[i] = ~0ull;
}
int launch_fbs_init(int batch, int nsweeps, int* usel, int* status, double* maxChange, hipStream_t s) {
  const long long nmc = (long long)nsweeps * batch;
  k_fbs_init<<<dim3((unsigned)((nmc + 255) / 256)), dim3(256), 0, s>>>(batch, nmc, usel, status, (unsigned long long*)maxChange);
  return hip_rc3(hipGetLastError());
}

int control_pts_parts(int nq) { return (nq + kPtsPerThread - 1) / kPtsPerThread; }
// the error-point mode on points sorted by interval (k_control_pts_sorted): registry problems
int control_pts_run_parts(int N) { return (N + kCtlRun - 1) / kCtlRun; }
bool control_pts_sorted_ok(const ProblemDesc& p) {
  return (p.functor == Functor::Logistic || p.functor == Functor::User) && p.nS >= 1 && p.nS <= 4;
}
template <class P>
static void run_control_pts_sorted(const ControlPtsArgs& a, const int* QS, hipStream_t s) {
  k_control_pts_sorted<P><<<dim3((a.batch + 255) / 256, (a.T.n - 1 + kCtlRun - 1) / kCtlRun), dim3(256), 0, s>>>(a, QS);
}
int launch_control_pts_sorted(const ProblemDesc& p, const FbsTables& t, int nq, const int* QS, const double* SQ,
                              const double* TUQ, int batch, const double* x, int ldx, const double* lam, double* out,
                              double* metric, double relTol, double absTol, hipStream_t s, double relax, const int* gate) {
  if (!control_pts_sorted_ok(p)) return -1;
  ControlPtsArgs a{nq, batch, make_tab(t), nullptr, SQ, TUQ, p.ps, p.pb, p.pmask, p.lb, p.ub, x, ldx, lam, out,
                   nullptr, 0, metric, nullptr, relTol, absTol, relax};
  a.gate = gate;
  if (p.functor == Functor::User) {
    void* args[] = {(void*)&a, (void*)&QS};
    return jit_launch(p.user, UK_CONTROL_PTS_SORTED, dim3((batch + 255) / 256, (t.n - 1 + kCtlRun - 1) / kCtlRun), dim3(256), args, s);
  }
  OCS_DISPATCH_LOGISTIC2(p.nS, run_control_pts_sorted<P>(a, QS, s));
  return hip_rc3(hipGetLastError());
}
int launch_fbs_advance(int batch, int sweep, int nparts, const double* metric, int* anyvalid, int* usel, int* status,
                       double* maxChange, int* nactive, hipStream_t s, int ldb, const int* gate) {
  k_fbs_advance<<<dim3((batch + 63) / 64), dim3(64), 0, s>>>(batch, sweep, nparts, metric, anyvalid, usel, status,
                                                                 maxChange, nactive, ldb, gate);
  return hip_rc3(hipGetLastError());
}

}  // namespace ocs
