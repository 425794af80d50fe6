// ocs_scan_kernels.hip -- launchers of the scan adjoint kernel (ocs_scan_kernel.hpp) for the registry problems, its
// record table, and the dispatch to the hipRTC instances of user problems given as row functions.
#include "ocs_scan_kernel.hpp"
#include "ocs_vscan_kernel.hpp"
#include "ocs_costate_scan_kernel.hpp"
#include "ocs_internal.hpp"
#include "ocs_jit.hpp"
#include "ocs_problems.hpp"
#include <cstdlib>

namespace ocs {

static inline int hip_rc6(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// RECS points at the record of step 0
__global__ void k_build_recs(int N, int RS, int SCO, const double* __restrict__ REC, double* __restrict__ RECS) {
  const int ip = blockIdx.x * blockDim.x + threadIdx.x - kScanPadFront;
  if (ip >= N + kScanPadBack) return;
  double* o = RECS + (long long)ip * kScanRec;
  const bool in = ip >= 0 && ip < N;
  const double* r = REC + (long long)(in ? ip : 0) * RS;
  for (int k = 0; k < kScanRec; ++k) o[k] = 0.0;
  if (!in) return;
  for (int k = 0; k < 4; ++k) o[k] = r[k];
  for (int k = 0; k < 3; ++k) o[4 + k] = r[SCO + k];   // the problem's step constants of stages 4, 3 (= 2), 1
  o[8] = r[4];                                          // time coefficients at t_2i, t_2i+1, t_2i+2 (NTC = 1)
  o[9] = r[5];
  o[10] = r[6];
}
size_t scan_recs_doubles(int N) { return (size_t)(N + kScanPadFront + kScanPadBack) * kScanRec; }
size_t scan_recs_front() { return (size_t)kScanPadFront * kScanRec; }
int launch_build_recs(int N, int rs, int sco, const double* REC, double* RECS_base, hipStream_t s) {
  const int n = N + kScanPadFront + kScanPadBack;
  k_build_recs<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(N, rs, sco, REC, RECS_base + (size_t)kScanPadFront * kScanRec);
  return hip_rc6(hipGetLastError());
}

// ---------------------------------------------------------------------------------------
bool scan_supported(Functor f, int nS, int nC) {
  return f == Functor::Logistic && (nS == 1 || nS == 2 || nS == 4) && nC == 1;
}

bool scan_problem_ok(const ProblemDesc& p) {
  if (p.functor == Functor::User) return user_rowsep(p.user) && (p.nS == 1 || p.nS == 2 || p.nS == 4) && p.nC == 1;
  return scan_supported(p.functor, p.nS, p.nC);
}


template <class P>
static void run_backward_scan(const BwdArgsScan& a, hipStream_t s) {
  constexpr int TPW = 64 / P::NS;
  const dim3 grid((a.batch + TPW - 1) / TPW), block(kScanW * 64);
  if (a.lamT) {
    if (a.lam && a.dJdu)
      k_backward_scan<P, kScanW, kScanL, true, true, true><<<grid, block, 0, s>>>(a);
    else if (a.lam)
      k_backward_scan<P, kScanW, kScanL, true, false, true><<<grid, block, 0, s>>>(a);
    else
      k_backward_scan<P, kScanW, kScanL, false, true, true><<<grid, block, 0, s>>>(a);
  } else if (a.lam && a.dJdu) {
#ifdef OCS_SCAN_ABL
    static const int abl = getenv("OCS_SCAN_ABL") ? atoi(getenv("OCS_SCAN_ABL")) : 0;
    if (abl == 1) return (void)(k_backward_scan<P, kScanW, kScanL, true, true, false, 1><<<grid, block, 0, s>>>(a));
    if (abl == 2) return (void)(k_backward_scan<P, kScanW, kScanL, true, true, false, 2><<<grid, block, 0, s>>>(a));
    if (abl == 3) return (void)(k_backward_scan<P, kScanW, kScanL, true, true, false, 3><<<grid, block, 0, s>>>(a));
    if (abl == 4) return (void)(k_backward_scan<P, kScanW, kScanL, true, true, false, 4><<<grid, block, 0, s>>>(a));
    if (abl == 5) return (void)(k_backward_scan<P, kScanW, kScanL, true, true, false, 5><<<grid, block, 0, s>>>(a));
#endif
    k_backward_scan<P, kScanW, kScanL, true, true, false><<<grid, block, 0, s>>>(a);
  } else if (a.lam) {
    k_backward_scan<P, kScanW, kScanL, true, false, false><<<grid, block, 0, s>>>(a);
  } else {
    k_backward_scan<P, kScanW, kScanL, false, true, false><<<grid, block, 0, s>>>(a);
  }
}

template <class P>
static void run_backward_vscan(const BwdArgsScan& a, hipStream_t s) {
  constexpr int W = VScanCfg<P::NS>::W, L = VScanCfg<P::NS>::L;
  static_assert(W == vscan_waves(P::NS) && L == kVScanL && L == kScanL, "launch shape of the hipRTC instances");
  const dim3 grid((a.batch + 63) / 64), block(W * 64);
  if (a.lamT) {
    if (a.lam && a.dJdu)
      k_backward_vscan<P, W, L, true, true, true><<<grid, block, 0, s>>>(a);
    else if (a.lam)
      k_backward_vscan<P, W, L, true, false, true><<<grid, block, 0, s>>>(a);
    else
      k_backward_vscan<P, W, L, false, true, true><<<grid, block, 0, s>>>(a);
  } else if (a.lam && a.dJdu) {
    k_backward_vscan<P, W, L, true, true, false><<<grid, block, 0, s>>>(a);
  } else if (a.lam) {
    k_backward_vscan<P, W, L, true, false, false><<<grid, block, 0, s>>>(a);
  } else {
    k_backward_vscan<P, W, L, false, true, false><<<grid, block, 0, s>>>(a);
  }
}
int launch_backward_vscan(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                          const double* lamT, double* lam, double* dJdu, double* lam0, const double* pend0,
                          hipStream_t s) {
  if (!vector_problem_ok(p) || (!lam && !dJdu) || g.N < kScanL || g.N % kScanL != 0 || batch < 1 || !g.RECS) return -1;
  const BwdArgsScan a{g.N, batch, g.RECS, p.ps, p.pb, p.pmask, xck, u, lamT, lam, dJdu, lam0, pend0};
  if (p.functor == Functor::User) {
    const int kid = lamT ? (lam && dJdu ? UK_VSCAN_LAM_DJDU_LT : (lam ? UK_VSCAN_LAM_LT : UK_VSCAN_DJDU_LT))
                         : (lam && dJdu ? UK_VSCAN_LAM_DJDU : (lam ? UK_VSCAN_LAM : UK_VSCAN_DJDU));
    void* args[] = {(void*)&a};
    return jit_launch(p.user, kid, dim3((batch + 63) / 64), dim3(vscan_waves(p.nS) * 64), args, s);
  }
  switch (p.nS) {
    case 1: run_backward_vscan<LogisticK<1>>(a, s); break;
    case 2: run_backward_vscan<LogisticK<2>>(a, s); break;
    case 3: run_backward_vscan<LogisticK<3>>(a, s); break;
    default: run_backward_vscan<LogisticK<4>>(a, s); break;
  }
  return hip_rc6(hipGetLastError());
}

// ---------------------------------------------------------------------------------------
// costate pass of the sweep as a scan (ocs_costate_scan_kernel.hpp)
bool costate_scan_ok(const ProblemDesc& p, const GridDesc& g, int batch) {
  static const bool off = getenv("OCS_COSTATE_SERIAL") != nullptr;   // the serial wave-specialised kernels, for A/B timing
  static const bool on = getenv("OCS_COSTATE_SCAN") != nullptr;       // ... and the scan at every batch
  // Selected while the chip has idle CUs (<= 128 workgroups): there the serial kernel's 1000-step chain sets the time
  // (measured, TestOCProblem, N = 1000, ms per solve of 12 sweeps, scan / serial: batch 2048 2.00 / 2.07, 4096 2.00 / 2.07,
  // 16 384 2.24 / 2.20 -- at BL-3's batch either pass moves its 24 B per instance-step at ~4.9 TB/s).
  const bool shape = g.RECS && g.N >= 8 && g.N % 8 == 0 && (p.nS == 1 || p.nS == 2 || p.nS == 4) && p.nC == 1 &&
                     tile_ok(batch, 64 / p.nS);
  // user problems that declare the costate equation free of u: the scan is their costate kernel at every batch
  if (p.functor == Functor::User) return user_fold(p.user) && shape;
  return !off && scan_supported(p.functor, p.nS, p.nC) && shape && (on || batch / (64 / p.nS) <= 128);
}
template <class P, bool MET>
static void run_costate_scan(const CostateScanArgs& a, hipStream_t s) {
  k_costate_scan<P, kScanW, kScanL, MET><<<dim3(tile_count(a.batch, 64 / P::NS)), dim3(kScanW * 64), 0, s>>>(a);
}
template <bool MET>
static int launch_costate_scan_t(const ProblemDesc& p, const CostateScanArgs& a, hipStream_t s) {
  if (p.functor == Functor::User) {
    if (!MET) return -1;   // (only the sweep's costate pass is instantiated for user problems)
    void* args[] = {(void*)&a};
    return jit_launch(p.user, UK_COSTATE_SCAN_MET, dim3(tile_count(a.batch, 64 / p.nS)), dim3(kScanW * 64), args, s);
  }
  if (p.nS == 1)
    run_costate_scan<LogisticK<1>, MET>(a, s);
  else if (p.nS == 2)
    run_costate_scan<LogisticK<2>, MET>(a, s);
  else
    run_costate_scan<LogisticK<4>, MET>(a, s);
  return hip_rc6(hipGetLastError());
}
int launch_costate_scan(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                        const int* frozen, double* lam, hipStream_t s, const int* gate) {
  if (!costate_scan_ok(p, g, batch) || !PR) return -1;
  CostateScanArgs a{};
  a.N = g.N; a.batch = batch; a.RECS = g.RECS; a.PR = PR; a.ps = p.ps; a.pb = p.pb; a.pmask = p.pmask;
  a.x = x; a.ldx = ldx; a.frozen = frozen; a.lam = lam; a.gate = gate;
  return launch_costate_scan_t<false>(p, a, s);
}
// any user problem given as row functions: the scan that reads the control samples (hipRTC instance)
bool costate_scan_u_ok(const ProblemDesc& p, const GridDesc& g, int batch) {
  return p.functor == Functor::User && user_rowsep(p.user) && g.RECS && g.N >= 8 && g.N % 8 == 0 &&
         (p.nS == 1 || p.nS == 2 || p.nS == 4) && p.nC == 1 && tile_ok(batch, 64 / p.nS);
}
int launch_costate_scan_u(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                          const double* u, const int* frozen, double* lam, hipStream_t s, const int* gate) {
  if (!costate_scan_u_ok(p, g, batch) || !PR || !u) return -1;
  CostateScanArgs a{};
  a.N = g.N; a.batch = batch; a.RECS = g.RECS; a.PR = PR; a.ps = p.ps; a.pb = p.pb; a.pmask = p.pmask;
  a.x = x; a.ldx = ldx; a.frozen = frozen; a.lam = lam; a.gate = gate; a.u = u;
  void* args[] = {(void*)&a};
  return jit_launch(p.user, UK_COSTATE_SCAN_U, dim3(tile_count(batch, 64 / p.nS)), dim3(kScanW * 64), args, s);
}
// any user problem given as full-vector methods, nS <= 4, nC <= 2 (hipRTC instance of k_costate_vscan)
bool costate_vscan_ok(const ProblemDesc& p, const GridDesc& g, int batch) {
  return p.functor == Functor::User && user_vector(p.user) && g.RECS && g.N >= 8 && g.N % 8 == 0 && batch >= 1;
}
int launch_costate_vscan(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                         const double* u, const int* frozen, double* lam, hipStream_t s, const int* gate) {
  if (!costate_vscan_ok(p, g, batch) || !PR || !u) return -1;
  CostateScanArgs a{};
  a.N = g.N; a.batch = batch; a.RECS = g.RECS; a.PR = PR; a.ps = p.ps; a.pb = p.pb; a.pmask = p.pmask;
  a.x = x; a.ldx = ldx; a.frozen = frozen; a.lam = lam; a.gate = gate; a.u = u;
  void* args[] = {(void*)&a};
  return jit_launch(p.user, UK_COSTATE_VSCAN, dim3((batch + 63) / 64), dim3(vscan_waves(p.nS) * 64), args, s);
}
int launch_costate_scan_met(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                            const double* lb, const double* ub, double relTol, double absTol, int sweep, int* status,
                            double* maxChange, int* nactive, double* lam, hipStream_t s, const int* gate) {
  if (!costate_scan_ok(p, g, batch) || !PR || !g.TU || !status || !maxChange || !nactive) return -1;
  CostateScanArgs a{};
  a.N = g.N; a.batch = batch; a.RECS = g.RECS; a.PR = PR; a.ps = p.ps; a.pb = p.pb; a.pmask = p.pmask;
  a.x = x; a.ldx = ldx; a.frozen = status; a.lam = lam; a.gate = gate;
  a.TU = g.TU; a.lb = lb; a.ub = ub; a.relTol = relTol; a.absTol = absTol; a.sweep = sweep; a.status = status;
  a.maxChange = maxChange; a.nactive = nactive;
  return launch_costate_scan_t<true>(p, a, s);
}

int scan_chunk_steps() { return kScanL; }
int launch_backward_scan(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                         const double* lamT, double* lam, double* dJdu, double* lam0, const double* pend0,
                         hipStream_t s) {
  if (!scan_problem_ok(p) || (!lam && !dJdu) || g.N < kScanL || g.N % kScanL != 0 || batch < 1 || !g.RECS) return -1;
  const BwdArgsScan a{g.N, batch, g.RECS, p.ps, p.pb, p.pmask, xck, u, lamT, lam, dJdu, lam0, pend0};
  if (p.functor == Functor::User) {   // the hipRTC instances of the same kernel template
    const int TPW = 64 / p.nS;
    const int kid = lamT ? (lam && dJdu ? UK_SCAN_LAM_DJDU_LT : (lam ? UK_SCAN_LAM_LT : UK_SCAN_DJDU_LT))
                         : (lam && dJdu ? UK_SCAN_LAM_DJDU : (lam ? UK_SCAN_LAM : UK_SCAN_DJDU));
    void* args[] = {(void*)&a};
    return jit_launch(p.user, kid, dim3((batch + TPW - 1) / TPW), dim3(kScanW * 64), args, s);
  }
  if (p.nS == 1)
    run_backward_scan<LogisticK<1>>(a, s);
  else if (p.nS == 2)
    run_backward_scan<LogisticK<2>>(a, s);
  else
    run_backward_scan<LogisticK<4>>(a, s);
  return hip_rc6(hipGetLastError());
}

}  // namespace ocs
