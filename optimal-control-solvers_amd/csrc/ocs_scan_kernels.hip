// ocs_scan_kernels.hip -- launchers of the scan adjoint kernel (ocs_scan_kernel.hpp) for the registry problems, its
// record table, and the dispatch to the hipRTC instances of user problems given as row functions.
#include "ocs_scan_kernel.hpp"
#include "ocs_vscan_kernel.hpp"
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
