// ocs_rowsplit_kernels.hip -- row-split mapping of the RK4 state / discrete-adjoint passes for
// row-separable problems (ocs_problems.hpp, ROW_SEPARABLE): a group of G = nS adjacent lanes owns
// one trajectory, one state row per lane; a 64-lane wave owns 64/G trajectories.
//
// Why: at small batch the passes are bound by the instruction issue of one wave per SIMD (one fp64
// instruction per ~5.4 cycles) on a chip that is mostly idle (batch 4096 = 64 waves on 1024 SIMDs).
// Splitting the rows over lanes divides the per-wave instruction count of a step by ~2.7 and
// multiplies the number of busy SIMDs by G.  The rows meet only in two places, both done with DPP
// quad permutes (no LDS): the running objective (each lane integrates its row's share q_r of the
// integrand, the shares are summed when the cost row is stored) and the dJdu columns.
// Summation order differs from the lane-per-trajectory kernels and from the reference only in the
// association of these per-row sums (fp64 round-off, covered by the stated 1e-12 tolerance).
//
// Same arrays, same batch-minor layout, same semantics as k_forward / k_backward
// (RK4Integrator.m:28-121); the launcher picks the mapping.
#include "ocs_device_common.hpp"
#include "ocs_internal.hpp"
#include "ocs_problems.hpp"

namespace ocs {

static inline int hip_rc4(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// fp64 cross-lane move inside a quad (DPP quad_perm on the two 32-bit halves)
template <int CTRL>
__device__ static inline double dpp_quad(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  const int hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi2, lo2);
}
// sum over the G lanes of a group; every lane of the group gets the same bits
template <int G>
__device__ static inline double group_sum(double v) {
  static_assert(G == 2 || G == 4, "group size");
  v += dpp_quad<0xB1>(v);               // quad_perm [1,0,3,2]: lane ^ 1
  if (G == 4) v += dpp_quad<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  return v;
}

struct FwdArgsRS {
  int N, batch;
  const double* REC;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* x0;
  const double* u;
  double* x;
  double* J;
};

template <class P, int CH, int PF, bool OUT_X>
__global__ __launch_bounds__(64) void k_forward_rs(const FwdArgsRS a) {
  constexpr int G = P::NS, NC = P::NC, NTC = P::NTC, NAUG = P::NAUG;
  static_assert(NC == 1 && NTC == 1, "row-split kernels are written for one control and one time coefficient");
  using Rec = StepRec<NTC>;
  constexpr int TPW = 64 / G;
  const int r = threadIdx.x % G;
  const int b0 = blockIdx.x * TPW + threadIdx.x / G;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  const uniform_ptr PS = as_uniform(a.ps);
  const typename P::RowPar rp = P::load_row([&](int k) OCS_INLINE {
    return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + b] : PS[k];
  }, r);
  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));

  double y = a.x0[(size_t)r * B + b];
  double pc = 0.0;  // this row's share of the running objective
  const size_t colB = (size_t)NAUG * B;
  double* xs = a.x + (size_t)r * B + b;   // x(r, i)
  double* xc = a.x + (size_t)G * B + b;   // x(end, i): written by every lane of the group (same value)
  if (OUT_X) {
    *xs = y;
    *xc = 0.0;
  }
  const double* up = a.u + b;
  double uprev = *up, uprev2 = uprev * uprev;
  up += B;

  auto step = [&](const Rec& rc, double uA, double uA2, double uM, double uM2, double uB, double uB2) OCS_INLINE {
    const double F1 = P::row_f(y, uA, rp), q1 = rc.tcA[0] * P::row_q(y, uA2, rp);
    double Y = __builtin_fma(rc.hh, F1, y);
    const double F2 = P::row_f(Y, uM, rp), q2 = rc.tcM[0] * P::row_q(Y, uM2, rp);
    Y = __builtin_fma(rc.hh, F2, y);
    const double F3 = P::row_f(Y, uM, rp), q3 = rc.tcM[0] * P::row_q(Y, uM2, rp);
    Y = __builtin_fma(rc.h, F3, y);
    const double F4 = P::row_f(Y, uB, rp), q4 = rc.tcB[0] * P::row_q(Y, uB2, rp);
    y = __builtin_fma(rc.h6, __builtin_fma(2.0, F3, __builtin_fma(2.0, F2, F1)) + F4, y);
    pc = __builtin_fma(rc.h6, __builtin_fma(2.0, q3, __builtin_fma(2.0, q2, q1)) + q4, pc);
    if (OUT_X) {
      xs += colB;
      xc += colB;
      *xs = y;
      *xc = group_sum<G>(pc);
    }
  };

  static_assert(PF <= kRecPad, "record ring deeper than the table padding");
  Rec rq[PF];
  const double* recp = a.REC;  // walks forward one record per step; the table is padded past step N-1
#pragma unroll
  for (int q = 0; q < PF; ++q) {
    rq[q] = load_rec<NTC>(recp);
    recp += rec_stride(NTC);
  }
  auto next_rec = [&]() OCS_INLINE {
    const Rec cur = rq[0];
#pragma unroll
    for (int q = 0; q + 1 < PF; ++q) rq[q] = rq[q + 1];
    rq[PF - 1] = load_rec<NTC>(recp);
    recp += rec_stride(NTC);
    return cur;
  };
  double ub0[2 * CH], ub1[2 * CH];
  auto load_chunk = [&](double (&dst)[2 * CH]) OCS_INLINE {
#pragma unroll
    for (int s = 0; s < 2 * CH; ++s) {
      dst[s] = *up;
      up += B;
    }
  };
  auto run_chunk = [&](const double (&src)[2 * CH]) OCS_INLINE {
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      const Rec cur = next_rec();
      const double uM = src[2 * s], uB = src[2 * s + 1];
      const double uM2 = uM * uM, uB2 = uB * uB;
      step(cur, uprev, uprev2, uM, uM2, uB, uB2);
      uprev = uB;
      uprev2 = uB2;
    }
  };
  const int nch = N / CH;
  if (nch > 0) load_chunk(ub0);
  int c = 0;
  for (; c + 1 < nch; c += 2) {
    load_chunk(ub1);
    run_chunk(ub0);
    if (c + 2 < nch) load_chunk(ub0);
    run_chunk(ub1);
  }
  if (c < nch) run_chunk(ub0);
  for (int i = nch * CH; i < N; ++i) {
    const double uM = *up;
    up += B;
    const double uB = *up;
    up += B;
    const Rec cur = next_rec();
    const double uM2 = uM * uM, uB2 = uB * uB;
    step(cur, uprev, uprev2, uM, uM2, uB, uB2);
    uprev = uB;
    uprev2 = uB2;
  }
  const double Jt = group_sum<G>(pc);
  a.J[b] = Jt;
  if (warm == 1.234567e300) a.J[b] = warm;  // never true; keeps the table sweep alive
}

struct BwdArgsRS {
  int N, batch;
  const double* REC;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* xck;
  const double* u;
  const double* lamT;
  double* lam;
  double* dJdu;
  double* lam0;
};

template <class P, int CH, int PF, bool OUT_LAM, bool OUT_DJDU>
__global__ __launch_bounds__(64) void k_backward_rs(const BwdArgsRS a) {
  constexpr int G = P::NS, NTC = P::NTC, NAUG = P::NAUG;
  using Rec = StepRec<NTC>;
  constexpr int TPW = 64 / G;
  const int r = threadIdx.x % G;
  const int b0 = blockIdx.x * TPW + threadIdx.x / G;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  const uniform_ptr PS = as_uniform(a.ps);
  const typename P::RowPar rp = P::load_row([&](int k) OCS_INLINE {
    return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + b] : PS[k];
  }, r);
  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));

  double lam, lamc;
  if (a.lamT) {
    lam = a.lamT[(size_t)r * B + b];
    lamc = a.lamT[(size_t)G * B + b];
  } else {
    lam = 0.0;
    lamc = 1.0;
  }
  const size_t colB = (size_t)NAUG * B;
  double* ls = a.lam + (size_t)N * colB + (size_t)r * B + b;  // lam(r, N+1), walks down by one column per step
  double* lc = a.lam + (size_t)N * colB + (size_t)G * B + b;  // lam(end, .)
  if (OUT_LAM) {
    *ls = lam;
    *lc = lamc;
  }
  const double* up = a.u + (size_t)(2 * N) * B + b;            // u(2N+1)
  const double* xp = a.xck + (size_t)N * colB + (size_t)r * B + b;  // x(r, N+1)
  double* dp = a.dJdu + (size_t)(2 * N) * B + b;               // dJdu(2N+1)
  double unext = *up, cunext = rp.cw * unext, pend = 0.0;

  auto step = [&](const Rec& rc, double xi, double uA, double uM, double uB, double cuB) OCS_INLINE {
    const double cuA = rp.cw * uA, cuM = rp.cw * uM;
    // stage states of this row, recomputed (compute_states :39-46)
    double f = P::row_f(xi, uA, rp);
    const double Y2 = __builtin_fma(rc.hh, f, xi);
    f = P::row_f(Y2, uM, rp);
    const double Y3 = __builtin_fma(rc.hh, f, xi);
    f = P::row_f(Y3, uM, rp);
    const double Y4 = __builtin_fma(rc.h, f, xi);
    // cost-row entries of dJdk are multiples of the constant lam(end,:)   :73,77,81,85
    const double k4c = rc.h6 * lamc, k3c = rc.h3 * lamc;
    const double ev4 = (2.0 * rc.tcB[0]) * k4c, ev3 = (2.0 * rc.tcM[0]) * k3c, ev1 = (2.0 * rc.tcA[0]) * k4c;
    const double h6l = rc.h6 * lam, h3l = rc.h3 * lam;
    const double k4 = h6l;                                   // :73
    const double g3 = P::row_dfdx(Y4, k4, ev4, rp);          // :74-75
    const double k3 = __builtin_fma(rc.h, g3, h3l);          // :77
    const double g2 = P::row_dfdx(Y3, k3, ev3, rp);          // :78-79
    const double k2 = __builtin_fma(rc.hh, g2, h3l);         // :81
    const double g1 = P::row_dfdx(Y2, k2, ev3, rp);          // :82-83
    const double k1 = __builtin_fma(rc.hh, g1, h6l);         // :85
    const double g0 = P::row_dfdx(xi, k1, ev1, rp);          // :87-88
    if (OUT_DJDU) {  // compute_dJdu :97-121: per-row shares, summed over the group per column
      const double p4 = P::row_dfdu(cuB, k4, ev4), p3 = P::row_dfdu(cuM, k3, ev3);
      const double p2 = P::row_dfdu(cuM, k2, ev3), p1 = P::row_dfdu(cuA, k1, ev1);
      *dp = group_sum<G>(pend + p4);   // column 2i+2
      dp -= B;
      *dp = group_sum<G>(p2 + p3);     // column 2i+1
      dp -= B;
      pend = p1;
    }
    lam = (((lam + g1) + g2) + g3) + g0;                     // :86-88
    if (OUT_LAM) {
      ls -= colB;
      lc -= colB;
      *ls = lam;
      *lc = lamc;
    }
  };

  static_assert(PF <= kRecPad, "record ring deeper than the table padding");
  Rec rq[PF];
  const double* recp = a.REC + (size_t)(N - 1) * rec_stride(NTC);  // walks down; padded before step 0
#pragma unroll
  for (int q = 0; q < PF; ++q) {
    rq[q] = load_rec<NTC>(recp);
    recp -= rec_stride(NTC);
  }
  auto next_rec = [&]() OCS_INLINE {
    const Rec cur = rq[0];
#pragma unroll
    for (int q = 0; q + 1 < PF; ++q) rq[q] = rq[q + 1];
    rq[PF - 1] = load_rec<NTC>(recp);
    recp -= rec_stride(NTC);
    return cur;
  };

  const int nch = N / CH;
  for (int i = N - 1; i >= nch * CH; --i) {  // remainder steps at the top, direct loads
    xp -= colB;
    const double xi = *xp;
    up -= B;
    const double uM = *up;
    up -= B;
    const double uA = *up;
    const Rec cur = next_rec();
    step(cur, xi, uA, uM, unext, cunext);
    unext = uA;
    cunext = rp.cw * uA;
  }
  double xb0[CH], xb1[CH], ub0[2 * CH], ub1[2 * CH];
  auto load_chunk = [&](double (&xd)[CH], double (&ud)[2 * CH]) OCS_INLINE {
#pragma unroll
    for (int s = CH - 1; s >= 0; --s) {
      xp -= colB;
      xd[s] = *xp;
    }
#pragma unroll
    for (int s = 2 * CH - 1; s >= 0; --s) {
      up -= B;
      ud[s] = *up;
    }
  };
  auto run_chunk = [&](const double (&xs_)[CH], const double (&us)[2 * CH]) OCS_INLINE {
#pragma unroll
    for (int s = CH - 1; s >= 0; --s) {
      const Rec cur = next_rec();
      step(cur, xs_[s], us[2 * s], us[2 * s + 1], unext, cunext);
      unext = us[2 * s];
      cunext = rp.cw * unext;
    }
  };
  int c = nch - 1;
  if (c >= 0) load_chunk(xb0, ub0);
  for (; c >= 1; c -= 2) {
    load_chunk(xb1, ub1);
    run_chunk(xb0, ub0);
    if (c >= 2) load_chunk(xb0, ub0);
    run_chunk(xb1, ub1);
  }
  if (c == 0) run_chunk(xb0, ub0);

  if (OUT_DJDU) *dp = group_sum<G>(pend);  // left end point :101-102
  if (a.lam0) {
    a.lam0[(size_t)r * B + b] = lam;
    a.lam0[(size_t)G * B + b] = lamc;
  }
  if (warm == 1.234567e300) {  // never true; keeps the table sweep alive
    if (OUT_DJDU) *dp = warm;
    if (OUT_LAM) *ls = warm;
    if (a.lam0) a.lam0[b] = warm;
  }
}

// ---------------------------------------------------------------------------------------
constexpr int kChunkRS = 4;
constexpr int kPFRS = 4;  // divides 2*kChunkRS: no ring rotation at the loop back-edge

bool rowsplit_supported(Functor f, int nS, int nC) {
  return f == Functor::Logistic && (nS == 2 || nS == 4) && nC == 1;
}

template <class P>
static void run_forward_rs(const FwdArgsRS& a, hipStream_t s) {
  constexpr int TPW = 64 / P::NS;
  const dim3 grid((a.batch + TPW - 1) / TPW), block(64);
  if (a.x)
    k_forward_rs<P, kChunkRS, kPFRS, true><<<grid, block, 0, s>>>(a);
  else
    k_forward_rs<P, kChunkRS, kPFRS, false><<<grid, block, 0, s>>>(a);
}
int launch_forward_rs(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                      double* x, double* J, hipStream_t s) {
  const FwdArgsRS a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, x0, u, x, J};
  if (p.nS == 2)
    run_forward_rs<LogisticK<2>>(a, s);
  else if (p.nS == 4)
    run_forward_rs<LogisticK<4>>(a, s);
  else
    return -1;
  return hip_rc4(hipGetLastError());
}

template <class P>
static void run_backward_rs(const BwdArgsRS& a, hipStream_t s) {
  constexpr int TPW = 64 / P::NS;
  const dim3 grid((a.batch + TPW - 1) / TPW), block(64);
  if (a.lam && a.dJdu)
    k_backward_rs<P, kChunkRS, kPFRS, true, true><<<grid, block, 0, s>>>(a);
  else if (a.lam)
    k_backward_rs<P, kChunkRS, kPFRS, true, false><<<grid, block, 0, s>>>(a);
  else
    k_backward_rs<P, kChunkRS, kPFRS, false, true><<<grid, block, 0, s>>>(a);
}
int launch_backward_rs(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                       const double* lamT, double* lam, double* dJdu, double* lam0, hipStream_t s) {
  if (!lam && !dJdu) return -1;
  const BwdArgsRS a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, xck, u, lamT, lam, dJdu, lam0};
  if (p.nS == 2)
    run_backward_rs<LogisticK<2>>(a, s);
  else if (p.nS == 4)
    run_backward_rs<LogisticK<4>>(a, s);
  else
    return -1;
  return hip_rc4(hipGetLastError());
}

}  // namespace ocs
