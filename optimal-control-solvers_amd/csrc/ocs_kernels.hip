// ocs_kernels.hip -- gfx950 (MI355X) kernels for the batched RK4 state pass and its exact
// discrete adjoint.  fp64 throughout.
//
// Mapping ("L", lane-per-trajectory): one lane integrates one trajectory; a 64-lane wave
// (= one workgroup) owns 64 consecutive trajectories.  All per-step quantities that do not
// depend on the trajectory (step sizes, time coefficients) are wave-uniform and come from
// scalar loads.  Arrays are batch-minor ([time][row][batch]) so every vector load/store of a
// wave is one contiguous 512-B segment: coalescing needs no LDS staging in this mapping.
// The time recursion is serial, so at small batch the passes are bound by fp64 issue on one
// wave per SIMD, not by HBM; control samples and checkpoints are
// software-prefetched CH steps ahead in registers so that HBM latency stays off the chain.
//
// Reference semantics: Integrator/RK4Integrator.m:28-56 (compute_states), :59-94
// (compute_adjoints), :97-121 (compute_dJdu).  The reference caches all four stage states
// (xK); here only y_i is checkpointed and Y2..Y4 are recomputed in the adjoint pass (same
// arithmetic, deterministic => same values), which cuts checkpoint traffic 4x.
#include "ocs_device_common.hpp"
#include "ocs_internal.hpp"
#include "ocs_problems.hpp"

namespace ocs {

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// ---------------------------------------------------------------------------------------
// time-coefficient table
// ---------------------------------------------------------------------------------------
template <class P>
__global__ void k_tcoef(int nT, const double* __restrict__ T, const double* __restrict__ ps,
                        double* __restrict__ TC, double* __restrict__ TU) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nT) return;
  double tc[P::NTC], tu[P::NTU];
  P::tcoef(T[j], ps, tc, tu);
#pragma unroll
  for (int k = 0; k < P::NTC; ++k) TC[(size_t)j * P::NTC + k] = tc[k];
  if (TU) {
#pragma unroll
    for (int k = 0; k < P::NTU; ++k) TU[(size_t)j * P::NTU + k] = tu[k];
  }
}

// ---------------------------------------------------------------------------------------
// per-step record table (layout: ocs_device_common.hpp)
// ---------------------------------------------------------------------------------------
// REC points at the record of step 0; records -kRecPad..-1 and N..N+kRecPad-1 are edge copies.
template <class P>
__global__ void k_build_rec(int N, const double* __restrict__ HT, const double* __restrict__ TC,
                            double* __restrict__ REC) {
  constexpr int NTC = P::NTC, RS = rec_stride(NTC), SCO = rec_sc_offset(NTC);
  const int ip = blockIdx.x * blockDim.x + threadIdx.x - kRecPad;
  if (ip >= N + kRecPad) return;
  const int i = ip < 0 ? 0 : (ip >= N ? N - 1 : ip);
  double* r = REC + (long long)ip * RS;
  for (int k = 0; k < 4; ++k) r[k] = HT[4 * i + k];
  for (int k = 0; k < 3 * NTC; ++k) r[4 + k] = TC[(size_t)(2 * i) * NTC + k];  // three consecutive grid points
  for (int k = 4 + 3 * NTC; k < RS; ++k) r[k] = 0.0;
  double sc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  static_assert(P::NSC <= 8, "at most 8 step constants");
  P::step_consts(r[2], r[3], r + 4, r + 4 + NTC, r + 4 + 2 * NTC, sc);
  for (int k = 0; k < P::NSC; ++k) r[SCO + k] = sc[k];
}

// ---------------------------------------------------------------------------------------
// forward pass: [x, J] = compute_states(obj, prob, x0, u)      RK4Integrator.m:28-56
// ---------------------------------------------------------------------------------------
struct FwdArgs {
  int N, batch;
  const double* REC;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* x0;
  const double* u;      // [2N+1][nC][B]; with UCONST: [nC], the same for every grid point and trajectory
  double* x;
  double* J;
  const double* Jadd;   // optional [B]: J = Jadd + x(end,end)  (RK4InfiniteIntegrator.m:23  J = J1 + J2)
  const int* usel;      // optional [B]: trajectory b reads its controls from u + usel[b] * udelta
  long long udelta;     //              (fb_sweep keeps the old and the new control in two buffers)
};

// Lanes past the end of the batch are clamped onto the last trajectory: they recompute it and
// store the same values to the same addresses, which keeps the step body free of exec-mask
// branches (one basic block per chunk, so the scheduler can hoist the prefetch loads).
template <class P, int CH, int PF, bool OUT_X, bool UCONST>
__global__ __launch_bounds__(64) void k_forward(const FwdArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC;
  using Rec = StepRec<NTC>;
  const int b0 = blockIdx.x * 64 + threadIdx.x;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  const uniform_ptr PS = as_uniform(a.ps);
  const double* REC = a.REC;

  const typename P::Par p = P::load([&](int k) OCS_INLINE {
    return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + b] : PS[k];
  });

  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));

  double y[NS], yc = 0.0;  // xK(:,1,1) = [x0; 0]   :33
#pragma unroll
  for (int k = 0; k < NS; ++k) y[k] = a.x0[(size_t)k * B + b];
  // All per-trajectory arrays are walked with one pointer that advances by B doubles per row:
  // rows of consecutive time points are B apart, so no per-row offset arithmetic is needed.
  double* xo = a.x + b;
  if (OUT_X) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      *xo = y[k];
      xo += B;
    }
    *xo = 0.0;
    xo += B;
  }

  const double* up = a.u;  // walks u(:,j) row by row
  double uprev[NC];
  if (UCONST) {
#pragma unroll
    for (int c = 0; c < NC; ++c) uprev[c] = PSU(a.u)[c];
  } else {
    up += b;
    if (a.usel) up += (long long)a.usel[b] * a.udelta;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uprev[c] = *up;
      up += B;
    }
  }

  // one RK4 step, controls at grid points 2i (uA), 2i+1 (uM), 2i+2 (uB)   :36-51
  auto step = [&](const Rec& r, const double* uA, const double* uM, const double* uB) OCS_INLINE {
    double F1[NS + 1], F2[NS + 1], F3[NS + 1], F4[NS + 1], Y[NS];
    P::F(r.tcA, y, uA, p, F1);                                             // :39
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.hh, F1[k], y[k]);  // :40
    P::F(r.tcM, Y, uM, p, F2);                                             // :42
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.hh, F2[k], y[k]);  // :43
    P::F(r.tcM, Y, uM, p, F3);                                             // :45
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.h, F3[k], y[k]);   // :46
    P::F(r.tcB, Y, uB, p, F4);                                             // :48
#pragma unroll
    for (int k = 0; k < NS; ++k)                                           // :50-51
      y[k] = __builtin_fma(r.h6, __builtin_fma(2.0, F3[k], __builtin_fma(2.0, F2[k], F1[k])) + F4[k], y[k]);
    yc = __builtin_fma(r.h6, __builtin_fma(2.0, F3[NS], __builtin_fma(2.0, F2[NS], F1[NS])) + F4[NS], yc);
    if (OUT_X) {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        *xo = y[k];
        xo += B;
      }
      *xo = yc;
      xo += B;
    }
  };

  // controls are prefetched one chunk (CH steps = 2*CH samples) ahead, ping-pong in registers;
  // the uniform step record is fetched one step ahead
  double ub0[2 * CH][NC], ub1[2 * CH][NC];
  auto load_chunk = [&](double (&dst)[2 * CH][NC]) OCS_INLINE {
#pragma unroll
    for (int s = 0; s < 2 * CH; ++s)
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        dst[s][cc] = *up;
        up += B;
      }
  };
  // uniform step records are fetched PF steps ahead (scalar loads miss the scalar cache on
  // every new 64-byte record, so one step of lead does not cover the L2 round trip)
  static_assert(PF <= kRecPad, "record ring deeper than the table padding");
  Rec rq[PF];
  const double* recp = REC;  // walks forward one record per step; the table is padded past step N-1
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
  auto run_chunk = [&](const double (&src)[2 * CH][NC]) OCS_INLINE {
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      const Rec cur = next_rec();
      const double* uA = (s == 0) ? uprev : src[2 * s - 1];
      step(cur, uA, src[2 * s], src[2 * s + 1]);
    }
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) uprev[cc] = src[2 * CH - 1][cc];
  };

  if (UCONST) {  // constant control (tail leg of RK4InfiniteIntegrator.m:15): nothing to load
    for (int i = 0; i < N; ++i) {
      const Rec cur = next_rec();
      step(cur, uprev, uprev, uprev);
    }
    a.J[b] = a.Jadd ? a.Jadd[b] + yc : yc;
    if (warm == 1.234567e300) a.J[b] = warm;
    return;
  }
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
  for (int i = nch * CH; i < N; ++i) {  // remainder steps, direct loads
    double uM[NC], uB[NC];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      uM[cc] = *up;
      up += B;
    }
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      uB[cc] = *up;
      up += B;
    }
    const Rec cur = next_rec();
    step(cur, uprev, uM, uB);
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) uprev[cc] = uB[cc];
  }
  a.J[b] = a.Jadd ? a.Jadd[b] + yc : yc;  // J = x(end,end)   :55
  if (warm == 1.234567e300) a.J[b] = warm;  // never true for a table of step sizes; keeps the sweep alive
}

// ---------------------------------------------------------------------------------------
// adjoint pass: [lam, dJdu] = compute_adjoints(obj, prob, u, lamT)   RK4Integrator.m:59-121
// ---------------------------------------------------------------------------------------
struct BwdArgs {
  int N, batch;
  const double* REC;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* xck;   // checkpoints y_i = x(:,i): [N+1][nAug][B]
  const double* u;
  const double* lamT;  // [nAug][B] or nullptr (default e_last :63-66)
  double* lam;
  double* dJdu;
  double* lam0;         // optional [nAug][B]: lam(:,1) only (RK4InfiniteIntegrator.m:29, single_shooting.m:149)
  const int* usel;      // as in FwdArgs
  long long udelta;
};

template <class P, int CH, int PF, bool OUT_LAM, bool OUT_DJDU, bool UCONST>
__global__ __launch_bounds__(64) void k_backward(const BwdArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC, NAUG = P::NAUG;
  using Rec = StepRec<NTC>;
  const int b0 = blockIdx.x * 64 + threadIdx.x;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  const uniform_ptr PS = as_uniform(a.ps);
  const double* REC = a.REC;

  const typename P::Par p = P::load([&](int k) OCS_INLINE {
    return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + b] : PS[k];
  });

  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));

  double lam[NS], lamc;  // lam(:,end) = lamT   :69.  lamc = lam(end,:) is constant: the last row
  if (a.lamT) {          // of dFdx_times_vec is 0 (OCProblem.m:14-15), adding zeros is exact.
#pragma unroll
    for (int k = 0; k < NS; ++k) lam[k] = a.lamT[(size_t)k * B + b];
    lamc = a.lamT[(size_t)NS * B + b];
  } else {
#pragma unroll
    for (int k = 0; k < NS; ++k) lam[k] = 0.0;
    lamc = 1.0;
  }
  // every array is walked downwards one row (B doubles) at a time, last row of a column first
  double* lo = a.lam + b + ((size_t)(N + 1) * NAUG) * B;      // one past lam(end,N+1)
  const double* up = a.u;
  if (!UCONST) {
    up += b + ((size_t)(2 * N + 1) * NC) * B;  // one past u(end,2N+1)
    if (a.usel) up += (long long)a.usel[b] * a.udelta;
  }
  const double* xp = a.xck + b + ((size_t)N * NAUG) * B;        // x(1,N+1): one past x(end,N)
  double* dp = a.dJdu + b + ((size_t)(2 * N + 1) * NC) * B;     // one past dJdu(end,2N+1)
  if (OUT_LAM) {
    lo -= B;
    *lo = lamc;
#pragma unroll
    for (int k = NS - 1; k >= 0; --k) {
      lo -= B;
      *lo = lam[k];
    }
  }

  double unext[NC], pend[NC];  // u(:,2i+3) carried from the step above; k1-term of that step
#pragma unroll
  for (int c = NC - 1; c >= 0; --c) {
    if (UCONST) {
      unext[c] = PSU(a.u)[c];
    } else {
      up -= B;
      unext[c] = *up;
    }
    pend[c] = 0.0;
  }

  // reverse of RK4 step i: xi = y_i (checkpoint), controls uA (2i), uM (2i+1), uB (2i+2)
  auto step = [&](const Rec& r, const double* xi, const double* uA, const double* uM, const double* uB) OCS_INLINE {
    // stage states xK(:,i,2:4), recomputed (compute_states :39-46)
    double f[NS], Y2[NS], Y3[NS], Y4[NS];
    P::Fx(r.tcA, xi, uA, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y2[k] = __builtin_fma(r.hh, f[k], xi[k]);
    P::Fx(r.tcM, Y2, uM, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y3[k] = __builtin_fma(r.hh, f[k], xi[k]);
    P::Fx(r.tcM, Y3, uM, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y4[k] = __builtin_fma(r.h, f[k], xi[k]);
    // dJdk(:,i,4..1) and the dJdx terms   :73-88
    double k4[NAUG], k3[NAUG], k2[NAUG], k1[NAUG], g3[NS], g2[NS], g1[NS], g0[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) k4[k] = r.h6 * lam[k];                             // :73
    k4[NS] = r.h6 * lamc;
    P::dFdxT(r.tcB, Y4, uB, p, k4, g3);                                             // :74-75
#pragma unroll
    for (int k = 0; k < NS; ++k) k3[k] = __builtin_fma(r.h, g3[k], r.h3 * lam[k]);  // :77
    k3[NS] = r.h3 * lamc;
    P::dFdxT(r.tcM, Y3, uM, p, k3, g2);                                             // :78-79
#pragma unroll
    for (int k = 0; k < NS; ++k) k2[k] = __builtin_fma(r.hh, g2[k], r.h3 * lam[k]); // :81
    k2[NS] = r.h3 * lamc;
    P::dFdxT(r.tcM, Y2, uM, p, k2, g1);                                             // :82-83
#pragma unroll
    for (int k = 0; k < NS; ++k) k1[k] = __builtin_fma(r.hh, g1[k], r.h6 * lam[k]); // :85
    k1[NS] = r.h6 * lamc;
    P::dFdxT(r.tcA, xi, uA, p, k1, g0);                                             // :87-88
    if (OUT_DJDU) {  // compute_dJdu :97-121, fused: column 2i+2 pairs k4 of step i with k1 of step i+1
      double d4[NC], d3[NC], d2[NC];
      P::dFduT(r.tcB, Y4, uB, p, k4, d4);
      P::dFduT(r.tcM, Y3, uM, p, k3, d3);
      P::dFduT(r.tcM, Y2, uM, p, k2, d2);
#pragma unroll
      for (int c = NC - 1; c >= 0; --c) {
        dp -= B;
        *dp = pend[c] + d4[c];  // column 2i+2  :112-116 (:119-120 at i = N-1)
      }
#pragma unroll
      for (int c = NC - 1; c >= 0; --c) {
        dp -= B;
        *dp = d2[c] + d3[c];    // column 2i+1  :105-109
      }
      P::dFduT(r.tcA, xi, uA, p, k1, pend);
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) lam[k] = (((lam[k] + g1[k]) + g2[k]) + g3[k]) + g0[k];  // :86-88
    if (OUT_LAM) {
      lo -= B;
      *lo = lamc;
#pragma unroll
      for (int k = NS - 1; k >= 0; --k) {
        lo -= B;
        *lo = lam[k];
      }
    }
  };

  const int nch = UCONST ? 0 : N / CH;
  static_assert(PF <= kRecPad, "record ring deeper than the table padding");
  Rec rq[PF];
  const double* recp = REC + (size_t)(N - 1) * rec_stride(NTC);  // walks down; padded before step 0
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
  // remainder steps at the top of the grid first (i = N-1 .. nch*CH), direct loads
  for (int i = N - 1; i >= nch * CH; --i) {
    double xi[NS], uA[NC], uM[NC];
    xp -= B;  // skip the cost row: y(end) is never read (OCProblem.m:14-15)
#pragma unroll
    for (int k = NS - 1; k >= 0; --k) {
      xp -= B;
      xi[k] = *xp;
    }
    if (UCONST) {
#pragma unroll
      for (int c = 0; c < NC; ++c) uA[c] = uM[c] = unext[c];
    } else {
#pragma unroll
      for (int c = NC - 1; c >= 0; --c) {
        up -= B;
        uM[c] = *up;
      }
#pragma unroll
      for (int c = NC - 1; c >= 0; --c) {
        up -= B;
        uA[c] = *up;
      }
    }
    const Rec cur = next_rec();
    step(cur, xi, uA, uM, unext);
#pragma unroll
    for (int c = 0; c < NC; ++c) unext[c] = uA[c];
  }

  // a chunk covers CH steps: checkpoints x(:,i) and the 2*CH samples below the carried one
  double xb0[CH][NS], xb1[CH][NS], ub0[2 * CH][NC], ub1[2 * CH][NC];
  auto load_chunk = [&](double (&xd)[CH][NS], double (&ud)[2 * CH][NC]) OCS_INLINE {
#pragma unroll
    for (int s = CH - 1; s >= 0; --s) {
      xp -= B;  // cost row skipped
#pragma unroll
      for (int k = NS - 1; k >= 0; --k) {
        xp -= B;
        xd[s][k] = *xp;
      }
    }
#pragma unroll
    for (int s = 2 * CH - 1; s >= 0; --s)
#pragma unroll
      for (int cc = NC - 1; cc >= 0; --cc) {
        up -= B;
        ud[s][cc] = *up;
      }
  };
  auto run_chunk = [&](const double (&xs)[CH][NS], const double (&us)[2 * CH][NC]) OCS_INLINE {
#pragma unroll
    for (int s = CH - 1; s >= 0; --s) {
      const Rec cur = next_rec();
      const double* uB = (s == CH - 1) ? unext : us[2 * s + 2];
      step(cur, xs[s], us[2 * s], us[2 * s + 1], uB);
    }
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) unext[cc] = us[0][cc];
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

  if (OUT_DJDU) {
#pragma unroll
    for (int cc = NC - 1; cc >= 0; --cc) {
      dp -= B;
      *dp = pend[cc];  // left end point :101-102
    }
  }
  if (a.lam0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) a.lam0[(size_t)k * B + b] = lam[k];
    a.lam0[(size_t)NS * B + b] = lamc;
  }
  if (warm == 1.234567e300) {  // never true; keeps the table sweep alive
    if (OUT_DJDU) *dp = warm;
    if (OUT_LAM) *lo = warm;
    if (a.lam0) a.lam0[b] = warm;
  }
}

// ---------------------------------------------------------------------------------------
// plugin methods for k columns (used by the API's ocs_problem_F & friends)
// ---------------------------------------------------------------------------------------
template <class P>
__global__ void k_eval(int which, int kcols, const double* __restrict__ t, const double* __restrict__ y,
                       const double* __restrict__ u, const double* __restrict__ v,
                       const double* __restrict__ ps, double* __restrict__ out) {
  constexpr int NS = P::NS, NC = P::NC, NAUG = P::NAUG;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= kcols) return;
  const typename P::Par p = P::load([&](int k) OCS_INLINE { return ps[k]; });
  double tc[P::NTC], tu[P::NTU], yy[NS], uu[NC], vv[NAUG];
  P::tcoef(t[j], ps, tc, tu);
#pragma unroll
  for (int k = 0; k < NS; ++k) yy[k] = y[(size_t)j * NAUG + k];
#pragma unroll
  for (int k = 0; k < NC; ++k) uu[k] = u[(size_t)j * NC + k];
  if (which != 0) {
#pragma unroll
    for (int k = 0; k < NAUG; ++k) vv[k] = v[(size_t)j * NAUG + k];
  }
  if (which == 0) {
    double f[NAUG];
    P::F(tc, yy, uu, p, f);
#pragma unroll
    for (int k = 0; k < NAUG; ++k) out[(size_t)j * NAUG + k] = f[k];
  } else if (which == 1) {
    double g[NS];
    P::dFdxT(tc, yy, uu, p, vv, g);
#pragma unroll
    for (int k = 0; k < NS; ++k) out[(size_t)j * NAUG + k] = g[k];
    out[(size_t)j * NAUG + NS] = 0.0;
  } else {
    double g[NC];
    P::dFduT(tc, yy, uu, p, vv, g);
#pragma unroll
    for (int k = 0; k < NC; ++k) out[(size_t)j * NC + k] = g[k];
  }
}

// ---------------------------------------------------------------------------------------
// layout helpers: 64x64 LDS-tiled transposes between [batch][per] and [per][batch]
// ---------------------------------------------------------------------------------------
// src is R x Cc row-major (R rows of Cc), dst is Cc x R row-major.
__global__ __launch_bounds__(256) void k_transpose(const double* __restrict__ src, double* __restrict__ dst,
                                                   int R, int Cc) {
  __shared__ double tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    if (r < R && c < Cc) tile[rr][tx] = src[(size_t)r * Cc + c];
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, r = r0 + tx;
    if (r < R && c < Cc) dst[(size_t)c * R + r] = tile[tx][cc];
  }
}

// straight 8-byte-per-lane copy: the known-byte-count reference for calibrating the HBM counters
__global__ __launch_bounds__(256) void k_copy8(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
int launch_copy8(const double* src, double* dst, size_t n, hipStream_t s) {
  k_copy8<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(src, dst, n);
  return hip_rc(hipGetLastError());
}

__global__ void k_count_nonfinite(const double* __restrict__ v, int n, int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool bad = (i < n) && !isfinite(v[i]);
  const unsigned long long m = __ballot(bad);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}

// ---------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------
#define OCS_DISPATCH_LOGISTIC(NSV, CALL) \
  switch (NSV) {                         \
    case 1: { using P = LogisticK<1>; CALL; } break; \
    case 2: { using P = LogisticK<2>; CALL; } break; \
    case 3: { using P = LogisticK<3>; CALL; } break; \
    case 4: { using P = LogisticK<4>; CALL; } break; \
    default: return -1;                  \
  }

bool functor_supported(Functor f, int nS, int nC) {
  if (f == Functor::Logistic) return nS >= 1 && nS <= 4 && nC == 1;
  return false;
}
int functor_ntc(Functor f, int nS) {
  (void)nS;
  if (f == Functor::Logistic) return LogisticK<1>::NTC;
  return 0;
}
int functor_ntu(Functor f, int nS) {
  (void)nS;
  if (f == Functor::Logistic) return LogisticK<1>::NTU;
  return 0;
}
unsigned functor_tc_param_mask(Functor f, int nS) {
  (void)nS;
  if (f == Functor::Logistic) return LogisticK<1>::TC_PARAM_MASK;
  return 0;
}

template <class P>
static void run_tcoef(const ProblemDesc& p, const GridDesc& g, hipStream_t s) {
  const int nT = 2 * g.N + 1;
  k_tcoef<P><<<dim3((nT + 255) / 256), dim3(256), 0, s>>>(nT, g.T, p.ps, g.TC, g.TU);
  k_build_rec<P><<<dim3((g.N + 2 * kRecPad + 255) / 256), dim3(256), 0, s>>>(g.N, g.HT, g.TC, g.REC);
}
int launch_tcoef(const ProblemDesc& p, const GridDesc& g, hipStream_t s) {
  OCS_DISPATCH_LOGISTIC(p.nS, run_tcoef<P>(p, g, s));
  return hip_rc(hipGetLastError());
}
int rec_stride_host(int ntc) { return rec_stride(ntc); }
int rec_pad_host() { return kRecPad; }

constexpr int kChunk = 4;
// step records in flight: a divisor of 2*kChunk so the ring needs no register rotation at the loop
// back-edge; deeper for the small problems, whose register budget is loose and which run with few waves
template <class P>
constexpr int pf_of() { return P::NS <= 2 ? 4 : 3; }

template <class P>
static void run_forward(const FwdArgs& a, bool uconst, hipStream_t s) {
  const dim3 grid((a.batch + 63) / 64), block(64);
  if (uconst)
    k_forward<P, kChunk, pf_of<P>(), true, true><<<grid, block, 0, s>>>(a);
  else if (a.x)
    k_forward<P, kChunk, pf_of<P>(), true, false><<<grid, block, 0, s>>>(a);
  else
    k_forward<P, kChunk, pf_of<P>(), false, false><<<grid, block, 0, s>>>(a);
}
// Mapping selection (measured on MI355X, Logistic4, N = 1008; pass pair in us):
//   batch      lane   row-split   pipeline
//    1024       644       322        230
//    4096       672       363        238
//    8192       731       520        446
//   16384       821       960        860
// The wave-specialised pipeline wins while its workgroups (one per 64/nS trajectories, most of a CU's
// LDS each) fit on the chip in at most two rounds; row-split while the lane mapping would leave most
// SIMDs idle; beyond that the lane mapping has the fewest instructions per trajectory and the
// passes turn HBM-bound anyway.
static int choose_mapping(const ProblemDesc& p, int N, int batch, int requested, bool plain, bool backward) {
  if (requested != MAP_AUTO) return requested;
  if (plain && pipeline_supported(p.functor, p.nS, p.nC) && pipeline_shape_ok(p.nS, N, batch, backward) &&
      batch / (64 / p.nS) <= 512)
    return MAP_PIPELINE;
  if (plain && rowsplit_supported(p.functor, p.nS, p.nC) && batch <= 8192) return MAP_ROWSPLIT;
  return MAP_LANE;
}

int launch_forward(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                   double* x, double* J, const FwdOpts& o, hipStream_t s) {
  const bool plain = !o.uconst && !o.Jadd && !o.usel;
  const int map = choose_mapping(p, g.N, batch, o.mapping, plain, false);
  if (map == MAP_PIPELINE) {
    if (!plain || !pipeline_supported(p.functor, p.nS, p.nC)) return -1;
    return launch_forward_pl(p, g, batch, x0, u, x, J, s);
  }
  if (map == MAP_ROWSPLIT) {
    if (!plain || !rowsplit_supported(p.functor, p.nS, p.nC)) return -1;
    return launch_forward_rs(p, g, batch, x0, u, x, J, s);
  }
  if (o.uconst && !x) return -1;
  const FwdArgs a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, x0, u, x, J, o.Jadd, o.usel, o.udelta};
  OCS_DISPATCH_LOGISTIC(p.nS, run_forward<P>(a, o.uconst, s));
  return hip_rc(hipGetLastError());
}

template <class P>
static void run_backward(const BwdArgs& a, bool uconst, hipStream_t s) {
  const dim3 grid((a.batch + 63) / 64), block(64);
  if (uconst)
    k_backward<P, kChunk, pf_of<P>(), false, false, true><<<grid, block, 0, s>>>(a);
  else if (a.lam && a.dJdu)
    k_backward<P, kChunk, pf_of<P>(), true, true, false><<<grid, block, 0, s>>>(a);
  else if (a.lam)
    k_backward<P, kChunk, pf_of<P>(), true, false, false><<<grid, block, 0, s>>>(a);
  else
    k_backward<P, kChunk, pf_of<P>(), false, true, false><<<grid, block, 0, s>>>(a);
}
int launch_backward(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                    const double* lamT, double* lam, double* dJdu, const BwdOpts& o, hipStream_t s) {
  const bool plain = !o.uconst && !o.usel;
  const int map = choose_mapping(p, g.N, batch, o.mapping, plain, true);
  if (map == MAP_PIPELINE) {
    if (!plain || !pipeline_supported(p.functor, p.nS, p.nC)) return -1;
    return launch_backward_pl(p, g, batch, xck, u, lamT, lam, dJdu, o.lam0, s);
  }
  if (map == MAP_ROWSPLIT) {
    if (!plain || !rowsplit_supported(p.functor, p.nS, p.nC)) return -1;
    return launch_backward_rs(p, g, batch, xck, u, lamT, lam, dJdu, o.lam0, s);
  }
  if (o.uconst ? (lam || dJdu || !o.lam0) : (!lam && !dJdu)) return -1;
  const BwdArgs a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, xck, u, lamT, lam, dJdu, o.lam0, o.usel, o.udelta};
  OCS_DISPATCH_LOGISTIC(p.nS, run_backward<P>(a, o.uconst, s));
  return hip_rc(hipGetLastError());
}

template <class P>
static void run_eval(const ProblemDesc& p, int which, int k, const double* t, const double* y, const double* u,
                     const double* v, double* out, hipStream_t s) {
  k_eval<P><<<dim3((k + 127) / 128), dim3(128), 0, s>>>(which, k, t, y, u, v, p.ps, out);
}
int launch_eval(const ProblemDesc& p, int which, int k, const double* t, const double* y, const double* u,
                const double* v, double* out, hipStream_t s) {
  OCS_DISPATCH_LOGISTIC(p.nS, run_eval<P>(p, which, k, t, y, u, v, out, s));
  return hip_rc(hipGetLastError());
}

// traj-major [batch][per]  ->  batch-minor [per][batch]
int launch_to_batch_minor(const double* src, double* dst, int per, int batch, hipStream_t s) {
  hipLaunchKernelGGL(k_transpose, dim3((per + 63) / 64, (batch + 63) / 64), dim3(256), 0, s, src, dst, batch,
                     per);
  return hip_rc(hipGetLastError());
}
// batch-minor [per][batch]  ->  traj-major [batch][per]
int launch_to_traj_major(const double* src, double* dst, int per, int batch, hipStream_t s) {
  hipLaunchKernelGGL(k_transpose, dim3((batch + 63) / 64, (per + 63) / 64), dim3(256), 0, s, src, dst, per,
                     batch);
  return hip_rc(hipGetLastError());
}

int launch_count_nonfinite(const double* v, int n, int* count, hipStream_t s) {
  hipLaunchKernelGGL(k_count_nonfinite, dim3((n + 255) / 256), dim3(256), 0, s, v, n, count);
  return hip_rc(hipGetLastError());
}

}  // namespace ocs
