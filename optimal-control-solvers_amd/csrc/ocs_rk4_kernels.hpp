// ocs_rk4_kernels.hpp -- kernel templates of the lane-per-trajectory RK4 passes (time-coefficient /
// record tables, forward, adjoint, plugin evaluation).  Included by ocs_kernels.hip for the built-in
// problems and handed to hipRTC for user-supplied problems (ocs_jit.cpp), so it must stay free of host
// code and of includes other than ocs_device_common.hpp.
#pragma once
#include "ocs_device_common.hpp"

namespace ocs {

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
  const int* frozen;    // optional [B]: a trajectory with frozen[b] != 0 is integrated but stores nothing (fb_sweep:
  double* dump;         //   a converged instance keeps the x, J of the sweep it converged in); its stores go to dump[b]
  const double* yc0;    // optional [B]: running objective at the first node (a pass continued from another
                        //               kernel's last column; default 0, RK4Integrator.m:33)
  int ld;               // distance between rows of the batch-minor arrays when it is not `batch` (a launch on a
                        // window of a larger batch: pointers are offset, `batch` counts the window); 0 = batch
  const int* gate = nullptr;   // optional: the launch does nothing if *gate == 0 (a sweep of fb_sweep enqueued before the
                               // host knew that the sweep before it had left no instance active)
};

// Lanes past the end of the batch are clamped onto the last trajectory: they recompute it and
// store the same values to the same addresses, which keeps the step body free of exec-mask
// branches (one basic block per chunk, so the scheduler can hoist the prefetch loads).
// stores of the per-step outputs (x, lam, dJdu): non-temporal in the instances that serve launches beyond the memory-side cache
// (NT: batch >= 32768, where the next pass cannot find them there anyway; pair at batch 65536 1704 -> 1652 us)
template <bool NT>
__device__ static inline void lane_store(double* ptr, double val) {
  if (NT) __builtin_nontemporal_store(val, ptr);
  else *ptr = val;
}
#define OCS_LANE_ST(ptr, val) lane_store<NT>((ptr), (val))
template <class P, int CH, int PF, bool OUT_X, bool UCONST, bool NT = false>
__global__ __launch_bounds__(64) void k_forward(const FwdArgs a) {
  if (a.gate && *a.gate == 0) return;
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC;
  using Rec = StepRec<NTC>;
  const int b0 = blockIdx.x * 64 + threadIdx.x;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)(a.ld ? a.ld : a.batch);
  const int N = a.N;
  const uniform_ptr PS = as_uniform(a.ps);
  const double* REC = a.REC;

  const typename P::Par p = P::load(ParamSrc{PS, a.pb, a.pmask, B, b});

  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));

  double y[NS], yc = a.yc0 ? a.yc0[b] : 0.0;  // xK(:,1,1) = [x0; 0]   :33
#pragma unroll
  for (int k = 0; k < NS; ++k) y[k] = a.x0[(size_t)k * B + b];
  // All per-trajectory arrays are walked with one pointer that advances by B doubles per row:
  // rows of consecutive time points are B apart, so no per-row offset arithmetic is needed.
  // frozen lanes write every row to one scratch double (pointer stride 0): no branch around the stores
  const bool fz = a.frozen && a.frozen[b] != 0;
  const size_t xstep = fz ? 0 : B;
  double* xo = fz ? a.dump + b : a.x + b;
  if (OUT_X) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      *xo = y[k];
      xo += xstep;
    }
    *xo = yc;
    xo += xstep;
  }

  const double* up = a.u;  // walks u(:,j) row by row
  double uprev[NC];
  if (UCONST) {
#pragma unroll
    for (int c = 0; c < NC; ++c) uprev[c] = PSU(a.u)[c];
  } else {
    up += b;
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
        OCS_LANE_ST(xo, y[k]);
        xo += xstep;
      }
      OCS_LANE_ST(xo, yc);
      xo += xstep;
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
    if (!fz) a.J[b] = a.Jadd ? a.Jadd[b] + yc : yc;
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
  if (!fz) a.J[b] = a.Jadd ? a.Jadd[b] + yc : yc;  // J = x(end,end)   :55
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
};

// XRC (checkpoint re-integration, as in the scan kernel, ocs_scan_kernel.hpp): of the CH checkpoints of a chunk only the first is
// read; the others are integrated forward from it with the state pass's own operations (the control samples and step records of
// the chunk are at hand anyway).  +3 evaluations of the state rows per step against 3/4 of the checkpoint reads: for launches
// that are HBM-bound (the chip full of waves), not for the latency-bound ones.  Needs PF >= CH (the ring holds the chunk's records).
template <class P, int CH, int PF, bool OUT_LAM, bool OUT_DJDU, bool UCONST, bool XRC = false>
__global__ __launch_bounds__(64) void k_backward(const BwdArgs a) {
  constexpr bool NT = XRC;   // (the same launches: HBM-bound, outputs larger than the memory-side cache)
  static_assert(!XRC || (PF >= CH && !UCONST), "re-integration: the record ring must hold a chunk");
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC, NAUG = P::NAUG;
  using Rec = StepRec<NTC>;
  const int b0 = blockIdx.x * 64 + threadIdx.x;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  const uniform_ptr PS = as_uniform(a.ps);
  const double* REC = a.REC;

  const typename P::Par p = P::load(ParamSrc{PS, a.pb, a.pmask, B, b});

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
        OCS_LANE_ST(dp, pend[c] + d4[c]);  // column 2i+2  :112-116 (:119-120 at i = N-1)
      }
#pragma unroll
      for (int c = NC - 1; c >= 0; --c) {
        dp -= B;
        OCS_LANE_ST(dp, d2[c] + d3[c]);    // column 2i+1  :105-109
      }
      P::dFduT(r.tcA, xi, uA, p, k1, pend);
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) lam[k] = (((lam[k] + g1[k]) + g2[k]) + g3[k]) + g0[k];  // :86-88
    if (OUT_LAM) {
      lo -= B;
      OCS_LANE_ST(lo, lamc);
#pragma unroll
      for (int k = NS - 1; k >= 0; --k) {
        lo -= B;
        OCS_LANE_ST(lo, lam[k]);
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
      if (XRC && s > 0) {   // integrated from checkpoint 0 of the chunk in run_chunk
        xp -= (size_t)(NS + 1) * B;
        continue;
      }
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
  auto run_chunk = [&](const double (&xs0)[CH][NS], const double (&us)[2 * CH][NC]) OCS_INLINE {
    double xs[CH][NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) xs[0][k] = xs0[0][k];
#pragma unroll
    for (int s = 1; s < CH; ++s) {
      if (XRC) {   // x(:, i0 + s) from x(:, i0 + s - 1): compute_states :39-51 on the state rows, the state pass's operations
        const Rec& r = rq[CH - s];   // the ring holds the records of steps i0 + CH - 1 (rq[0]) .. i0 (rq[CH - 1])
        const double *y = xs[s - 1], *uA = us[2 * s - 2], *uM = us[2 * s - 1], *uB = us[2 * s];
        double F1[NS], F2[NS], F3[NS], F4[NS], Y[NS];
        P::Fx(r.tcA, y, uA, p, F1);
#pragma unroll
        for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.hh, F1[k], y[k]);
        P::Fx(r.tcM, Y, uM, p, F2);
#pragma unroll
        for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.hh, F2[k], y[k]);
        P::Fx(r.tcM, Y, uM, p, F3);
#pragma unroll
        for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.h, F3[k], y[k]);
        P::Fx(r.tcB, Y, uB, p, F4);
#pragma unroll
        for (int k = 0; k < NS; ++k)
          xs[s][k] = __builtin_fma(r.h6, __builtin_fma(2.0, F3[k], __builtin_fma(2.0, F2[k], F1[k])) + F4[k], y[k]);
      } else {
#pragma unroll
        for (int k = 0; k < NS; ++k) xs[s][k] = xs0[s][k];
      }
    }
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
                       const double* __restrict__ ps, double* __restrict__ out, const double* __restrict__ lb,
                       const double* __restrict__ ub) {
  constexpr int NS = P::NS, NC = P::NC, NAUG = P::NAUG;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= kcols) return;
  const typename P::Par p = P::load(ParamSrc{as_uniform(ps), nullptr, 0u, 0, 0});
  double tc[P::NTC], tu[P::NTU], yy[NS], uu[NC], vv[NAUG];
  P::tcoef(t[j], ps, tc, tu);
  if (which == 3) {  // ControlChar(t, x, lam) with the clamp (make_from_symbolic.m:33-38, 111): y = x, v = lam, nS rows each
    double xx[NS], ll[NS], lo[NC], hi[NC], uo[NC];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      xx[k] = y[(size_t)j * NS + k];
      ll[k] = v[(size_t)j * NS + k];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      lo[c] = lb[c];
      hi[c] = ub[c];
    }
    P::control_char(tu, xx, ll, p, lo, hi, uo);
#pragma unroll
    for (int c = 0; c < NC; ++c) out[(size_t)j * NC + c] = uo[c];
    return;
  }
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
// compute_equilibrium (functions/compute_equilibrium.m:10-27), batched: one thread per instance solves the
// (2 nS + nC)-dimensional steady-state system of the optimality conditions
//     F(0, x, u)(1:nS) = 0,   r lam - dFdx_times_vec(0, x, u, [lam; 1])(1:nS) = 0,   dFdu_times_vec(0, x, u, [lam; 1]) = 0
// inside the box lb <= y <= ub, y = [x; lam; u], with a projected Levenberg-Marquardt iteration.  The reference
// hands the same residual to lsqnonlin (a MATLAB toolbox; trust-region-reflective with a finite-difference
// Jacobian); here the Jacobian is a central difference of the plugin methods as well, so any OCProblem works
// unchanged, and since the residual itself is exact the iteration still converges to round-off.
// ---------------------------------------------------------------------------------------
struct EqArgs {
  int batch;
  double r;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* y0;      // [n][B] initial guesses (clamped to the bounds, compute_equilibrium.m:26-27 via lsqnonlin)
  const double* lb;      // [n]
  const double* ub;      // [n]
  double* y;             // [n][B]
  double* resnorm;       // [B] squared 2-norm of the residual
  double* residual;      // [n][B] or nullptr
  int* exitflag;         // [B]: 1 converged (residual at round-off or no further decrease possible), 0 iteration limit,
                         //      -1 the residual is not finite at the (clamped) guess or became so (lsqnonlin raises an
                         //      error on undefined values, compute_equilibrium.m:27: never reported as a solution)
  int max_iter;
  double tol;            // stop when the max-norm of the residual is below tol (relative to max(1, |terms|))
};

template <class P>
__global__ __launch_bounds__(64) void k_equilibrium(const EqArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NAUG = P::NAUG, NV = 2 * NS + NC;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  const size_t B = (size_t)a.batch;
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  double tc[P::NTC], tu[P::NTU];
  P::tcoef(0.0, a.ps, tc, tu);
  auto residual = [&](const double* yv, double* R) {
    double xx[NS], uu[NC], vv[NAUG], f[NAUG], g[NS], gu[NC];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      xx[k] = yv[k];
      vv[k] = yv[NS + k];
    }
    vv[NS] = 1.0;
#pragma unroll
    for (int k = 0; k < NC; ++k) uu[k] = yv[2 * NS + k];
    P::F(tc, xx, uu, p, f);                      // :16-17
    P::dFdxT(tc, xx, uu, p, vv, g);              // :19-20
    P::dFduT(tc, xx, uu, p, vv, gu);             // :22
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      R[k] = f[k];
      R[NS + k] = a.r * vv[k] - g[k];
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) R[2 * NS + k] = gu[k];
  };
  double y[NV], lo[NV], hi[NV], R[NV];
  for (int k = 0; k < NV; ++k) {
    lo[k] = a.lb[k];
    hi[k] = a.ub[k];
    const double g0 = a.y0[(size_t)k * B + b];
    y[k] = g0 != g0 ? g0 : fmin(hi[k], fmax(lo[k], g0));   // (fmin / fmax would replace a NaN guess by a bound)
  }
  residual(y, R);
  double cost = 0.0;
  for (int k = 0; k < NV; ++k) cost += R[k] * R[k];
  for (int k = 0; k < NV; ++k) cost += (y[k] != y[k]) ? y[k] : 0.0;   // a NaN guess: not a point to start from
  double mu = 1e-3;
  int flag = 0;
  for (int it = 0; it < a.max_iter; ++it) {
    if (!isfinite(cost)) {   // (fmax below would drop a NaN and report convergence)
      flag = -1;
      break;
    }
    double rmax = 0.0;
    for (int k = 0; k < NV; ++k) rmax = fmax(rmax, fabs(R[k]));
    if (rmax <= a.tol) {
      flag = 1;
      break;
    }
    // Jacobian by central differences, one column at a time (kept inside the box: one-sided at an active bound)
    double Jm[NV][NV];
    for (int j = 0; j < NV; ++j) {
      const double hstep = 6.0554544523933395e-06 * fmax(1.0, fabs(y[j]));   // eps^(1/3)
      double yp[NV], ym[NV], Rp[NV], Rm[NV];
      for (int k = 0; k < NV; ++k) yp[k] = ym[k] = y[k];
      yp[j] = fmin(hi[j], y[j] + hstep);
      ym[j] = fmax(lo[j], y[j] - hstep);
      if (!(yp[j] > ym[j])) {   // a variable fixed by lb == ub: no column (it is frozen below)
        for (int k = 0; k < NV; ++k) Jm[k][j] = 0.0;
        continue;
      }
      residual(yp, Rp);
      residual(ym, Rm);
      const double inv = 1.0 / (yp[j] - ym[j]);
      for (int k = 0; k < NV; ++k) Jm[k][j] = (Rp[k] - Rm[k]) * inv;
    }
    // normal equations (J'J + mu diag(J'J)) d = -J'R, solved by Gaussian elimination with partial pivoting;
    // variables sitting on a bound with the step pointing outward are frozen (projected step)
    double Am[NV][NV], gv[NV];
    for (int i = 0; i < NV; ++i) {
      double s = 0.0;
      for (int k = 0; k < NV; ++k) s += Jm[k][i] * R[k];
      gv[i] = s;
      for (int j = 0; j < NV; ++j) {
        double t = 0.0;
        for (int k = 0; k < NV; ++k) t += Jm[k][i] * Jm[k][j];
        Am[i][j] = t;
      }
    }
    bool improved = false;
    for (int tries = 0; tries < 12 && !improved; ++tries) {
      double M[NV][NV + 1];
      for (int i = 0; i < NV; ++i) {
        const bool frozen = (y[i] <= lo[i] && gv[i] > 0.0) || (y[i] >= hi[i] && gv[i] < 0.0) || !(hi[i] > lo[i]);
        for (int j = 0; j < NV; ++j) M[i][j] = frozen ? (i == j ? 1.0 : 0.0) : Am[i][j];
        if (!frozen) M[i][i] += mu * fmax(Am[i][i], 1e-300);
        M[i][NV] = frozen ? 0.0 : -gv[i];
      }
      for (int c = 0; c < NV; ++c) {
        int piv = c;
        for (int i = c + 1; i < NV; ++i)
          if (fabs(M[i][c]) > fabs(M[piv][c])) piv = i;
        for (int j = 0; j <= NV; ++j) {
          const double t = M[c][j];
          M[c][j] = M[piv][j];
          M[piv][j] = t;
        }
        const double d = M[c][c];
        const double inv = d != 0.0 ? 1.0 / d : 0.0;
        for (int i = c + 1; i < NV; ++i) {
          const double fct = M[i][c] * inv;
          for (int j = c; j <= NV; ++j) M[i][j] -= fct * M[c][j];
        }
      }
      double dlt[NV];
      for (int i = NV - 1; i >= 0; --i) {
        double s = M[i][NV];
        for (int j = i + 1; j < NV; ++j) s -= M[i][j] * dlt[j];
        dlt[i] = M[i][i] != 0.0 ? s / M[i][i] : 0.0;
      }
      double yn[NV], Rn[NV], cn = 0.0;
      for (int k = 0; k < NV; ++k) yn[k] = fmin(hi[k], fmax(lo[k], y[k] + dlt[k]));
      residual(yn, Rn);
      for (int k = 0; k < NV; ++k) cn += Rn[k] * Rn[k];
      if (cn < cost) {
        for (int k = 0; k < NV; ++k) {
          y[k] = yn[k];
          R[k] = Rn[k];
        }
        cost = cn;
        mu = fmax(mu * (1.0 / 3.0), 1e-15);
        improved = true;
      } else {
        mu *= 4.0;
      }
    }
    if (!improved) {   // no decrease at any damping: at round-off level of the residual, or at a constrained minimum
      flag = isfinite(cost) ? 1 : -1;
      break;
    }
  }
  if (!isfinite(cost)) flag = -1;
  for (int k = 0; k < NV; ++k) {
    a.y[(size_t)k * B + b] = y[k];
    if (a.residual) a.residual[(size_t)k * B + b] = R[k];
  }
  a.resnorm[b] = cost;
  a.exitflag[b] = flag;
}


}  // namespace ocs
