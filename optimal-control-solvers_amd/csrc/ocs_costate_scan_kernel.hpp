// ocs_costate_scan_kernel.hpp -- the costate pass of the forward-backward sweep (functions/compute_x_lam.m:11-14,
// lam' = adjointRHS(t, x(t), lam, u(t)), lam(TF) = 0, integrated backwards; here RK4 on the node grid with x(t) the
// pchip interpolant of the node values) as a scan over time, for row-separable problems whose adjoint
// right-hand side does not read u: the registry's logistic family and hipRTC problems given as row functions that declare
// it (OCS_USER_CC_NOX, ocs_user_functor.hpp).  UR: the same scan for ANY problem given as row functions -- the samples of
// the control on the grid are read next to the state (8 more bytes per instance and step), dF_r/dy_r and dq_r/dy_r are
// evaluated with them; the control update stays a kernel of its own (k_control_grid).
//
// adjointRHS = -dFdx_times_vec(t, [x; 0], u, [lam; 1])(1:nS) (the A9 adapter) is affine in lam, so an RK4
// step from t_{i+1} down to t_i is an affine map per row, lam_i = alpha_i lam_{i+1} + beta_i, whose coefficients depend on
// x at the two nodes and at the middle of the interval only.  As in k_backward_scan (ocs_scan_kernel.hpp):
//
//   phase 1  (time-parallel)  wave w takes a chunk of L consecutive intervals of the workgroup's 64 (row, instance) lanes:
//            the pchip slopes of x at the chunk's nodes (Fritsch-Carlson, the formulas of k_costate_plx), x at the middle
//            of every interval, (alpha_i, beta_i), and their composition, the chunk map;
//   phase 2  the W chunk maps of a superblock go through LDS; every wave composes the maps above its own on top of the
//            carry;
//   phase 3  lam_i = alpha_i lam_{i+1} + beta_i down the chunk (the step maps are still in registers), the stores of lam --
//            and, MET, the weighted change of the control this costate implies against the one the costate of the sweep
//            before implied, ControlChar(lam_new) against ControlChar(lam_old) at the nodes (fb_sweep.m:107), as a running
//            fraction per instance.  After the last superblock the partial maxima of the W waves meet in LDS and wave 0
//            does check_convergence and the loop bookkeeping (fb_sweep.m:79-87, :99-115) for its instances.
//
// The serial kernel (k_costate_plx) runs one recursion wave per 64 lanes through 1000 dependent steps (~175 cycles each);
// here no chain is longer than a chunk.  Results: lam as composed affine maps -- another association of the same sums
// than the step-by-step recursion (round-off level; the sweep's tolerance against the oracle is 1e-10).
#pragma once
#include "ocs_device_common.hpp"
#include "ocs_scan_kernel.hpp"

namespace ocs {

struct CostateScanArgs {
  int N, batch;            // N: a multiple of 8
  const double* RECS;      // scan records of the grid (ocs_scan_kernel.hpp): {h, h/2, h/6, .. | .. | tcA, tcM, tcB}
  const double* PR;        // [N][kPRec] pchip interval records (ocs_device_common.hpp)
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* x;         // [N+1][ldx][B] node values of the state
  int ldx;
  const int* frozen;       // optional [B]: instances that store nothing (MET: == status)
  double* lam;             // [N+1][G][B]; MET: holds the costate of the sweep before on entry
  const int* gate;         // optional: the launch does nothing if *gate == 0
  // MET
  const double* TU;        // [2N+1 (+128 readable)] ControlChar-side time coefficients
  const double* lb;
  const double* ub;
  double relTol, absTol;
  int sweep;               // sweep 1: the control before it is the lower bound (u0, fb_sweep.m:23)
  int* status;
  double* maxChange;       // [nSWEEPS][B]
  int* nactive;
  // UR
  const double* u;         // [2N+1][B] samples of the control on the grid (nC = 1): problems whose adjoint right-hand side reads u
};

template <class P, int W, int L, bool MET, bool UR = false>
__global__ __launch_bounds__(W * 64) void k_costate_scan(const CostateScanArgs a) {
  constexpr int G = P::NS, TPW = 64 / G;
  static_assert(P::NC == 1 && P::NTC == 1 && P::ROW_SEPARABLE && L == 4 && W * L + 1 <= kScanPadFront, "chunk shape");
  static_assert(!(MET && UR), "the convergence test inside the pass is for problems whose control follows from the costate alone");
  __shared__ __attribute__((aligned(16))) double2 sm[2][W][64];               // chunk maps
  __shared__ double csm[2][64];                                                // lam at the bottom of a superblock
  __shared__ __attribute__((aligned(16))) double tab[2][W][2][128];            // per wave: records | interval records
  __shared__ double xres[W][3][64];                                            // MET: partial maxima of the waves
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane / TPW, tl = lane % TPW;
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  if (a.gate && *a.gate == 0) return;
  const int b = tile_base(blockIdx.x, TPW, a.batch) + tl;   // (a ragged last tile overlaps its neighbour)
  const uniform_ptr PS = as_uniform(a.ps);
  const typename P::RowPar rp = P::load_row(ParamSrc{PS, a.pb, a.pmask, B, b}, r);
  const bool fz = a.frozen != nullptr && a.frozen[b] != 0;
  const size_t colB = (size_t)G * B, xcolB = (size_t)a.ldx * B;
  const unsigned col8 = (unsigned)(colB * 8);
  const unsigned vrow = (unsigned)(((size_t)r * B + b) * 8);
  const unsigned vst = fz ? kOffDrop : vrow;
  if (!fz && wave == 0) a.lam[(size_t)N * colB + (size_t)r * B + b] = 0.0;   // lam(TF) = 0   compute_x_lam.m:4

  // MET: the largest weighted change as a fraction nmax / dmax (k_costate_plx's bookkeeping)
  double nmax = 0.0, dmax = 1.0;
  bool any = false;
  auto take = [&](double un, double uo) OCS_INLINE {
    const double n = fabs(un - uo), d = a.relTol * fabs(uo) + a.absTol;
    const bool valid = (n + d) > 0.0;
    const bool rep = valid & (!any | (n * dmax > nmax * d));
    nmax = rep ? n : nmax;
    dmax = rep ? d : dmax;
    any = any | valid;
  };
  typename P::CCPre ccp{};
  double lbv = 0.0, ubv = 0.0;
  const bool u0lb = MET && a.sweep == 1;
  if (MET) {
    ccp = P::cc_pre(P::load(ParamSrc{PS, a.pb, a.pmask, B, b}));
    lbv = a.lb[0];
    ubv = a.ub[0];
    if (wave == 0) {   // node t_N: lam = 0 before and after the pass
      double z[G];
#pragma unroll
      for (int q = 0; q < G; ++q) z[q] = 0.0;
      const double uN = P::control_char_pre(a.TU[(size_t)2 * N], z, ccp, lbv, ubv);
      take(uN, u0lb ? lbv : uN);
    }
  }

  struct Ld {
    double w[L + 3];    // x(r, lo-1 .. lo+L+1), clamped to the grid
    double lo_[L];      // MET: the costate of the sweep before at nodes lo .. lo+L-1
    double tu[L];       // MET: ControlChar-side time coefficients of those nodes (wave-uniform: scalar loads)
    double u[UR ? 2 * L + 1 : 1];   // UR: the control at grid points 2 lo .. 2 (lo + L)
  };
  const uniform_ptr TUu = as_uniform(a.TU);

  auto chunk_lo = [&](int sb) OCS_INLINE { return N - (sb * W + wave + 1) * L; };
  auto load = [&](int sb, Ld& d, int slot) OCS_INLINE {
    const int lo = chunk_lo(sb);
    const int lr = lo >= 0 ? lo : -kScanPadFront;                       // records lo .. lo+7 (zero records below step 0)
    const int lp = lo < 0 ? 0 : (lo > N - 8 ? N - 8 : lo);              // interval records lp .. lp+7, inside the table
    dma16_sc(a.RECS + (long long)lr * kScanRec + 2 * lane, &tab[slot][wave][0][0]);
    dma16_sc(a.PR + (size_t)lp * kPRec + 2 * lane, &tab[slot][wave][1][0]);
#pragma unroll
    for (int t = 0; t < L + 3; ++t) {
      int i = lo - 1 + t;
      i = i < 0 ? 0 : (i > N ? N : i);
      d.w[t] = a.x[(size_t)i * xcolB + (size_t)r * B + b];
    }
    if (MET) {
#pragma unroll
      for (int q = 0; q < L; ++q) {
        const int i = lo + q < 0 ? 0 : lo + q;
        d.lo_[q] = a.lam[(size_t)i * colB + (size_t)r * B + b];
        d.tu[q] = TUu[2 * i];
      }
    }
    if (UR) {
#pragma unroll
      for (int t = 0; t < 2 * L + 1; ++t) {
        int j = 2 * lo + t;
        j = j < 0 ? 0 : (j > 2 * N ? 2 * N : j);
        d.u[t] = a.u[(size_t)j * B + b];
      }
    }
  };

  double carry = 0.0;
  // (Loads one superblock ahead.  A superblock is only a few hundred instructions per wave, less than a round trip to
  //  memory, so part of the latency stays exposed at small batch; a lead of two superblocks was tried: the compiler's own
  //  counter bookkeeping then waits for everything at the loop's merge points and the pass got slower, 2.07 against
  //  2.00 ms per solve at batch 2048.)
  auto process = [&](int sb, const Ld& d, int slot, Ld& dn) OCS_INLINE {
    const int lo = chunk_lo(sb);
    const bool live = lo >= 0;
    // the loads and the two tables of this superblock have landed once everything but the L stores of the superblock
    // before (issued behind them) has
    if (sb == 0)
      __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
    else
      __builtin_amdgcn_s_waitcnt(0x0F70 | L);   // vmcnt(L)
    asm volatile("" ::: "memory");
    const double* recs = &tab[slot][wave][0][0];
    const int lp = lo < 0 ? 0 : (lo > N - 8 ? N - 8 : lo);
    const double* prs = &tab[slot][wave][1][0] + (size_t)((lo < 0 ? 0 : lo) - lp) * kPRec;
    // ---------------- phase 1: pchip midpoints of x, the step maps, the chunk map ----------------
    double sec[L + 2];   // secants of intervals lo-1 .. lo+L
#pragma unroll
    for (int t = 0; t < L + 2; ++t) {
      // reciprocal spacing of interval lo-1+t: from the record of a neighbouring interval of the chunk
      const double ih = t == 0 ? prs[3] : (t == L + 1 ? prs[(L - 1) * kPRec + 5] : prs[(t - 1) * kPRec + 4]);
      sec[t] = (d.w[t + 1] - d.w[t]) * ih;
    }
    double dsl[L + 1];   // slopes at nodes lo .. lo+L
#pragma unroll
    for (int c = 0; c < L + 1; ++c)
      dsl[c] = pchip_interior_f(sec[c], sec[c + 1], c < L ? prs[c * kPRec + 6] : prs[(L - 1) * kPRec + 8],
                                c < L ? prs[c * kPRec + 7] : prs[(L - 1) * kPRec + 9]);
    if (lo == 0) dsl[0] = pchip_end_pl(prs[1], prs[2], sec[1], sec[2]);
    if (lo + L == N) dsl[L] = pchip_end_pl(prs[(L - 1) * kPRec + 1], prs[(L - 1) * kPRec + 0], sec[L], sec[L - 1]);
    double al[L], be[L];
#pragma unroll
    for (int q = 0; q < L; ++q) {
      const double* rc = recs + q * kScanRec;
      const double h = rc[0], hh = rc[1], h6 = rc[2];
      const double eA = rc[8], eM = rc[9], eB = rc[10];   // time coefficient at the left node, the middle, the right node
      const double xA = d.w[q + 1], xB = d.w[q + 2];
      const double xM = __builtin_fma(prs[q * kPRec + 11], dsl[q] - dsl[q + 1], 0.5 * (xA + xB));
      double aA, bA, aM, bM, aB, bB;
      if constexpr (UR) {
        P::costate_row_pre_u(xA, d.u[2 * q], eA, rp, aA, bA);
        P::costate_row_pre_u(xM, d.u[2 * q + 1], eM, rp, aM, bM);
        P::costate_row_pre_u(xB, d.u[2 * q + 2], eB, rp, aB, bB);
      } else {
        P::costate_row_pre(xA, eA, rp, aA, bA);
        P::costate_row_pre(xM, eM, rp, aM, bM);
        P::costate_row_pre(xB, eB, rp, aB, bB);
      }
      // the RK4 step of k_costate_plx on the pair (coefficient of lam_{i+1}, constant):
      //   k1 = -(aB l + bB);  L = l - hh k1;  k2 = -(aM L + bM);  L = l - hh k2;  k3 = -(aM L + bM);  L = l - h k3;
      //   k4 = -(aA L + bA);  l <- l - h6 (k1 + 2 k2 + 2 k3 + k4)
      const double k1p = -aB, k1q = -bB;
      const double p2 = __builtin_fma(-hh, k1p, 1.0), q2 = -hh * k1q;
      const double k2p = -aM * p2, k2q = -__builtin_fma(aM, q2, bM);
      const double p3 = __builtin_fma(-hh, k2p, 1.0), q3 = -hh * k2q;
      const double k3p = -aM * p3, k3q = -__builtin_fma(aM, q3, bM);
      const double p4 = __builtin_fma(-h, k3p, 1.0), q4 = -h * k3q;
      const double k4p = -aA * p4, k4q = -__builtin_fma(aA, q4, bA);
      al[q] = __builtin_fma(-h6, k4p, __builtin_fma(-h6, __builtin_fma(2.0, k3p, __builtin_fma(2.0, k2p, k1p)), 1.0));
      be[q] = __builtin_fma(-h6, k4q, -h6 * __builtin_fma(2.0, k3q, __builtin_fma(2.0, k2q, k1q)));
    }
    double A = al[L - 1], Bq = be[L - 1];
#pragma unroll
    for (int q = L - 2; q >= 0; --q) {
      Bq = __builtin_fma(al[q], Bq, be[q]);
      A = al[q] * A;
    }
    sm[sb & 1][wave][lane] = double2{A, Bq};
    load(sb + 1, dn, slot ^ 1);
    lds_barrier_sc();
    // ---------------- phase 2: lam at the top of this chunk ----------------
    double lam = (sb == 0) ? 0.0 : csm[(sb & 1) ^ 1][lane];
#pragma unroll
    for (int j0 = 0; j0 < W; j0 += 4) {
      if (j0 < wave) {   // wave-uniform
        double2 ab[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ab[j] = sm[sb & 1][j0 + j][lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j0 + j < wave) {
            lam = __builtin_fma(ab[j].x, lam, ab[j].y);
            asm volatile("" : "+v"(lam));
          }
        }
      }
    }
    // ---------------- phase 3: the costate at the chunk's nodes, its stores, the change of the control ----------------
    const int lc = live ? lo : 0;
    const Buf bl = Buf::make(a.lam + (size_t)lc * colB, live ? kNumRec : 0);
#pragma unroll
    for (int q = L - 1; q >= 0; --q) {
      lam = __builtin_fma(al[q], lam, be[q]);
      bl.st(lam, vst, (unsigned)q * col8);
      if (MET) {
        // ControlChar reads the costate of all rows: the sum over the rows of an instance, in every lane of the instance
        double ln[G], lo_[G];
        if constexpr (P::HAS_SHIFT) {   // registry rows: ControlChar reads the sum of the costate rows only
          ln[0] = group_sum_sc<G>(lam);
          lo_[0] = group_sum_sc<G>(d.lo_[q]);
#pragma unroll
          for (int k = 1; k < G; ++k) ln[k] = lo_[k] = 0.0;
        } else {                        // any ControlChar: every lane of an instance gets the costate of all its rows
#pragma unroll
          for (int k = 0; k < G; ++k) {
            ln[k] = __shfl(lam, k * TPW + tl);
            lo_[k] = __shfl(d.lo_[q], k * TPW + tl);
          }
        }
        const double tu = d.tu[q];
        const double un = P::control_char_pre(tu, ln, ccp, lbv, ubv), uo = P::control_char_pre(tu, lo_, ccp, lbv, ubv);
        if (live) take(un, u0lb ? lbv : uo);   // (wave-uniform)
      }
    }
    if (wave == W - 1) csm[sb & 1][lane] = lam;
    carry = lam;
  };

  const int nsb = (N + W * L - 1) / (W * L);
  Ld d0, d1;
  load(0, d0, 0);
  for (int sb = 0; sb < nsb; sb += 2) {   // (a superblock past the horizon: dead chunks, identity maps, no stores)
    process(sb, d0, 0, d1);
    process(sb + 1, d1, 1, d0);
  }
  (void)carry;
  if (!MET) return;
  // ---------------- check_convergence for the workgroup's instances   fb_sweep.m:99-115, :79-87 ----------------
  if (wave > 0) {
    xres[wave][0][lane] = nmax;
    xres[wave][1][lane] = dmax;
    xres[wave][2][lane] = any ? 1.0 : 0.0;
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 1; w < W; ++w) {
    const double on = xres[w][0][lane], od = xres[w][1][lane];
    const bool oa = xres[w][2][lane] != 0.0;
    if (oa && (!any || on * dmax > nmax * od)) {
      nmax = on;
      dmax = od;
    }
    any = any || oa;
  }
  bool still = false;
  if (r == 0 && a.status[b] == 0) {
    const double mx = any ? nmax / dmax : __builtin_nan("");
    a.maxChange[(size_t)(a.sweep - 1) * B + b] = mx;
    if (mx <= 1.0)
      a.status[b] = a.sweep;
    else
      still = true;
  }
  const unsigned long long mk = __ballot(still);
  if (lane == 0 && mk) atomicAdd(a.nactive, __popcll(mk));
}

}  // namespace ocs
