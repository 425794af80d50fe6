// ocs_costate_vscan_kernel.hpp -- the costate pass of the forward-backward sweep (functions/compute_x_lam.m:11-14) as a scan
// over time for ANY OCProblem (OCProblem/OCProblem.m:8-21: coupled rows, several controls), the state vector of an instance
// in one lane.
//
// adjointRHS(t, x, lam, u) = -dFdx_times_vec(t, [x; 0], u, [lam; 1])(1:nS) (the A9 adapter) is affine in lam for
// every problem -- dFdx_times_vec is linear in its vector argument by definition -- so the RK4 step from node i+1 down to
// node i (the lines of k_costate, ocs_fbs_device.hpp) is lam_i = M_i lam_{i+1} + b_i with M_i an nS x nS matrix.  The three
// phases of ocs_vscan_kernel.hpp (chunk maps from the unit vectors and the zero vector, maps of the chunks above through
// LDS, the recursion inside the chunk with the true lam), on the inputs of ocs_costate_scan_kernel.hpp: x at the chunk's
// nodes with the pchip slopes and interval midpoints formed here (Fritsch-Carlson, the formulas of k_costate_plx), the
// control samples on the grid, the scan records and the pchip interval records of a chunk by LDS-DMA.
// Work per interval: (nS + 2) RK4 steps of the costate equation instead of one, on W waves per 64 instances.
// Functor interface: P::Par / load / dFdxT (ocs_problems.hpp, ocs_user_functor.hpp), NTC = 1.
#pragma once
#include "ocs_costate_scan_kernel.hpp"

namespace ocs {

template <class P, int W, int L>
__global__ __launch_bounds__(W * 64) void k_costate_vscan(const CostateScanArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NM = NS * NS + NS;
  static_assert(P::NTC == 1 && L == 4 && W * L + 1 <= kScanPadFront && L * NS <= 63, "chunk shape");
  __shared__ double sm[2][W][NM][64];                                  // chunk maps: M row-major, then b
  __shared__ double csm[2][NS][64];                                    // lam at the bottom of a superblock
  __shared__ __attribute__((aligned(16))) double tab[2][W][2][128];    // per wave: scan records | pchip interval records
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  if (a.gate && *a.gate == 0) return;
  const int b0 = blockIdx.x * 64 + lane;
  const bool valid = b0 < a.batch;
  const int b = valid ? b0 : a.batch - 1;
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  const bool fz = a.frozen != nullptr && a.frozen[b] != 0;
  const size_t colB = (size_t)NS * B, xcolB = (size_t)a.ldx * B, ucolB = (size_t)NC * B;
  const unsigned col8 = (unsigned)(colB * 8), B8 = (unsigned)(B * 8);
  const unsigned vst = (valid && !fz) ? (unsigned)((size_t)b * 8) : kOffDrop;
  if (valid && !fz && wave == 0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) a.lam[(size_t)N * colB + (size_t)k * B + b] = 0.0;   // lam(TF) = 0   compute_x_lam.m:4
  }

  struct Ld {
    double w[L + 3][NS];        // x(:, lo-1 .. lo+L+1), clamped to the grid
    double u[2 * L + 1][NC];    // u at grid points 2 lo .. 2 (lo + L)
  };
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
#pragma unroll
      for (int k = 0; k < NS; ++k) d.w[t][k] = a.x[(size_t)i * xcolB + (size_t)k * B + b];
    }
#pragma unroll
    for (int t = 0; t < 2 * L + 1; ++t) {
      int j = 2 * lo + t;
      j = j < 0 ? 0 : (j > 2 * N ? 2 * N : j);
#pragma unroll
      for (int c = 0; c < NC; ++c) d.u[t][c] = a.u[(size_t)j * ucolB + (size_t)c * B + b];
    }
  };

  struct Rc { double h, hh, h6, tA, tM, tB; };
  // adjointRHS with the cost-row entry of the vector given (1 for the true costate, 0 for a unit vector)
  auto rhs = [&](const double* tc, const double* x, const double (&lm)[NS], double cs, const double* u, double (&out)[NS]) OCS_INLINE {
    double v[NS + 1], g[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = lm[k];
    v[NS] = cs;
    P::dFdxT(tc, x, u, p, v, g);
#pragma unroll
    for (int k = 0; k < NS; ++k) out[k] = -g[k];
  };
  // classical RK4 with step -h from node i+1 to node i (k_costate's lines)
  auto step = [&](const Rc& c, const double* xA, const double* xM, const double* xB, const double* uA, const double* uM,
                  const double* uB, const double (&l)[NS], double cs, double (&out)[NS]) OCS_INLINE {
    double k1[NS], k2[NS], k3[NS], k4[NS], Lv[NS];
    rhs(&c.tB, xB, l, cs, uB, k1);
#pragma unroll
    for (int k = 0; k < NS; ++k) Lv[k] = __builtin_fma(-c.hh, k1[k], l[k]);
    rhs(&c.tM, xM, Lv, cs, uM, k2);
#pragma unroll
    for (int k = 0; k < NS; ++k) Lv[k] = __builtin_fma(-c.hh, k2[k], l[k]);
    rhs(&c.tM, xM, Lv, cs, uM, k3);
#pragma unroll
    for (int k = 0; k < NS; ++k) Lv[k] = __builtin_fma(-c.h, k3[k], l[k]);
    rhs(&c.tA, xA, Lv, cs, uA, k4);
#pragma unroll
    for (int k = 0; k < NS; ++k)
      out[k] = __builtin_fma(-c.h6, __builtin_fma(2.0, k3[k], __builtin_fma(2.0, k2[k], k1[k])) + k4[k], l[k]);
  };

  constexpr int NST = L * NS;   // stores of a chunk
  double carry[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) carry[k] = 0.0;

  auto process = [&](int sb, const Ld& d, int slot, Ld& dn) OCS_INLINE {
    const int lo = chunk_lo(sb);
    const bool live = lo >= 0;
    // the loads and the two tables of this superblock have landed once everything but the stores of the superblock
    // before (issued behind them) has
    if (sb == 0)
      __builtin_amdgcn_s_waitcnt(0x0F70);
    else
      __builtin_amdgcn_s_waitcnt(0x0F70 | (NST & 15) | ((NST >> 4) << 14));
    asm volatile("" ::: "memory");
    const double* recs = &tab[slot][wave][0][0];
    const int lp = lo < 0 ? 0 : (lo > N - 8 ? N - 8 : lo);
    const double* prs = &tab[slot][wave][1][0] + (size_t)((lo < 0 ? 0 : lo) - lp) * kPRec;
    // ---------------- pchip midpoints of x (per row, the formulas of k_costate_scan) ----------------
    double xM[L][NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      double sec[L + 2];
#pragma unroll
      for (int t = 0; t < L + 2; ++t) {
        const double ih = t == 0 ? prs[3] : (t == L + 1 ? prs[(L - 1) * kPRec + 5] : prs[(t - 1) * kPRec + 4]);
        sec[t] = (d.w[t + 1][k] - d.w[t][k]) * ih;
      }
      double dsl[L + 1];
#pragma unroll
      for (int c = 0; c < L + 1; ++c)
        dsl[c] = pchip_interior_f(sec[c], sec[c + 1], c < L ? prs[c * kPRec + 6] : prs[(L - 1) * kPRec + 8],
                                  c < L ? prs[c * kPRec + 7] : prs[(L - 1) * kPRec + 9]);
      if (lo == 0) dsl[0] = pchip_end_pl(prs[1], prs[2], sec[1], sec[2]);
      if (lo + L == N) dsl[L] = pchip_end_pl(prs[(L - 1) * kPRec + 1], prs[(L - 1) * kPRec + 0], sec[L], sec[L - 1]);
#pragma unroll
      for (int q = 0; q < L; ++q)
        xM[q][k] = __builtin_fma(prs[q * kPRec + 11], dsl[q] - dsl[q + 1], 0.5 * (d.w[q + 1][k] + d.w[q + 2][k]));
    }
    auto rec_of = [&](int q) OCS_INLINE {
      const double* rc = recs + q * kScanRec;
      return Rc{rc[0], rc[1], rc[2], rc[8], rc[9], rc[10]};
    };
    // ---------------- phase 1: the chunk map ----------------
    double M[NS][NS], bv[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      bv[i] = 0.0;
#pragma unroll
      for (int j = 0; j < NS; ++j) M[i][j] = i == j ? 1.0 : 0.0;
    }
#pragma unroll
    for (int q = L - 1; q >= 0; --q) {
      const Rc c = rec_of(q);
      double Ms[NS][NS], bs[NS];
#pragma unroll
      for (int s = 0; s <= NS; ++s) {
        double e[NS], o[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) e[k] = k == s ? 1.0 : 0.0;
        step(c, d.w[q + 1], xM[q], d.w[q + 2], d.u[2 * q], d.u[2 * q + 1], d.u[2 * q + 2], e, s == NS ? 1.0 : 0.0, o);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          if (s < NS) Ms[k][s] = o[k];
          else bs[k] = o[k];
        }
      }
      double Mn[NS][NS], bn[NS];   // (M, bv) <- (Ms M, Ms bv + bs): the interval lies below the ones composed so far
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        double t = bs[i];
#pragma unroll
        for (int k = 0; k < NS; ++k) t = __builtin_fma(Ms[i][k], bv[k], t);
        bn[i] = t;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          double m = 0.0;
#pragma unroll
          for (int k = 0; k < NS; ++k) m = __builtin_fma(Ms[i][k], M[k][j], m);
          Mn[i][j] = m;
        }
      }
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        bv[i] = bn[i];
#pragma unroll
        for (int j = 0; j < NS; ++j) M[i][j] = Mn[i][j];
      }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
      for (int j = 0; j < NS; ++j) sm[sb & 1][wave][i * NS + j][lane] = M[i][j];
      sm[sb & 1][wave][NS * NS + i][lane] = bv[i];
    }
    load(sb + 1, dn, slot ^ 1);
    lds_barrier_sc();
    // ---------------- phase 2: lam at the top of this chunk ----------------
    double lam[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) lam[k] = (sb == 0) ? 0.0 : csm[(sb & 1) ^ 1][k][lane];
    for (int j = 0; j < wave; ++j) {   // wave-uniform trip count
      double ln[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        double t = sm[sb & 1][j][NS * NS + i][lane];
#pragma unroll
        for (int k = 0; k < NS; ++k) t = __builtin_fma(sm[sb & 1][j][i * NS + k][lane], lam[k], t);
        ln[i] = t;
      }
#pragma unroll
      for (int i = 0; i < NS; ++i) lam[i] = ln[i];
    }
    // ---------------- phase 3: the recursion inside the chunk and the stores ----------------
    const int lc = live ? lo : 0;
    const Buf bl = Buf::make(a.lam + (size_t)lc * colB, live ? kNumRec : 0);   // a dead chunk stores nothing
#pragma unroll
    for (int q = L - 1; q >= 0; --q) {
      const Rc c = rec_of(q);
      double ln[NS];
      step(c, d.w[q + 1], xM[q], d.w[q + 2], d.u[2 * q], d.u[2 * q + 1], d.u[2 * q + 2], lam, 1.0, ln);
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        lam[k] = ln[k];
        bl.st(lam[k], vst, (unsigned)q * col8 + (unsigned)k * B8);
      }
    }
    if (wave == W - 1) {
#pragma unroll
      for (int k = 0; k < NS; ++k) csm[sb & 1][k][lane] = lam[k];
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) carry[k] = lam[k];
  };

  const int nsb = (N + W * L - 1) / (W * L);
  Ld d0, d1;
  load(0, d0, 0);
  for (int sb = 0; sb < nsb; sb += 2) {   // (a superblock past the horizon: dead chunks, identity maps, no stores)
    process(sb, d0, 0, d1);
    process(sb + 1, d1, 1, d0);
  }
  (void)carry;
}

}  // namespace ocs
