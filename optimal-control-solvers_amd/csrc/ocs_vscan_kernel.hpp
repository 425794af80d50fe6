// ocs_vscan_kernel.hpp -- the discrete-adjoint pass (RK4Integrator.m:59-121) as a scan over time for ANY OCProblem
// (OCProblem/OCProblem.m:8-21: coupled rows, several controls), the state vector of a trajectory in one lane.
//
// The adjoint recursion of the reference (:72-89) is linear in lam whatever the problem: with the stage states of step
// i fixed, lam_i = M_i lam_{i+1} + b_i, M_i an nS x nS matrix and b_i the contribution of the constant cost row
// lam(end) (the last row of dFdx_times_vec is zero, OCProblem.m:14-15).  k_backward_scan (ocs_scan_kernel.hpp) uses
// that for row-separable problems, where M_i is diagonal and a lane owns one row; here a lane owns a whole trajectory
// and the maps are small dense matrices:
//
//   phase 1  (time-parallel)  wave w takes a chunk of L consecutive steps of the workgroup's 64 trajectories,
//            recomputes Y2..Y4 from the checkpoint, obtains the columns of M_i and b_i by running lines :73-88 on the
//            unit vectors (cost row 0) and on the zero vector (cost row lam(end)) -- nS + 1 applications of the
//            plugin's dFdx_times_vec per stage, which is linear in its vector argument by definition -- and composes
//            the chunk map (nS x nS product per step);
//   phase 2  the W chunk maps of a superblock go through LDS; a wave applies the maps of the chunks above its own to
//            the carry (lam at the top of the superblock);
//   phase 3  with the true lam at the top of its chunk a wave runs :73-88 as the reference writes them, stores lam and
//            assembles the dJdu columns (:97-121).
//
// Work per step: (nS + 2) applications of the adjoint stage chain instead of one, spread over W waves per 64
// trajectories: the dependent chain of the pass shrinks by W / (nS + 2), which is what bounds it wherever the serial
// lane kernel leaves most SIMDs of the chip without a wave.  Records, liveness of chunks below step 0 (zero records:
// exact identity maps), buffer addressing and the split at N % L as in ocs_scan_kernel.hpp.
// Functor interface: P::Par / load / Fx / dFdxT / dFduT (ocs_problems.hpp, ocs_user_functor.hpp), NTC = 1.
#pragma once
#include "ocs_scan_kernel.hpp"

namespace ocs {

template <int NS>
struct VScanCfg {
  static constexpr int NM = NS * NS + NS;                       // doubles of a chunk map (M, b)
  // waves per workgroup = chunks per superblock: as many as the two map buffers leave room for in LDS, and at most 8
  // beyond one state (a workgroup of 1024 threads caps a wave at 128 registers, which the dense maps do not fit in)
  static constexpr int W = NS == 1 ? 16 : (NM <= 12 ? 8 : 4);
  static constexpr int L = 4;
};

template <class P, int W, int L, bool OUT_LAM, bool OUT_DJDU, bool LT>
__global__ __launch_bounds__(W * 64) void k_backward_vscan(const BwdArgsScan a) {
  constexpr int NS = P::NS, NC = P::NC, NAUG = P::NAUG, NM = NS * NS + NS;
  static_assert(P::NTC == 1 && W * L + 1 <= kScanPadFront && L + 1 <= 8, "chunk shape");
  __shared__ double sm[2][W][NM][64];   // chunk maps: M row-major, then b
  __shared__ double csm[2][NS][64];     // lam at the bottom of a superblock
  __shared__ __attribute__((aligned(16))) double rcs[2][W][8 * kScanRec];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  const int b0 = blockIdx.x * 64 + lane;
  const bool valid = b0 < a.batch;
  const int b = valid ? b0 : a.batch - 1;
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  const double lamc = LT ? a.lamT[(size_t)NS * B + b] : 1.0;
  const size_t colB = (size_t)NAUG * B;
  double carry[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) carry[k] = LT ? a.lamT[(size_t)k * B + b] : 0.0;   // lam(:, N+1)   :63-66
  if (OUT_LAM && wave == 0 && valid) {
#pragma unroll
    for (int k = 0; k < NS; ++k) a.lam[(size_t)N * colB + (size_t)k * B + b] = carry[k];
    a.lam[(size_t)N * colB + (size_t)NS * B + b] = lamc;
  }
  double pend_top[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) pend_top[c] = (OUT_DJDU && a.pend0) ? a.pend0[(size_t)c * B + b] : 0.0;
  const unsigned col8 = (unsigned)(colB * 8), B8 = (unsigned)(B * 8), ucol8 = (unsigned)((size_t)NC * B * 8);
  const unsigned vb = (unsigned)((size_t)b * 8);
  const unsigned vs = valid ? vb : kOffDrop;

  struct Ld {
    double x[L][NS];
    double u[2 * L + 1][NC];
    double xb[NS], ub0[NC], ub1[NC];   // stage state 4 of the step below the chunk (its B'k4 belongs to column 2 lo)
  };
  auto chunk_lo = [&](int sb) OCS_INLINE { return N - (sb * W + wave + 1) * L; };
  auto load_part = [&](int sb, Ld& d, int slot, int q) OCS_INLINE {
    const int lo = chunk_lo(sb), lc = lo > 0 ? lo : 0;
    const int lr = lo >= 0 ? lo - 1 : -kScanPadFront;
    if (q == 0) dma16_sc(a.RECS + (long long)lr * kScanRec + 2 * lane, &rcs[slot][wave][0]);
    const Buf bx = Buf::make(a.xck + (size_t)lc * colB), bu = Buf::make(a.u + (size_t)(2 * lc) * NC * B);
#pragma unroll
    for (int k = 0; k < NS; ++k) d.x[q][k] = bx.ld(vb, (unsigned)q * col8 + (unsigned)k * B8);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      d.u[2 * q][c] = bu.ld(vb, (unsigned)(2 * q) * ucol8 + (unsigned)c * B8);
      d.u[2 * q + 1][c] = bu.ld(vb, (unsigned)(2 * q + 1) * ucol8 + (unsigned)c * B8);
      if (q == L - 1) d.u[2 * L][c] = bu.ld(vb, (unsigned)(2 * L) * ucol8 + (unsigned)c * B8);
    }
    if (OUT_DJDU && q == 0) {
      const int lb = lo > 0 ? lo - 1 : 0;   // (lo = 0: the values meet a zero record)
      const Buf bxb = Buf::make(a.xck + (size_t)lb * colB), bub = Buf::make(a.u + (size_t)(2 * lb) * NC * B);
#pragma unroll
      for (int k = 0; k < NS; ++k) d.xb[k] = bxb.ld(vb, (unsigned)k * B8);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        d.ub0[c] = bub.ld(vb, (unsigned)c * B8);
        d.ub1[c] = bub.ld(vb, ucol8 + (unsigned)c * B8);
      }
    }
  };
  auto load = [&](int sb, Ld& d, int slot) OCS_INLINE {
#pragma unroll
    for (int q = 0; q < L; ++q) load_part(sb, d, slot, q);
  };
  constexpr int NST = (OUT_LAM ? L * NAUG : 0) + (OUT_DJDU ? (2 * L + 1) * NC : 0);   // stores of a chunk (upper bound)
  struct Rc { double h, hh, h6, h3, tA, tM, tB; };
  auto rec_of = [&](const double* w, int q) OCS_INLINE {
    const double2* pq = reinterpret_cast<const double2*>(w + (q + 1) * kScanRec);
    const double2 a0 = pq[0], a1 = pq[1], a4 = pq[4], a5 = pq[5];
    return Rc{a0.x, a0.y, a1.x, a1.y, a4.x, a4.y, a5.x};
  };
  // stage states of a step from its checkpoint   compute_states :39-46
  auto stages = [&](const Rc& c, const double* xi, const double* uA, const double* uM, double (&Y2)[NS], double (&Y3)[NS],
                    double (&Y4)[NS]) OCS_INLINE {
    double f[NS];
    P::Fx(&c.tA, xi, uA, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y2[k] = __builtin_fma(c.hh, f[k], xi[k]);
    P::Fx(&c.tM, Y2, uM, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y3[k] = __builtin_fma(c.hh, f[k], xi[k]);
    P::Fx(&c.tM, Y3, uM, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y4[k] = __builtin_fma(c.h, f[k], xi[k]);
  };
  // lines :73-88 for one vector: lam (state rows) and lc (cost row) -> lam of the step below; K: optionally k1..k4
  auto adj = [&](const Rc& c, const double* xi, const double (&Y2)[NS], const double (&Y3)[NS], const double (&Y4)[NS],
                 const double* uA, const double* uM, const double* uB, const double (&lm)[NS], double lc, double (&out)[NS],
                 double (*K)[NAUG]) OCS_INLINE {
    double k4[NAUG], k3[NAUG], k2[NAUG], k1[NAUG], g3[NS], g2[NS], g1[NS], g0[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) k4[k] = c.h6 * lm[k];                             // :73
    k4[NS] = c.h6 * lc;
    P::dFdxT(&c.tB, Y4, uB, p, k4, g3);                                            // :74-75
#pragma unroll
    for (int k = 0; k < NS; ++k) k3[k] = __builtin_fma(c.h, g3[k], c.h3 * lm[k]);  // :77
    k3[NS] = c.h3 * lc;
    P::dFdxT(&c.tM, Y3, uM, p, k3, g2);                                            // :78-79
#pragma unroll
    for (int k = 0; k < NS; ++k) k2[k] = __builtin_fma(c.hh, g2[k], c.h3 * lm[k]); // :81
    k2[NS] = c.h3 * lc;
    P::dFdxT(&c.tM, Y2, uM, p, k2, g1);                                            // :82-83
#pragma unroll
    for (int k = 0; k < NS; ++k) k1[k] = __builtin_fma(c.hh, g1[k], c.h6 * lm[k]); // :85
    k1[NS] = c.h6 * lc;
    P::dFdxT(&c.tA, xi, uA, p, k1, g0);                                            // :87-88
#pragma unroll
    for (int k = 0; k < NS; ++k) out[k] = (((lm[k] + g1[k]) + g2[k]) + g3[k]) + g0[k];   // :86-88
    if (K) {
#pragma unroll
      for (int k = 0; k < NAUG; ++k) {
        K[0][k] = k1[k];
        K[1][k] = k2[k];
        K[2][k] = k3[k];
        K[3][k] = k4[k];
      }
    }
  };

  auto process = [&](int sb, const Ld& d, int slot, bool first, Ld& dn) OCS_INLINE {
    const int lo = chunk_lo(sb);
    const bool live = lo >= 0, topc = lo + L == N;
    if (first)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST < 63 ? NST : 63) : "memory");
    const double* rw = &rcs[slot][wave][0];
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
      const Rc c = rec_of(rw, q);
      double Y2[NS], Y3[NS], Y4[NS];
      stages(c, d.x[q], d.u[2 * q], d.u[2 * q + 1], Y2, Y3, Y4);
      double Ms[NS][NS], bs[NS];   // the step's map: column s of Ms from the unit vector e_s, bs from the cost row
#pragma unroll
      for (int s = 0; s <= NS; ++s) {
        double e[NS], o[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) e[k] = k == s ? 1.0 : 0.0;
        adj(c, d.x[q], Y2, Y3, Y4, d.u[2 * q], d.u[2 * q + 1], d.u[2 * q + 2], e, s == NS ? lamc : 0.0, o, nullptr);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          if (s < NS) Ms[k][s] = o[k];
          else bs[k] = o[k];
        }
      }
      // (M, bv) <- (Ms M, Ms bv + bs): the step lies below the steps composed so far
      double Mn[NS][NS], bn[NS];
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
      __builtin_amdgcn_sched_barrier(0);
      load_part(sb + 1, dn, slot ^ 1, L - 1 - q);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
      for (int j = 0; j < NS; ++j) sm[sb & 1][wave][i * NS + j][lane] = M[i][j];
      sm[sb & 1][wave][NS * NS + i][lane] = bv[i];
    }
    lds_barrier_sc();
    // ---------------- phase 2: lam at the top of this chunk ----------------
    double lam[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) lam[k] = (sb == 0) ? carry[k] : csm[(sb & 1) ^ 1][k][lane];
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
    // ---------------- phase 3: the recursion inside the chunk, lam and dJdu stores ----------------
    const int lc = live ? lo : 0, nrec = live ? kNumRec : 0;   // a dead chunk stores nothing
    const Buf bl = Buf::make(a.lam + (size_t)lc * colB, nrec), bd = Buf::make(a.dJdu + (size_t)(2 * lc) * NC * B, nrec);
    double pend[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) pend[c] = topc ? pend_top[c] : 0.0;   // B'k1 of the node above (another chunk's, or none)
#pragma unroll
    for (int q = L - 1; q >= 0; --q) {
      const Rc c = rec_of(rw, q);
      double Y2[NS], Y3[NS], Y4[NS], K[4][NAUG], ln[NS];
      stages(c, d.x[q], d.u[2 * q], d.u[2 * q + 1], Y2, Y3, Y4);
      adj(c, d.x[q], Y2, Y3, Y4, d.u[2 * q], d.u[2 * q + 1], d.u[2 * q + 2], lam, lamc, ln, K);
#pragma unroll
      for (int k = 0; k < NS; ++k) lam[k] = ln[k];
      if (OUT_LAM) {
#pragma unroll
        for (int k = 0; k < NS; ++k) bl.st(lam[k], vs, (unsigned)q * col8 + (unsigned)k * B8);
        bl.st(lamc, vs, (unsigned)q * col8 + (unsigned)NS * B8);
      }
      if (OUT_DJDU) {                                            // compute_dJdu :97-121
        double d4[NC], d3[NC], d2[NC];
        P::dFduT(&c.tB, Y4, d.u[2 * q + 2], p, K[3], d4);
        P::dFduT(&c.tM, Y3, d.u[2 * q + 1], p, K[2], d3);
        P::dFduT(&c.tM, Y2, d.u[2 * q + 1], p, K[1], d2);
        const unsigned so = (unsigned)(2 * q) * ucol8;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
          bd.st(d2[cc] + d3[cc], vs, so + ucol8 + (unsigned)cc * B8);                  // column 2i+1  :105-109
          // column 2i+2 = B'k4 of step i + B'k1 of step i+1: the chunk above stores it as its lowest column, except 2N
          if (q == L - 1)
            bd.st(pend[cc] + d4[cc], topc ? vs : kOffDrop, so + 2 * ucol8 + (unsigned)cc * B8);
          else
            bd.st(pend[cc] + d4[cc], vs, so + 2 * ucol8 + (unsigned)cc * B8);
        }
        P::dFduT(&c.tA, d.x[q], d.u[2 * q], p, K[0], pend);
        if (q == 0) {
          // column 2 lo = B'k1 of step lo + B'k4 of step lo-1 (k4 = h/6 [lam_lo; lam(end)]); column 0: the k1 half :101-102
          const Rc cb = rec_of(rw, -1);
          double Y2b[NS], Y3b[NS], Y4b[NS], k4b[NAUG], d4b[NC];
          stages(cb, d.xb, d.ub0, d.ub1, Y2b, Y3b, Y4b);
#pragma unroll
          for (int k = 0; k < NS; ++k) k4b[k] = cb.h6 * lam[k];
          k4b[NS] = cb.h6 * lamc;
          P::dFduT(&cb.tB, Y4b, d.u[0], p, k4b, d4b);
#pragma unroll
          for (int cc = 0; cc < NC; ++cc) bd.st(pend[cc] + d4b[cc], vs, (unsigned)cc * B8);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wave == W - 1) {
#pragma unroll
      for (int k = 0; k < NS; ++k) csm[sb & 1][k][lane] = lam[k];   // lam at the bottom of the superblock
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) carry[k] = lam[k];
  };

  const int nsb = (N + W * L - 1) / (W * L);
  Ld d0, d1;
  load(0, d0, 0);
  for (int sb = 0; sb < nsb; sb += 2) {
    process(sb, d0, 0, sb == 0, d1);
    process(sb + 1, d1, 1, false, d0);
  }
  if (a.lam0 && wave == W - 1 && valid) {
#pragma unroll
    for (int k = 0; k < NS; ++k) a.lam0[(size_t)k * B + b] = carry[k];
    a.lam0[(size_t)NS * B + b] = lamc;
  }
}

}  // namespace ocs
