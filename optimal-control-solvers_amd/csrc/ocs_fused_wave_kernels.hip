// ocs_fused_wave_kernels.hip -- the shooting objective and its gradient (functions/single_shooting.m:137-150) for a
// dense control basis with few functions (Control/ChebyshevControl.m:21-43), on the wave-specialised state pass and
// the time-parallel adjoint scan, with both products with the basis matrix on the matrix cores:
//
//   u = reshape(v,nC,[]) * B          (ChebyshevControl.m:35-38)   [B x nBasis] . [nBasis x (2N+1)]
//   dJdv = dJdu * B'                  (ChebyshevControl.m:41-43)   [B x (2N+1)] . [(2N+1) x nBasis]
//
// are genuine matrix products over the batch: as tiles of v_mfma_f64_16x16x4_f64 they take one issue slot per 1024
// multiply-adds and keep their operands in registers, instead of 32 + 32 + 32 of the ~170 fp64 vector instructions per
// (trajectory, step).  (They do not take less PIPE time: on gfx950 the fp64 matrix instruction occupies the SIMD's fp64
// datapath for its 64 cycles -- scripts/probe/mfma_valu_overlap.hip -- which is why this path is the choice of the
// latency-bound small batches and the lane kernels that of the full chip.)  Neither u nor dJdu exists in memory: HBM
// traffic per (trajectory, step) is the checkpoint write + read of the state rows, 16 nS bytes.
//
//   state pass    k_forward_p2<..., NKS> (ocs_pipeline2_kernel.hpp): expansion waves U produce the block of samples the
//                 recursion wave reads two intervals later.
//   adjoint pass  k_backward_fcs below: k_backward_scan's three phases (ocs_scan_kernel.hpp) on another lane mapping.
//                 A wave owns 16 trajectories and FOUR consecutive chunks of L steps: lane (g, n) = (lane >> 4,
//                 lane & 15) runs chunk g of trajectory n.  That is the operand layout of the matrix instruction:
//                   expansion    D[i][n] = sum_k A[i][k] B[k][n]: row i = 4 m + g' of register m of lane (g', n) is
//                                made sample 4 t + m of chunk g' of tile set t, so a lane receives exactly the samples
//                                of its own chunk (two sets: samples 0..7; a third for sample 8, the node it shares
//                                with the chunk above);
//                   contraction  D[i][n] += sum_k A[i][k] B[k][n] with k-slot g fed by lane (g, n)'s own column m of
//                                dJdu and A[i][g] = B(i, 2 lo_g + m): one instruction folds one column of each of
//                                the wave's four chunks into dJdv (16 functions x 16 trajectories, in registers for
//                                the whole pass).
//                 No transposition through LDS on either side.  The scan over the chunk maps is two-level: across the
//                 four chunks of a wave by row-swapping permlane instructions, across the waves through LDS.
#include "ocs_internal.hpp"
#include "ocs_pipeline2_kernel.hpp"
#include "ocs_problems.hpp"
#include "ocs_scan_kernel.hpp"
#include <cstdio>
#include <cstdlib>

namespace ocs {

static inline int hip_rc_fw(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

typedef double d4_fw __attribute__((ext_vector_type(4)));
__device__ static inline d4_fw mma_fw(double a, double b, d4_fw c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// rows (16 lanes) of a: [a0 a1 a2 a3] -> e = [a0 a0 a2 a2], o = [a1 a1 a3 a3]
__device__ static inline void swap16_fw(double a, double& e, double& o) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(a), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(a), false, false);
  e = __hiloint2double((int)hi[0], (int)lo[0]);
  o = __hiloint2double((int)hi[1], (int)lo[1]);
}
// -> lower = [a0 a1 a0 a1], upper = [a2 a3 a2 a3]
__device__ static inline void swap32_fw(double a, double& lower, double& upper) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(a), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(a), false, false);
  lower = __hiloint2double((int)hi[0], (int)lo[0]);
  upper = __hiloint2double((int)hi[1], (int)lo[1]);
}

// 16-byte-per-lane LDS-DMA as dma16_sc, issued from inline assembly.  The compiler orders every later LDS read behind the
// LDS-DMA instructions it knows about once there are more than a few of them in flight (a vmcnt(0) right behind their
// issue: the whole memory latency, every superblock); the kernel below waits for its DMAs itself, one superblock later.
__device__ static inline void dma16_fw(const double* src, const double* lds_dst) {
  const unsigned m0v = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)lds_dst;
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
}

#ifdef OCS_FCS_STAMPS
__device__ static long long g_fcs_stamp[8 * 8];   // workgroup 0: per wave, cycles per segment of a superblock, summed
#define FCS_T(k) do { const long long t_ = __builtin_amdgcn_s_memtime(); seg_[k] += t_ - tl_; tl_ = t_; } while (0)
#else
#define FCS_T(k)
#endif

struct BwdArgsFcs {
  int N, batch, nBasis, ldbt;   // N: a multiple of L
  const double* RECS;           // scan records (ocs_scan_kernel.hpp), record of step 0
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* xck;            // [N+1][nAug][B] checkpoints (state rows)
  const double* BT;             // [2N+1][ldbt] transposed basis, zero-padded
  const double* v;              // [nBasis][B]
  double* dJdv;                 // [nBasis][B]
  double* lam0;                 // optional [nAug][B]: lam(:,1)  (single_shooting.m:149)
};

constexpr int kFcsCH = 4;   // chunks per wave (the four 16-lane rows)

// W waves per workgroup of 16 trajectories, L steps per chunk: a superblock is 4 W L steps.
// NKS: k-steps of the expansion (4 basis functions each); the gradient has NRT = ceil(NKS / 4) tiles of 16 functions.
// ABL (diagnostic builds, -DOCS_FCS_ABL): 1 no expansion products, 2 no contraction products, 3 neither, 4 no barrier,
// 5 no phase 3, 6 no phase 1
template <class P, int W, int L, int NKS, int ABL = 0>
__global__ __launch_bounds__(W * 64, (NKS > 4 ? 1024 : 2048) / (W * 64)) void k_backward_fcs(const BwdArgsFcs a) {
  constexpr int NAUG = P::NAUG, CH = kFcsCH, SB = W * CH * L, NRT = (NKS + 3) / 4, NSET = (2 * L + 1 + 3) / 4;
  constexpr int NRD = (CH * L + 1 + 7) / 8;   // record DMAs per wave and superblock (8 records each)
  static_assert(P::NS == 1 && P::NC == 1 && P::NTC == 1 && P::ROW_SEPARABLE && !P::DFDU_READS_Y, "one state row per trajectory");
  static_assert(SB + 1 <= kScanPadFront && 8 * NRD - (CH * L + 1) < kScanPadBack && L % 2 == 0, "superblock shape");
  typedef typename P::Stage Stage;
  __shared__ __attribute__((aligned(16))) double2 sm[2][W][16];          // wave maps of a superblock
  __shared__ double csm[2][16];                                          // lam at the bottom of a superblock
  __shared__ __attribute__((aligned(16))) double rcs[2][W][NRD * 128];   // records lo_low-1 .. of a wave
  constexpr int LDW = NKS > 4 ? 32 : 16;   // doubles per row of the basis table (= a.ldbt)
  __shared__ double red[W][NRT * 4][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, n = lane & 15;
  const size_t B = (size_t)a.batch;
  const int N = a.N, nT = 2 * N + 1;
  const int b0 = blockIdx.x * 16 + n;
  const bool valid = b0 < a.batch;
  const int b = valid ? b0 : a.batch - 1;
  const typename P::RowPar rp = P::load_row(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b}, 0);
  const double lamc = 1.0;   // lam(:, end) = e_last   RK4Integrator.m:63-69
  const size_t colB = (size_t)NAUG * B;
  const unsigned col8 = (unsigned)(colB * 8), b8 = (unsigned)((size_t)b * 8);

  // B operand of the expansion: k-slot g of trajectory n
  double vB[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int kk = 4 * ks + g;
    const double val = a.v[(size_t)(kk < a.nBasis ? kk : 0) * B + b];
    vB[ks] = kk < a.nBasis ? val : 0.0;
  }
  d4_fw acc[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) acc[rt] = d4_fw{0.0, 0.0, 0.0, 0.0};

  // lowest step of the wave's four chunks in superblock sb; chunk g covers lo_low + (3 - g) L .. + L - 1
  auto wave_lo = [&](int sb) OCS_INLINE { return N - (sb * W * CH + wave * CH + CH) * L; };
  auto clamp_smp = [&](int s) OCS_INLINE { return s < 0 ? 0 : (s >= nT ? nT - 1 : s); };

  struct Ld { double x[L]; };
  auto load_part = [&](int sb, Ld& d, int slot, int q) OCS_INLINE {
    const int lo_low = wave_lo(sb);
    if (q == 0) {
      const int lr = lo_low - 1 >= -kScanPadFront ? lo_low - 1 : -kScanPadFront;   // (below that: zero records anyway)
#pragma unroll
      for (int k = 0; k < NRD; ++k)
        dma16_fw(a.RECS + (long long)lr * kScanRec + k * 128 + 2 * lane, &rcs[slot][wave][k * 128]);
    }
    const int base = lo_low > 0 ? lo_low : 0;
    const int lo_g = lo_low + (CH - 1 - g) * L;
    const unsigned voff = lo_g >= 0 ? (unsigned)(lo_g - base) * col8 + b8 : kOffDrop;   // a dead chunk reads zeros
    const Buf bx = Buf::make(a.xck + (size_t)base * colB);
    d.x[q] = bx.ld(voff, (unsigned)q * col8);
  };
  struct Rc { double h, hh, h6, h3, s4, s3, s1, tA, tM, tB; };
  auto rec_of = [&](const double* w, int q) OCS_INLINE {   // record of step lo_g + q (q = -1: the step below the chunk)
    const double2* p = reinterpret_cast<const double2*>(w + ((CH - 1 - g) * L + q + 1) * kScanRec);
    const double2 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5];
    return Rc{a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a4.x, a4.y, a5.x};
  };

  double carry = 0.0;
#ifdef OCS_FCS_STAMPS
  long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
#endif
  auto process = [&](int sb, const Ld& d, int slot, Ld& dn) OCS_INLINE {
    const int lo_low = wave_lo(sb);
    FCS_T(7);
    // everything in flight belongs to this superblock (x, expansion operands, records); the records go to LDS, which
    // the compiler's counters do not see
    // (the builtin, not inline assembly: the compiler must know that its own loads have landed too, or its counted
    //  waits for them -- which do not count the DMAs of dma16_fw -- would wait for the DMAs issued in between)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    asm volatile("" ::: "memory");
    FCS_T(0);   // wait for the loads
    const double* rw = &rcs[slot][wave][0];
    const Buf bbt = Buf::make(a.BT);
    // ---------------- the control samples of the four chunks ----------------
    double uu[2 * L + 1];
    {
      // row i = lane & 15 = 4 m + g' is sample 4 t + m of chunk g'; this lane holds k-slot lane >> 4
      // (read when they are needed: with four waves on a SIMD the others cover the latency, and no register waits a
      //  superblock for its turn)
      double aE[NSET][NKS];
      const int s0 = 2 * (lo_low + (CH - 1 - (n & 3)) * L) + (n >> 2);
#pragma unroll
      for (int t = 0; t < NSET; ++t) {
        const unsigned voff = (unsigned)clamp_smp(s0 + 4 * t) * (unsigned)(LDW * 8) + (unsigned)g * 8u;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) aE[t][ks] = bbt.ld(voff, (unsigned)(4 * ks) * 8u);
      }
      d4_fw e[NSET];
#pragma unroll
      for (int t = 0; t < NSET; ++t) e[t] = d4_fw{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int t = 0; t < NSET; ++t) {
          if (ABL == 1 || ABL == 3)
            e[t] += d4_fw{aE[t][ks], vB[ks], aE[t][ks], vB[ks]};
          else
            e[t] = mma_fw(aE[t][ks], vB[ks], e[t]);
        }
#pragma unroll
      for (int t = 0; t < NSET; ++t) {
        if (4 * t + 0 <= 2 * L) uu[4 * t + 0] = e[t].x;
        if (4 * t + 1 <= 2 * L) uu[4 * t + 1] = e[t].y;
        if (4 * t + 2 <= 2 * L) uu[4 * t + 2] = e[t].z;
        if (4 * t + 3 <= 2 * L) uu[4 * t + 3] = e[t].w;
      }
    }
    FCS_T(1);   // expansion
    // ---------------- phase 1: stage states and the step maps ----------------
    // lam_i = alpha_i lam_{i+1} + beta_i per step (ocs_scan_kernel.hpp); the L steps are independent of each other, the
    // stage states and the step maps stay in registers for phase 3
#pragma unroll
    for (int q = 0; q < L; ++q) load_part(sb + 1, dn, slot ^ 1, q);
    // (the stage states are formed again in phase 3: nine operations per step against six registers per step, which
    //  decide between three and four waves per SIMD)
    double al[L], be[L];
#pragma unroll
    for (int q = 0; q < L; ++q) {
      if (ABL == 6) {
        al[q] = uu[2 * q + 2] + d.x[q]; be[q] = 1.0;
        continue;
      }
      const Rc c = rec_of(rw, q);
      const double xi = d.x[q], uA = uu[2 * q], uM = uu[2 * q + 1], uB = uu[2 * q + 2];
      double f = P::g_row_f(xi, uA, c.tA, rp);                 // compute_states :39-46
      const double Y2q = __builtin_fma(c.hh, f, xi);
      f = P::g_row_f(Y2q, uM, c.tM, rp);
      const double Y3q = __builtin_fma(c.hh, f, xi);
      f = P::g_row_f(Y3q, uM, c.tM, rp);
      const double Y4q = __builtin_fma(c.h, f, xi);
      const Stage s4 = P::template stage<false>(c.s4, c.h6, c.tB, lamc), s3 = P::template stage<false>(c.s3, c.h3, c.tM, lamc),
                  s1 = P::template stage<false>(c.s1, c.h6, c.tA, lamc);
      double a4, b4, a3, b3, a2, b2, a1, b1;                    // (p, q): the quantity is p lam_{i+1} + q   :73-88
      P::g_row_dfdx_pre(Y4q, uB, s4, rp, a4, b4);
      const double g3p = a4 * c.h6, g3q = b4;
      const double k3p = __builtin_fma(c.h, g3p, c.h3), k3q = c.h * g3q;
      P::g_row_dfdx_pre(Y3q, uM, s3, rp, a3, b3);
      const double g2p = a3 * k3p, g2q = __builtin_fma(a3, k3q, b3);
      const double k2p = __builtin_fma(c.hh, g2p, c.h3), k2q = c.hh * g2q;
      P::g_row_dfdx_pre(Y2q, uM, s3, rp, a2, b2);
      const double g1p = a2 * k2p, g1q = __builtin_fma(a2, k2q, b2);
      const double k1p = __builtin_fma(c.hh, g1p, c.h6), k1q = c.hh * g1q;
      P::g_row_dfdx_pre(xi, uA, s1, rp, a1, b1);
      const double g0p = a1 * k1p, g0q = __builtin_fma(a1, k1q, b1);
      al[q] = (((1.0 + g1p) + g2p) + g3p) + g0p;
      be[q] = ((g1q + g2q) + g3q) + g0q;
    }
    double A = al[L - 1], Bq = be[L - 1];   // the chunk: lam at its bottom = A lam at its top + Bq
#pragma unroll
    for (int q = L - 2; q >= 0; --q) {
      Bq = __builtin_fma(al[q], Bq, be[q]);
      A = al[q] * A;
    }
    __builtin_amdgcn_sched_barrier(0);
    FCS_T(2);   // phase 1
    // ---------------- the maps of the chunks above, inside the wave ----------------
    // chunk g maps lam at its top to lam at its bottom: (A, Bq).  With e / o the maps of the even / odd chunk of the own
    // pair and P01 = chunk 1 after chunk 0, P23 = chunk 3 after chunk 2:
    //   above chunk 0: identity, chunk 1: e, chunk 2: P01, chunk 3: e after P01; the wave's map: P23 after P01.
    double eA, oA, eB, oB;
    swap16_fw(A, eA, oA);
    swap16_fw(Bq, eB, oB);
    const double pA = oA * eA, pB = __builtin_fma(oA, eB, oB);          // the own pair: odd chunk after even chunk
    double p01A, p23A, p01B, p23B;
    swap32_fw(pA, p01A, p23A);
    swap32_fw(pB, p01B, p23B);
    const double TA = p23A * p01A, TB = __builtin_fma(p23A, p01B, p23B);
    const double e3A = eA * p01A, e3B = __builtin_fma(eA, p01B, eB);
    const double EA = g == 0 ? 1.0 : g == 1 ? eA : g == 2 ? p01A : e3A;
    const double EB = g == 0 ? 0.0 : g == 1 ? eB : g == 2 ? p01B : e3B;
    if (g == 0) sm[sb & 1][wave][n] = double2{TA, TB};
    FCS_T(3);   // maps inside the wave
    if (ABL != 4) lds_barrier_sc();
    FCS_T(4);   // barrier
    // ---------------- phase 2: lam at the top of this chunk ----------------
    double lam = (sb == 0) ? 0.0 : csm[(sb & 1) ^ 1][n];
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if (j < wave) {   // wave-uniform
        const double2 ab = sm[sb & 1][j][n];
        lam = __builtin_fma(ab.x, lam, ab.y);
        asm volatile("" : "+v"(lam));
      }
    }
    lam = __builtin_fma(EA, lam, EB);
    // ---------------- phase 3: lam above every step, then the columns of dJdu (the steps independent again) -------------
    double lt[L];   // lam(i+1) for step i = lo + q
    lt[L - 1] = lam;
#pragma unroll
    for (int q = L - 1; q >= 1; --q) lt[q - 1] = __builtin_fma(al[q], lt[q], be[q]);
    lam = __builtin_fma(al[0], lt[0], be[0]);   // lam at the bottom of the chunk
    double col[2 * L], ctop = 0.0, pk1[L], p4s[L];
#pragma unroll
    for (int q = 0; q < L; ++q) {
      if (ABL == 5) {
        p4s[q] = lt[q]; col[2 * q + 1] = al[q]; pk1[q] = be[q];
        continue;
      }
      const Rc c = rec_of(rw, q);
      const double xi = d.x[q], uA = uu[2 * q], uM = uu[2 * q + 1], uB = uu[2 * q + 2];
      double f = P::g_row_f(xi, uA, c.tA, rp);
      const double Y2q = __builtin_fma(c.hh, f, xi);
      f = P::g_row_f(Y2q, uM, c.tM, rp);
      const double Y3q = __builtin_fma(c.hh, f, xi);
      f = P::g_row_f(Y3q, uM, c.tM, rp);
      const double Y4q = __builtin_fma(c.h, f, xi);
      const Stage s4 = P::template stage<false>(c.s4, c.h6, c.tB, lamc), s3 = P::template stage<false>(c.s3, c.h3, c.tM, lamc),
                  s1 = P::template stage<false>(c.s1, c.h6, c.tA, lamc);
      const double h6l = c.h6 * lt[q], h3l = c.h3 * lt[q];
      const double k4 = h6l;                                        // :73
      const double g3 = P::g_row_dfdx(Y4q, uB, k4, s4, rp);       // :74-75
      const double k3 = __builtin_fma(c.h, g3, h3l);                // :77
      const double g2 = P::g_row_dfdx(Y3q, uM, k3, s3, rp);       // :78-79
      const double k2 = __builtin_fma(c.hh, g2, h3l);               // :81
      const double g1 = P::g_row_dfdx(Y2q, uM, k2, s3, rp);       // :82-83
      const double k1 = __builtin_fma(c.hh, g1, h6l);               // :85
      // compute_dJdu :97-121: column 2i+1 = B'k2 + B'k3; column 2i+2 = B'k4 of step i + B'k1 of step i+1 -- the latter
      // belongs to the chunk of step i+1 (its lowest column) except column 2N
      p4s[q] = P::g_row_dfdu(Y4q, uB, k4, s4, rp);
      col[2 * q + 1] = P::g_row_dfdu(Y3q, uM, k3, s3, rp) + P::g_row_dfdu(Y2q, uM, k2, s3, rp);
      pk1[q] = P::g_row_dfdu(xi, uA, k1, s1, rp);
    }
#pragma unroll
    for (int q = 0; q + 1 < L; ++q) col[2 * q + 2] = pk1[q + 1] + p4s[q];
    ctop = p4s[L - 1];          // (the k1 half of column 2N is zero: RK4Integrator.m:119-120)
    {
      // column 2 lo = B'k1 of step lo + B'k4 of step lo-1 (k4 = h/6 lam_lo); column 0 has the k1 half only :101-102
      const Rc cb = rec_of(rw, -1);
      const Stage sb4 = P::template stage<false>(cb.s4, cb.h6, cb.tB, lamc);
      col[0] = pk1[0] + P::g_row_dfdu(0.0, uu[0], cb.h6 * lam, sb4, rp);
    }
    __builtin_amdgcn_sched_barrier(0);
    FCS_T(5);   // phases 2 and 3
    // ---------------- dJdv += dJdu(:, j) B(:, j)' ----------------
    // A operands: k-slot lane >> 4 takes column m of chunk lane >> 4; row i = lane & 15
    double aC[2 * L][NRT];
#pragma unroll
    for (int m = 0; m < 2 * L; ++m) {
      const unsigned voff = (unsigned)clamp_smp(2 * (lo_low + (CH - 1 - g) * L) + m) * (unsigned)(LDW * 8) + (unsigned)n * 8u;
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) aC[m][rt] = bbt.ld(voff, (unsigned)(16 * rt) * 8u);
    }
#pragma unroll
    for (int m = 0; m < 2 * L; ++m)
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
        if (ABL == 2 || ABL == 3)
          acc[rt] += d4_fw{aC[m][rt], col[m], aC[m][rt], col[m]};
        else
          acc[rt] = mma_fw(aC[m][rt], col[m], acc[rt]);
      }
    if (sb == 0 && wave == 0) {   // column 2N, from the topmost chunk
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
        acc[rt] = mma_fw(a.BT[(size_t)(nT - 1) * LDW + 16 * rt + n], g == 0 ? ctop : 0.0, acc[rt]);
    }
    if (wave == W - 1 && g == CH - 1) csm[sb & 1][n] = lam;   // lam at the bottom of the superblock
    carry = lam;
    FCS_T(6);   // contraction
  };

  const int nsb = (N + SB - 1) / SB;
  Ld d0, d1;
#pragma unroll
  for (int q = 0; q < L; ++q) load_part(0, d0, 0, q);
  for (int sb = 0; sb < nsb; sb += 2) {
    process(sb, d0, 0, d1);
    process(sb + 1, d1, 1, d0);   // (past the horizon: dead chunks, identity maps, zero columns)
  }
  if (a.lam0 && wave == W - 1 && g == CH - 1 && valid) {
    a.lam0[b] = carry;
    a.lam0[B + b] = lamc;
  }
#ifdef OCS_FCS_STAMPS
  if (blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 8; ++k) g_fcs_stamp[wave * 8 + k] = seg_[k];
#endif
  // the W partial gradients of the workgroup's 16 trajectories
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    red[wave][4 * rt + 0][lane] = acc[rt].x;
    red[wave][4 * rt + 1][lane] = acc[rt].y;
    red[wave][4 * rt + 2][lane] = acc[rt].z;
    red[wave][4 * rt + 3][lane] = acc[rt].w;
  }
  __syncthreads();
  if (wave == 0 && valid) {
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        double s = red[0][4 * rt + m][lane];
#pragma unroll
        for (int w = 1; w < W; ++w) s += red[w][4 * rt + m][lane];
        const int kk = 16 * rt + 4 * m + g;   // register m of lane (g, n): row 4 m + g
        if (kk < a.nBasis) a.dJdv[(size_t)kk * B + b] = s;
      }
  }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
constexpr int kFcsW = 8, kFcsL = 4;

bool fused_wave_supported(Functor f, int nS, int nC, int nBasis, int N, int batch) {
  // state pass: whole blocks of 8 steps, whole tiles of 64 / nS trajectories; adjoint pass: one state row
  return f == Functor::Logistic && nS == 1 && nC == 1 && nBasis >= 1 && nBasis <= 32 && N >= 8 && N % 8 == 0 &&
         batch % 64 == 0 && (size_t)batch * 16u * 16u < 0x7FFFFFF0u;
}

template <class P, int NKS, int TPW>
static void run_forward_fcw_t(const FwdArgsP2& a, bool uniform, hipStream_t s) {
  using C_ = P2Cfg<P::NS, NKS, TPW>;
  const dim3 grid(a.batch / C_::TPW), block(C_::NWAVE * 64);
  if (uniform)
    k_forward_p2<P, true, false, true, NKS, TPW><<<grid, block, 0, s>>>(a);
  else
    k_forward_p2<P, true, false, false, NKS, TPW><<<grid, block, 0, s>>>(a);
}
// Half tiles (32 trajectories per workgroup) while that fills the chip at most twice: the waves beside the recursion wave
// (objective, expansion: three quarters of the fp64 work of a block) then have half the work per CU, and the pass runs
// at the pace of the recursion (measured at batch 8192, N = 1000: 97 us with 64-trajectory tiles).
template <class P, int NKS>
static void run_forward_fcw(const FwdArgsP2& a, bool uniform, hipStream_t s) {
  if (a.batch / 32 <= 512)
    run_forward_fcw_t<P, NKS, 32>(a, uniform, s);
  else
    run_forward_fcw_t<P, NKS, 64>(a, uniform, s);
}
template <class P, int NKS>
static void run_backward_fcs(const BwdArgsFcs& a, hipStream_t s) {
  const dim3 grid((a.batch + 15) / 16), block(kFcsW * 64);
#ifdef OCS_FCS_ABL
  static const int abl = getenv("OCS_FCS_ABL") ? atoi(getenv("OCS_FCS_ABL")) : 0;
  if (NKS == 4) {
#define OCS_ABL_CASE(K) if (abl == K) return (void)(k_backward_fcs<P, kFcsW, kFcsL, 4, K><<<grid, block, 0, s>>>(a));
    OCS_ABL_CASE(1) OCS_ABL_CASE(2) OCS_ABL_CASE(3) OCS_ABL_CASE(4) OCS_ABL_CASE(5) OCS_ABL_CASE(6)
#undef OCS_ABL_CASE
  }
#endif
  k_backward_fcs<P, kFcsW, kFcsL, NKS><<<grid, block, 0, s>>>(a);
}

// BT: [2N+1][ldbt] (ldbt = 16 or 32: the fused-control layout of ocs_control.cpp)
int launch_forward_fcw(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, int ldbt, const double* BT,
                       const double* v, const double* x0, double* ck, double* J, hipStream_t s) {
  if (!fused_wave_supported(p.functor, p.nS, p.nC, nBasis, g.N, batch) || ldbt < 4 * ((nBasis + 3) / 4)) return -1;
  FwdArgsP2 a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, x0, nullptr, ck, J, nullptr, 0, 1, nullptr};
  a.BT = BT; a.v = v; a.nBasis = nBasis; a.ldbt = ldbt;
  switch ((nBasis + 3) / 4) {
    case 1: run_forward_fcw<LogisticK<1>, 1>(a, g.uniform, s); break;
    case 2: run_forward_fcw<LogisticK<1>, 2>(a, g.uniform, s); break;
    case 3: run_forward_fcw<LogisticK<1>, 3>(a, g.uniform, s); break;
    case 4: run_forward_fcw<LogisticK<1>, 4>(a, g.uniform, s); break;
    default: run_forward_fcw<LogisticK<1>, 8>(a, g.uniform, s); break;
  }
#ifdef OCS_P2_STAMPS
  {   // diagnostic build: cycles of every role of workgroup 0 (barrier wait / total), roles as in P2Cfg::role
    static int calls = 0;
    if (++calls % 8 == 0) {
      (void)hipStreamSynchronize(s);
      long long h[64];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_p2_stamp), sizeof(h));
      fprintf(stderr, "[p2 fcw batch %d] role: barrier wait / total cycles:", batch);
      const char* names[10] = {"M", "S", "C0", "C1", "C2", "C3", "J", "P", "U0", "U1"};
      for (int w = 0; w < 10; ++w) fprintf(stderr, " %s %lld/%lld", names[w], h[4 * w], h[4 * w + 1]);
      fprintf(stderr, "\n");
    }
  }
#endif
  return hip_rc_fw(hipGetLastError());
}
int launch_backward_fcs(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, int ldbt, const double* BT,
                        const double* v, const double* ck, double* dJdv, double* lam0, hipStream_t s) {
  if (!fused_wave_supported(p.functor, p.nS, p.nC, nBasis, g.N, batch) || !g.RECS || g.N % kFcsL != 0 ||
      ldbt < 16 * ((nBasis + 15) / 16))
    return -1;
  const BwdArgsFcs a{g.N, batch, nBasis, ldbt, g.RECS, p.ps, p.pb, p.pmask, ck, BT, v, dJdv, lam0};
  switch ((nBasis + 3) / 4) {
    case 1: run_backward_fcs<LogisticK<1>, 1>(a, s); break;
    case 2: run_backward_fcs<LogisticK<1>, 2>(a, s); break;
    case 3: run_backward_fcs<LogisticK<1>, 3>(a, s); break;
    case 4: run_backward_fcs<LogisticK<1>, 4>(a, s); break;
    default: run_backward_fcs<LogisticK<1>, 8>(a, s); break;
  }
#ifdef OCS_FCS_STAMPS
  {
    static int calls = 0;
    if (++calls % 8 == 0) {
      (void)hipStreamSynchronize(s);
      long long h[64];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fcs_stamp), sizeof(h));
      for (int w = 0; w < kFcsW; ++w)
        fprintf(stderr, "[fcs batch %d wave %d] wait %lld expansion %lld phase1 %lld maps %lld barrier %lld phase2+3 %lld contraction %lld other %lld\n",
                batch, w, h[8 * w], h[8 * w + 1], h[8 * w + 2], h[8 * w + 3], h[8 * w + 4], h[8 * w + 5], h[8 * w + 6], h[8 * w + 7]);
    }
  }
#endif
  return hip_rc_fw(hipGetLastError());
}

}  // namespace ocs
