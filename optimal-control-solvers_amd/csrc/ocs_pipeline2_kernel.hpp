// ocs_pipeline2_kernel.hpp -- state pass (RK4Integrator.m:28-56) for row-separable problems, wave-specialised
// with a MINIMAL recursion wave.
//
// The pass has one inherently serial part, the state recursion y_{i+1} = Phi_i(y_i): eight dependent fp64
// operations per step and row.  Everything else -- the objective integrand at the four stage states, its
// quadrature, the stores of the trajectory -- only needs y_i and is independent from step to step.  A lone wave
// issues one instruction per ~6 cycles whatever it is, so the pass runs at (instructions of the recursion wave
// per step) x 6 cycles x N.  Here the recursion wave S executes per step: one 16-byte LDS read (the two new
// prepared control terms), the eleven arithmetic instructions of the step, one 8-byte LDS write (y_i).  All other
// work is time-parallel and spread over the other SIMDs of the CU:
//
//   wave M/P  streams control samples and step records HBM -> LDS (LDS-DMA, 1 KiB per instruction, Q blocks
//             ahead, the only wave that waits for memory) and prepares the control terms of the NEXT block for S
//             (P::row_vertex: m_r^2/4 - u, so that a stage evaluation of S is one fused multiply-add);
//   wave S    the recursion, on z = y - m_r/2 (P::row_f_shifted);
//   waves C   (2 or 4) take the steps of the block S finished in the previous interval: recompute the stage
//             states from y_i (three more F evaluations per step: cheaper than a second LDS write on S), form the
//             objective increment d_i, store x(t_i).  A lane takes all rows of one (trajectory, step), so row
//             sums need no cross-lane traffic; lanes are trajectory-fastest, so every quad stores into one line;
//   wave J    prefix-sums the objective increments of the block before that, stores the cost row and J.
//
// One LDS-only barrier per block of D = 8 steps.  Interval k (between barriers k and k+1):
//   M: issues block k+1+Q, waits for block k+2    P: prepares block k+1    S: block k    C: block k-1    J: block k-2
// Results: the arithmetic of a step is that of k_forward_pl (same formulas, same association) -- the recursion on
// z and the objective summed over rows before the quadrature weights (Numerics) -- to round-off the
// lane kernels' and the oracle's.
// This header holds the kernel template only (also compiled by hipRTC for user problems given as row functions); the
// launchers are in ocs_pipeline2_kernels.hip.  Registry problems use the shifted form of their rows (HAS_SHIFT), user
// problems the generic g_row_f / g_row_q.
#pragma once
#include "ocs_device_common.hpp"

namespace ocs {

#ifdef OCS_P2_STAMPS
__device__ static long long g_p2_stamp[16 * 4];   // per wave role: {barrier wait, total, -, -} of workgroup 0
__device__ static long long g_p2_wg[1024 * 4];    // per workgroup: S wave {start, end, barrier wait, xcc}
#define P2_BARRIER() do { const long long t0_ = __builtin_amdgcn_s_memtime(); lds_barrier_p2_(); tbar_ += __builtin_amdgcn_s_memtime() - t0_; } while (0)
#define P2_BEGIN() long long tbar_ = 0; const long long tstart_ = __builtin_amdgcn_s_memtime(); const long long rstart_ = __builtin_amdgcn_s_memrealtime()
#define P2_END(w) do { if (blockIdx.x == 0 && lane == 0) { g_p2_stamp[(w) * 4] = tbar_; g_p2_stamp[(w) * 4 + 1] = __builtin_amdgcn_s_memtime() - tstart_; } if ((w) == 1 && lane == 0 && blockIdx.x < 1024) { g_p2_wg[blockIdx.x * 4] = tstart_; g_p2_wg[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); g_p2_wg[blockIdx.x * 4 + 2] = rstart_; g_p2_wg[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime() - rstart_; } } while (0)
#else
#define P2_BARRIER() lds_barrier_p2_()
#define P2_BEGIN()
#define P2_END(w)
#endif
__device__ static inline void lds_barrier_p2_() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ static inline void dma16_p2(const double* src, double* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
// wait until at most `blocks` * LPB of this wave's vector-memory operations are outstanding (blocks is wave-uniform)
template <int LPB, int QMAX>
__device__ static inline void wait_blocks_p2(int blocks) {
  static_assert(QMAX * LPB <= 63 && QMAX <= 8, "vmcnt is a 6-bit counter");
#define OCS_WB(n) case n: if (n <= QMAX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n <= QMAX ? n : 0) * LPB) : "memory"); break;
  switch (blocks < 0 ? 0 : blocks) {
    OCS_WB(0) OCS_WB(1) OCS_WB(2) OCS_WB(3) OCS_WB(4) OCS_WB(5) OCS_WB(6) OCS_WB(7) OCS_WB(8)
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QMAX * LPB) : "memory"); break;
  }
#undef OCS_WB
}
typedef unsigned v2u_p2 __attribute__((ext_vector_type(2)));
constexpr unsigned kDropP2 = 0xFFFFFFF0u;
constexpr int kNumRecP2 = 0x7FFFFFF0;
struct BufP2 {
  __amdgpu_buffer_rsrc_t r;
  __device__ static inline BufP2 make(double* p) {
    return BufP2{__builtin_amdgcn_make_buffer_rsrc(p, 0, kNumRecP2, 0x00020000)};
  }
  __device__ inline void st(double v, unsigned voff, unsigned soff) const {
#ifndef OCS_P2_X_ST_AUX
#define OCS_P2_X_ST_AUX 0   // cache policy of the state rows of x (tuning builds: scripts/build_variants.sh)
#endif
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_p2, v), r, voff, soff, OCS_P2_X_ST_AUX);
  }
  // non-temporal: for rows nobody reads back soon (the running-objective row: the adjoint pass reads the state rows only)
  __device__ inline void st_nt(double v, unsigned voff, unsigned soff) const {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_p2, v), r, voff, soff, 2);
  }
};

// NKS > 0: the control samples are not read from memory but expanded in the kernel from the coefficients of a dense
// basis, u(:, j) = sum_k v_k B(k, j) (Control/ChebyshevControl.m:35-38), by NUW waves on the matrix cores: NKS k-steps of
// v_mfma_f64_16x16x4_f64 per tile of 16 samples x 16 trajectories (role U below).  Only the step records are streamed then.
// TPW_: trajectories per workgroup, 64 / G by default (every lane of the recursion wave owns one state row).  Half of
// that (G = 1, TPW_ = 32: the upper half of the recursion wave repeats the lower) halves everything a workgroup does
// besides the recursion -- for batches that leave CUs idle, where the other waves of the CU set the pace.
template <int G, int NKS = 0, int TPW_ = 64 / G>
struct P2Cfg {
  static constexpr int D = 8;                        // steps per block
  static constexpr int TPW = TPW_;                   // trajectories per workgroup
  static constexpr int GS = 64 / TPW;                // lane groups of a wave (objective waves: steps per pass)
  static_assert(GS * TPW == 64 && GS % G == 0 && GS <= 4, "lane groups");
  static constexpr int Q = NKS > 0 ? 2 : (G == 1) ? 6 : 8;   // blocks the DMA runs ahead of the preparation (an interval is
                                                     // ~0.3 us: HBM latency needs several; vmcnt counts to 63; the record
                                                     // table alone is shared by all workgroups and sits in L2)
  static constexpr int NSLOT = Q + 3;                // input ring (block j: prepared in interval j-1, read by C in j+1)
  static constexpr int RS = rec_stride(1), SCO = rec_sc_offset(1);
  static constexpr int REC_DBL = D * RS, NREC = REC_DBL / 128;
  static constexpr int U_DBL = 2 * D * TPW, NU = U_DBL / 128;
  static constexpr int SLOT = REC_DBL + U_DBL;
  static constexpr int LPB = NREC + (NKS > 0 ? 0 : NU);
  static constexpr int NCW = (GS == 4) ? 2 : 4;      // objective/store waves
  static constexpr int SPW = D / NCW;                // steps per such wave and block
  static constexpr int NPASS = SPW / GS > 0 ? SPW / GS : 1;
  static constexpr int NT = TPW / 16;                // tiles of 16 trajectories (NKS > 0)
  static constexpr int NUW = NKS > 0 ? (NT >= 2 ? 2 : 1) : 0;   // expansion waves
  static constexpr int TW = NKS > 0 ? NT / NUW : 0;  // tiles per expansion wave
  static constexpr int NWAVE = 4 + NCW + NUW;        // M, S, C.., J, P (, U..)
  static_assert(REC_DBL % 128 == 0 && U_DBL % 128 == 0 && (SPW % GS == 0 || GS > SPW), "block shapes");
  static_assert(NKS == 0 || (2 * D == 16 && NT >= 1 && NKS <= 8), "a block of samples is one 16-row tile");
  // role of a wave (0 M, 1 S, 2.. C, 2 + NCW J, 3 + NCW P, 4 + NCW.. U).  A workgroup's waves are dealt to the four
  // SIMDs in turn (wave w and w + 4 share one): with the expansion waves present the recursion wave gets the SIMD of
  // the nearly idle M wave, the matrix work goes beside the objective waves.
  __device__ static constexpr int role(int w) {
    if (NKS == 0) return w;
    constexpr int M_ = 0, S_ = 1, C_ = 2, J_ = 2 + NCW, P_ = 3 + NCW, U_ = 4 + NCW;
    // (a matrix instruction occupies the SIMD's fp64 datapath for its 64 cycles: an expansion wave gets ONE objective
    //  wave beside it, the two remaining objective waves share the fourth SIMD)
    if (NWAVE == 10) { constexpr int r[10] = {U_, U_ + 1, S_, C_ + 2, C_, C_ + 1, M_, C_ + 3, J_, P_}; return r[w]; }
    constexpr int r[7] = {U_, C_, S_, C_ + 1, J_, P_, M_};
    return r[w];
  }
};
typedef double d4_p2 __attribute__((ext_vector_type(4)));

struct FwdArgsP2 {
  int N, batch;
  const double* REC;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* x0;
  const double* u;
  double* x;
  double* J;
  const int* frozen;   // optional [B]: trajectories with frozen[b] != 0 store nothing
  int ld;              // row distance of the arrays (window of a larger batch) or 0
  int nocost;          // leave the running-objective row of x unwritten (J only)
  const int* gate;     // optional: the launch does nothing if *gate == 0
  // NKS > 0 (u expanded in the kernel; `u` is not read):
  const double* BT = nullptr;   // [2N+1][ldbt]: transposed basis, zero-padded to 4 NKS functions
  const double* v = nullptr;    // [nBasis][B] coefficients (one control)
  int nBasis = 0, ldbt = 0;
};

// UNI: uniform grid -- the step sizes are the same for every step and stay in registers
template <class P, bool OUT_X, bool FRZ, bool UNI, int NKS = 0, int TPW_ = 64 / P::NS>
__global__ __launch_bounds__((P2Cfg<P::NS, NKS, TPW_>::NWAVE * 64)) void k_forward_p2(const FwdArgsP2 a) {
  constexpr int G = P::NS, NAUG = P::NAUG;   // G: state rows
  static_assert(P::NC == 1 && P::NTC == 1, "pipeline kernels: one control, one time coefficient");
  using C_ = P2Cfg<G, NKS, TPW_>;
  constexpr int GS = C_::GS;                 // lane groups of a wave: G rows of TPW trajectories, repeated GS / G times
  constexpr int D = C_::D, TPW = C_::TPW, Q = C_::Q, NSLOT = C_::NSLOT, RS = C_::RS, SCO = C_::SCO;
  constexpr int NCW = C_::NCW, SPW = C_::SPW;
  __shared__ __attribute__((aligned(16))) double inp[NSLOT][C_::SLOT];   // {records | u}
  __shared__ __attribute__((aligned(16))) double zb[2][D][64];            // z_i at the start of a step, per S lane
  __shared__ double ufirst[4][TPW];
  __shared__ __attribute__((aligned(16))) double2 prep[2][D][64];        // (cM, cB) of a step, per S lane (wave P -> S)                                       // control sample at the first node of a block
  __shared__ double dd[2][D][TPW];                                        // objective increments of a block
  const int wave = C_::role(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const size_t B = (size_t)(a.ld ? a.ld : a.batch);
  const int nb = a.N / D;
  // A batch that is not a multiple of the tile: the LAST workgroup takes the last TPW trajectories, overlapping its neighbour --
  // the overlap is computed twice with the same operations and stored twice with the same values (nothing of this kernel is
  // accumulated across trajectories).  The launcher asks for it only with batch >= TPW and an even row distance (the 16-byte
  // DMA chunks stay aligned).
  const int bw_ = blockIdx.x * TPW;
  const int bw = bw_ + TPW <= a.batch ? bw_ : a.batch - TPW;
  if (a.gate && *a.gate == 0) return;
  const uniform_ptr PS = as_uniform(a.ps);
  const size_t colB = (size_t)NAUG * B;

  if (wave == 0) {
    // ---------------- M/P: HBM -> LDS, and the control terms of the next block for S ----------------

    // (ring positions are carried along instead of taken as remainders: a wave's time is its instruction count)
    int cI = 0;   // ring position of the next block to issue (blocks are issued in order)
    auto issue = [&](int j) OCS_INLINE {
      double* dst = &inp[0][0] + cI * C_::SLOT;
      cI = cI + 1 == NSLOT ? 0 : cI + 1;
#pragma unroll
      for (int q = 0; q < C_::NREC; ++q)
        dma16_p2(a.REC + (size_t)j * C_::REC_DBL + q * 128 + 2 * lane, dst + q * 128);
#pragma unroll
      for (int q = 0; q < C_::NU && NKS == 0; ++q) {
        const int e = q * 128 + 2 * lane, row = e / TPW, t2 = e % TPW;
        dma16_p2(a.u + ((size_t)(2 * D * j + 1 + row)) * B + bw + t2, dst + C_::REC_DBL + q * 128);
      }
    };
    int cP = NSLOT - 1;   // ring position of block j-1 of the next call of prepare (calls are in order of j)
    auto prepare = [&](int j) OCS_INLINE {   // block j has landed
      if (NKS > 0) return;   // (the expansion waves do this)
      // the node before the block's first step, for the objective waves (its slot is recycled before they run)
      if (lane < TPW) ufirst[j & 3][lane] = j > 0 ? (&inp[0][0] + cP * C_::SLOT)[C_::REC_DBL + (2 * D - 1) * TPW + lane] : a.u[bw + lane];
      cP = cP + 1 == NSLOT ? 0 : cP + 1;
    };
    P2_BEGIN();
    // Before barrier k the blocks k and k+1 have landed: P prepares block k+1 during interval k.
    for (int j = 0; j <= Q && j < nb; ++j) issue(j);
    {
      const int last = (nb - 1) < Q ? (nb - 1) : Q;   // youngest block issued
      wait_blocks_p2<C_::LPB, Q>(last - (nb > 1 ? 1 : 0));
      prepare(0);
      if (nb > 1) prepare(1);
    }
    for (int k = -1; k <= nb + 1; ++k) {   // interval -1: P prepares block 0
      P2_BARRIER();
      if (k < 0) continue;
      // interval k: issue block k+1+Q (its slot last held block k-2, read by C in interval k-1), wait for block k+2
      if (k + 1 + Q < nb) issue(k + 1 + Q);
      if (k + 2 < nb) {
        const int youngest = (k + 1 + Q) < (nb - 1) ? (k + 1 + Q) : (nb - 1);
        wait_blocks_p2<C_::LPB, Q>(youngest - (k + 2));
        prepare(k + 2);
      }
    }
    P2_END(0);
  } else if (wave == 1) {
    // ---------------- S: the recursion ----------------
    chain_wave_priority();
    const int r = (lane / TPW) % G, tl = lane % TPW, b = bw + tl;   // (GS > G: the upper lane groups repeat the lower)
    const typename P::RowPar rp = P::load_row(ParamSrc{PS, a.pb, a.pmask, B, b}, r);
    const bool fz = FRZ && a.frozen != nullptr && a.frozen[b] != 0;
    const double mh = P::HAS_SHIFT ? P::row_shift(rp) : 0.0;
    double z = a.x0[(size_t)r * B + b] - mh;
    double uprev = NKS > 0 ? 0.0 : a.u[b];   // generic problems: the recursion evaluates F(t, y, u) itself
    double cprev = P::HAS_SHIFT ? P::row_vertex(mh, uprev) : 0.0;
    const uniform_ptr R0 = as_uniform(a.REC);
    const double hU = R0[0], hhU = R0[1], h6U = R0[2];   // step 0's; all steps' on a uniform grid
    int cS = 0;   // ring position of block k
    P2_BEGIN();
    for (int k = -1; k <= nb + 1; ++k) {
      P2_BARRIER();
      if (NKS > 0 && k == -1) {   // the first sample, from the expansion waves
        uprev = ufirst[0][tl];
        cprev = P::HAS_SHIFT ? P::row_vertex(mh, uprev) : 0.0;
      }
      if (!P::HAS_SHIFT && k >= 0 && k < nb) {
        // generic row functions (user problems given as row functions): no shifted form, no prepared terms
        const double* rec = &inp[0][0] + cS * C_::SLOT;
        cS = cS + 1 == NSLOT ? 0 : cS + 1;
        const double* us = rec + C_::REC_DBL + tl;
        double* zw = &zb[k & 1][0][lane];
        struct Ing { double uM, uB, h, hh, h6, tA, tM, tB; };
        auto fetchg = [&](int s) OCS_INLINE {
          Ing v;
          v.uM = us[(2 * s) * TPW];
          v.uB = us[(2 * s + 1) * TPW];
          v.h = rec[RS * s];
          v.hh = rec[RS * s + 1];
          v.h6 = rec[RS * s + 2];
          v.tA = rec[RS * s + 4];
          v.tM = rec[RS * s + 5];
          v.tB = rec[RS * s + 6];
          return v;
        };
        Ing nxt = fetchg(0);
#pragma unroll
        for (int s = 0; s < D; ++s) {
          const Ing c = nxt;
          if (s + 1 < D) nxt = fetchg(s + 1);
          __builtin_amdgcn_sched_barrier(0);
          zw[s * 64] = z;
          const double F1 = P::g_row_f(z, uprev, c.tA, rp);
          double Y = __builtin_fma(c.hh, F1, z);
          const double F2 = P::g_row_f(Y, c.uM, c.tM, rp);
          Y = __builtin_fma(c.hh, F2, z);
          const double F3 = P::g_row_f(Y, c.uM, c.tM, rp);
          Y = __builtin_fma(c.h, F3, z);
          const double F4 = P::g_row_f(Y, c.uB, c.tB, rp);
          z = __builtin_fma(c.h6, F4, __builtin_fma(c.h6, __builtin_fma(2.0, F3, __builtin_fma(2.0, F2, F1)), z));
          uprev = c.uB;
        }
      }
      if (P::HAS_SHIFT && k >= 0 && k < nb) {
        const double* rec = &inp[0][0] + cS * C_::SLOT;
        cS = cS + 1 == NSLOT ? 0 : cS + 1;
        const double2* pw = &prep[k & 1][0][lane];
        double* zw = &zb[k & 1][0][lane];
        struct In { double2 c; double h, hh, h6; };
        auto fetch = [&](int s) OCS_INLINE {
          In v;
          v.c = pw[s * 64];
          if (!UNI) {
            v.h = rec[RS * s];
            v.hh = rec[RS * s + 1];
            v.h6 = rec[RS * s + 2];
          } else {
            v.h = hU; v.hh = hhU; v.h6 = h6U;
          }
          return v;
        };
        In nxt = fetch(0);
#pragma unroll
        for (int s = 0; s < D; ++s) {
          const In c = nxt;
          if (s + 1 < D) nxt = fetch(s + 1);   // the LDS reads of the next step under this step's arithmetic
          __builtin_amdgcn_sched_barrier(0);
          const double cM = c.c.x, cB = c.c.y;   // prepared by wave P: every instruction here costs the pass ~9 cycles per step
          zw[s * 64] = z;
          const double F1 = P::row_f_shifted(z, cprev);
          double Z = __builtin_fma(c.hh, F1, z);
          const double F2 = P::row_f_shifted(Z, cM);
          Z = __builtin_fma(c.hh, F2, z);
          const double F3 = P::row_f_shifted(Z, cM);
          Z = __builtin_fma(c.h, F3, z);
          const double F4 = P::row_f_shifted(Z, cB);
          z = __builtin_fma(c.h6, F4, __builtin_fma(c.h6, __builtin_fma(2.0, F3, __builtin_fma(2.0, F2, F1)), z));
          cprev = cB;
        }
      }
    }
    P2_END(1);
    if (OUT_X && !fz) a.x[((size_t)a.N * NAUG + r) * B + b] = z + mh;   // x(t_N); the other nodes are stored by C
  } else if (wave < 2 + NCW) {
    // ---------------- C: objective increments and the stores of the trajectory ----------------
    const int cw = wave - 2;
    const int csub = lane / TPW, ctl = lane % TPW, b = bw + ctl;   // trajectory fastest
    typename P::RowPar rpr[G];
    double mhr[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
      rpr[q] = P::load_row(ParamSrc{PS, a.pb, a.pmask, B, b}, q);
      mhr[q] = P::HAS_SHIFT ? P::row_shift(rpr[q]) : 0.0;
    }
    const bool fz = FRZ && a.frozen != nullptr && a.frozen[b] != 0;
    const unsigned B8 = (unsigned)(B * 8), col8 = (unsigned)(colB * 8);
    const unsigned vx = fz ? kDropP2 : (unsigned)((size_t)b * 8) + (unsigned)csub * col8;   // node of this lane's step
    const uniform_ptr R0 = as_uniform(a.REC);
    const double hU = R0[0], hhU = R0[1];
    int cC = 0;   // ring position of block k-1
    P2_BEGIN();
    for (int k = -1; k <= nb + 1; ++k) {
      P2_BARRIER();
      if (k >= 1 && k <= nb) {
        const int j = k - 1;
        const double* rec = &inp[0][0] + cC * C_::SLOT;
        cC = cC + 1 == NSLOT ? 0 : cC + 1;
        const double* us = rec + C_::REC_DBL + ctl;
        const double* zr = &zb[j & 1][0][0];
        const double ublk = ufirst[j & 3][ctl];
        const BufP2 bx = BufP2::make(a.x + (size_t)(j * D) * colB);
#pragma unroll
        for (int p = 0; p < C_::NPASS; ++p) {
          const int s0 = cw * SPW + p * GS;    // wave-uniform first step of the pass
          const int s = s0 + csub;             // this lane's step (csub < GS)
          const double wA = rec[RS * s + SCO + 3], wM = rec[RS * s + SCO + 4], wB = rec[RS * s + SCO + 5];
          const double h = UNI ? hU : rec[RS * s], hh = UNI ? hhU : rec[RS * s + 1];
          const double uM = us[(2 * s) * TPW], uB = us[(2 * s + 1) * TPW];
          const double uAl = us[(s > 0 ? 2 * s - 1 : 0) * TPW];
          const double uA = s > 0 ? uAl : ublk;
          double zq[G];
#pragma unroll
          for (int q = 0; q < G; ++q) zq[q] = zr[s * 64 + q * TPW + ctl];
          double d;
          if (P::HAS_SHIFT) {
            const double uA2 = uA * uA, uM2 = uM * uM, uB2 = uB * uB;
            const double cqM = P::control_q(uM2, rpr[0]);
            double q1 = P::control_q(uA2, rpr[0]), q2 = cqM, q3 = cqM, q4 = P::control_q(uB2, rpr[0]);
#pragma unroll
            for (int q = 0; q < G; ++q) {
              const double z = zq[q], mh = mhr[q];
              const double cA = P::row_vertex(mh, uA), cM = P::row_vertex(mh, uM);
              const double F1 = P::row_f_shifted(z, cA);
              const double Z2 = __builtin_fma(hh, F1, z);
              const double F2 = P::row_f_shifted(Z2, cM);
              const double Z3 = __builtin_fma(hh, F2, z);
              const double F3 = P::row_f_shifted(Z3, cM);
              const double Z4 = __builtin_fma(h, F3, z);
              const double y1 = z + mh;
              q1 = P::state_q_acc(y1, q1);
              q2 = P::state_q_acc(Z2 + mh, q2);
              q3 = P::state_q_acc(Z3 + mh, q3);
              q4 = P::state_q_acc(Z4 + mh, q4);
              if (OUT_X) bx.st(y1, vx, (unsigned)s0 * col8 + (unsigned)q * B8);   // x(q, t_i), i = j D + s
            }
            d = __builtin_fma(wA, q1, __builtin_fma(wM, q2 + q3, wB * q4));
          } else {
            // generic row functions: the integrand at the four stage states as the reference sums it (:50)
            const double h6 = rec[RS * s + 2], tA = rec[RS * s + 4], tM = rec[RS * s + 5], tB = rec[RS * s + 6];
            double q1 = 0.0, q2 = 0.0, q3 = 0.0, q4 = 0.0;
#pragma unroll
            for (int q = 0; q < G; ++q) {
              const double y = zq[q];
              const double F1 = P::g_row_f(y, uA, tA, rpr[q]);
              const double Y2 = __builtin_fma(hh, F1, y);
              const double F2 = P::g_row_f(Y2, uM, tM, rpr[q]);
              const double Y3 = __builtin_fma(hh, F2, y);
              const double F3 = P::g_row_f(Y3, uM, tM, rpr[q]);
              const double Y4 = __builtin_fma(h, F3, y);
              q1 += P::g_row_q(y, uA, tA, rpr[q]);
              q2 += P::g_row_q(Y2, uM, tM, rpr[q]);
              q3 += P::g_row_q(Y3, uM, tM, rpr[q]);
              q4 += P::g_row_q(Y4, uB, tB, rpr[q]);
              if (OUT_X) bx.st(y, vx, (unsigned)s0 * col8 + (unsigned)q * B8);
            }
            d = h6 * (__builtin_fma(2.0, q3, __builtin_fma(2.0, q2, q1)) + q4);
          }
          if (csub < GS) dd[j & 1][s][ctl] = d;
        }
      }
    }
    P2_END(wave);
  } else if (wave == 2 + NCW) {
    // ---------------- J: running objective ----------------
    // lane (sg, tl): SPJ = D / GS consecutive steps of trajectory tl, starting at step sg SPJ; the sum over the lanes
    // of a trajectory through the LDS crossbar
    constexpr int SPJ = D / GS;
    const int sg = lane / TPW, tl = lane % TPW, b = bw + tl;
    const bool fz = FRZ && a.frozen != nullptr && a.frozen[b] != 0;
    const bool wc = OUT_X && !a.nocost;
    const unsigned col8 = (unsigned)(colB * 8);
    const unsigned vj = (fz || !wc) ? kDropP2 : (unsigned)(((size_t)G * B + b) * 8) + (unsigned)(sg * SPJ + 1) * col8;
    double carry = 0.0;   // running objective at the first node of the block
    if (wc && !fz && sg == 0) a.x[(size_t)G * B + b] = 0.0;
    P2_BEGIN();
    for (int k = -1; k <= nb + 1; ++k) {
      P2_BARRIER();
      if (k >= 2) {   // k <= nb + 1: block j <= nb - 1
        const int j = k - 2;
        double pre[SPJ];
#pragma unroll
        for (int q = 0; q < SPJ; ++q) pre[q] = dd[j & 1][sg * SPJ + q][tl];
#pragma unroll
        for (int q = 1; q < SPJ; ++q) pre[q] += pre[q - 1];
        // exclusive prefix of the group totals over sg, and the block total
        const double tot = pre[SPJ - 1];
        double excl = 0.0;
        if (GS >= 2) {
          const int below = (lane + 64 - TPW) & 63;                      // the lane of group sg-1
          const double t1 = __shfl(tot, below);
          double inc = tot + (sg >= 1 ? t1 : 0.0);                       // inclusive over two groups
          if (GS == 4) {
            const double t2 = __shfl(inc, (lane + 64 - 2 * TPW) & 63);   // from group sg-2
            inc += (sg >= 2 ? t2 : 0.0);
          }
          const double e = __shfl(inc, below);                           // inclusive sum of the groups below
          excl = sg >= 1 ? e : 0.0;
        }
        const double base = carry + excl;
        const BufP2 bx = BufP2::make(a.x + (size_t)(j * D) * colB);
#pragma unroll
        for (int q = 0; q < SPJ; ++q) bx.st_nt(base + pre[q], vj, (unsigned)q * col8);   // objective at node i+1
        // the next block starts from the value stored for this block's last node (so that J == x(end, end) bit for bit)
        const double lastv = base + pre[SPJ - 1];
        carry = (GS == 1) ? lastv : __shfl(lastv, (GS - 1) * TPW + tl);
      }
    }
    P2_END(wave);
    if (!fz && sg == 0) a.J[b] = carry;
  } else if (wave == 3 + NCW) {
    // ---------------- P: the control terms of the next block for S ----------------
    // P::row_vertex(m_r/2, u) = m_r^2/4 - u for the two new samples of every step, per S lane: two instructions less
    // on the recursion wave, which is bound by its instruction count
    const int r = (lane / TPW) % G, tl = lane % TPW, b = bw + tl;
    const typename P::RowPar rp = P::load_row(ParamSrc{PS, a.pb, a.pmask, B, b}, r);
    const double mh = P::HAS_SHIFT ? P::row_shift(rp) : 0.0;
    int cQ = 0;   // ring position of block j of the next call of prepare
    auto prepare = [&](int j) OCS_INLINE {   // block j has landed (M waits one block ahead of the barrier)
      if (!P::HAS_SHIFT) return;
      const double* us = &inp[0][0] + cQ * C_::SLOT + C_::REC_DBL + tl;
      cQ = cQ + 1 == NSLOT ? 0 : cQ + 1;
      double2* w = &prep[j & 1][0][lane];
#pragma unroll
      for (int s = 0; s < D; ++s)
        if (P::HAS_SHIFT) w[s * 64] = double2{P::row_vertex(mh, us[(2 * s) * TPW]), P::row_vertex(mh, us[(2 * s + 1) * TPW])};
    };
    // block 0 before the first barrier: M's own wait for it is not visible here, so P waits for the data itself --
    // the first barrier below is only passed by M after blocks 0 and 1 have landed; prepare(0) therefore runs in
    // "interval -1": one extra barrier at the head of every wave's loop
    P2_BEGIN();
    for (int k = -1; k <= nb + 1; ++k) {
      P2_BARRIER();
      if (k + 1 < nb) prepare(k + 1);   // read by S in interval k+1; prep[(k+1)&1] was last read in interval k-1
    }
    P2_END(wave);
  } else if constexpr (NKS > 0) {
    // ---------------- U: the control samples of block k+2 from the coefficients ----------------
    // One block = 16 samples (2 D): the tile D[i][n] = sum_k A[i][k] B[k][n] with A[i][k] = B(k, sample 16 j + 1 + i)
    // (wave-uniform data: lane (g, i) = (lane >> 4, lane & 15) holds the entry of k-slot g), B[k][n] = v_k of
    // trajectory n (lane (g, n) holds k-slot g: constant for the whole kernel).  Lane (g, n) ends up with rows 4 m + g
    // of column n in register m and writes them where the samples landed when they came from memory.
    constexpr int TW = C_::TW;
    const int uw = wave - (4 + NCW);
    const int g = lane >> 4, n = lane & 15;
    double vB[TW][NKS];
#pragma unroll
    for (int tq = 0; tq < TW; ++tq)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int kk = 4 * ks + g;
        const double val = a.v[(size_t)(kk < a.nBasis ? kk : 0) * B + bw + 16 * (uw * TW + tq) + n];
        vB[tq][ks] = kk < a.nBasis ? val : 0.0;
      }
    const int nT = 2 * a.N + 1;
    double aA[NKS];
    auto load_a = [&](int j) OCS_INLINE {   // rows of block j, clamped to the grid (block -1: sample 0 in every row)
      int smp = 2 * D * j + 1 + n;
      smp = smp < 0 ? 0 : (smp >= nT ? nT - 1 : smp);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) aA[ks] = a.BT[(size_t)smp * a.ldbt + 4 * ks + g];
    };
    double ulast[TW];   // lanes g == 3: the last sample of the block before (the node before a block's first step)
    int cU = 0;
    auto compute = [&](int j) OCS_INLINE {   // a's rows are those of block j
      double* dst = &inp[0][0] + cU * C_::SLOT + C_::REC_DBL;
      d4_p2 acc[TW];
#pragma unroll
      for (int tq = 0; tq < TW; ++tq) acc[tq] = d4_p2{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int tq = 0; tq < TW; ++tq) acc[tq] = __builtin_amdgcn_mfma_f64_16x16x4f64(aA[ks], vB[tq][ks], acc[tq], 0, 0, 0);
      load_a(j + 1);
      if (j >= 0) {
        cU = cU + 1 == NSLOT ? 0 : cU + 1;
#pragma unroll
        for (int tq = 0; tq < TW; ++tq) {
          const int t2 = 16 * (uw * TW + tq) + n;
          if (g == 3) ufirst[j & 3][t2] = ulast[tq];
          dst[(0 + g) * TPW + t2] = acc[tq].x;
          dst[(4 + g) * TPW + t2] = acc[tq].y;
          dst[(8 + g) * TPW + t2] = acc[tq].z;
          dst[(12 + g) * TPW + t2] = acc[tq].w;
        }
      }
#pragma unroll
      for (int tq = 0; tq < TW; ++tq) ulast[tq] = acc[tq].w;
    };
    P2_BEGIN();
    load_a(-1);
    compute(-1);
    compute(0);
    if (nb > 1) compute(1);
    for (int k = -1; k <= nb + 1; ++k) {
      P2_BARRIER();
      if (k >= 0 && k + 2 < nb) compute(k + 2);   // its slot last held block k-1-Q, read by C long ago
    }
    P2_END(wave);
  }
}


}  // namespace ocs
