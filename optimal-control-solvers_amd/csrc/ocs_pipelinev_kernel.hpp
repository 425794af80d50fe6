// ocs_pipelinev_kernel.hpp -- the state pass (RK4Integrator.m:28-56) wave-specialised for ANY OCProblem (coupled rows,
// several controls): k_forward_p2's division of labour (ocs_pipeline2_kernel.hpp) with the state vector of a trajectory
// in one lane instead of one row per lane.
//
//   wave M   streams control samples and step records HBM -> LDS (LDS-DMA), Q blocks ahead;
//   wave S   the recursion y_{i+1} = Phi_i(y_i): per step the state rows of four F evaluations and the updates of
//            :40-51 -- nothing else; publishes y_i through LDS;
//   waves C  (4) take the steps of the block S finished one interval earlier: the four stage evaluations again, this
//            time for the objective row F(end) (:50 sums F(end) of the four stages), and the stores of x(t_i);
//   wave J   prefix-sums the objective increments of the block before that, stores the cost row and J.
//
// One LDS-only barrier per block of D = 8 steps; interval k: M issues block k+1+Q and waits for block k+2, S runs
// block k, C block k-1, J block k-2.  A workgroup holds 64 trajectories (lane = trajectory in every wave).
// Results: the state rows are the lane kernel's operations in the lane kernel's order (bit-equal); the running
// objective is summed per step as h/6 (F1 + 2 F2 + 2 F3 + F4)(end) and prefix-summed per block of 8 steps instead of
// updated in place (:50-51): round-off level, J == x(end, end) bit for bit.
// Functor interface: P::Par / load / F / Fx (ocs_problems.hpp, ocs_user_functor.hpp), NTC = 1.
#pragma once
#include "ocs_pipeline2_kernel.hpp"

namespace ocs {

template <int NS, int NC>
struct PVCfg {
  static constexpr int D = 8, Q = NC == 1 ? 4 : 2, NSLOT = Q + 3, NCW = 4, SPW = D / NCW, NWAVE = 3 + NCW;   // M, S, C x 4, J
  static constexpr int RS = rec_stride(1);
  static constexpr int REC_DBL = D * RS, NREC = REC_DBL / 128;
  static constexpr int U_DBL = 2 * D * NC * 64, NU = U_DBL / 128;   // samples 2 D j + 1 .. 2 D j + 2 D, [sample][control][lane]
  static constexpr int SLOT = REC_DBL + U_DBL;
  static constexpr int LPB = NREC + NU;
  static_assert(REC_DBL % 128 == 0 && LPB * Q <= 63, "block shapes");
};

template <class P, bool OUT_X>
__global__ __launch_bounds__((PVCfg<P::NS, P::NC>::NWAVE * 64)) void k_forward_pv(const FwdArgsP2 a) {
  constexpr int NS = P::NS, NC = P::NC, NAUG = P::NAUG;
  static_assert(P::NTC == 1, "one time coefficient");
  using C_ = PVCfg<NS, NC>;
  constexpr int D = C_::D, Q = C_::Q, NSLOT = C_::NSLOT, RS = C_::RS, NCW = C_::NCW, SPW = C_::SPW;
  __shared__ __attribute__((aligned(16))) double inp[NSLOT][C_::SLOT];   // {records | u}
  __shared__ double yb[2][D][NS][64];                                      // y_i at the start of a step
  __shared__ double ufirst[4][NC][64];                                     // control sample at the first node of a block
  __shared__ double dd[2][D][64];                                          // objective increments of a block
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t B = (size_t)a.batch;
  const int nb = a.N / D;
  const int bw = tile_base(blockIdx.x, 64, a.batch);   // (a ragged last tile overlaps its neighbour, ocs_device_common.hpp)
  const int b0 = bw + lane;
  const bool valid = b0 < a.batch;
  const int b = valid ? b0 : a.batch - 1;
  if (a.gate && *a.gate == 0) return;
  const size_t colB = (size_t)NAUG * B;

  if (wave == 0) {
    // ---------------- M: HBM -> LDS ----------------
    int cI = 0, cP = NSLOT - 1;
    auto issue = [&](int j) OCS_INLINE {
      double* dst = &inp[0][0] + cI * C_::SLOT;
      cI = cI + 1 == NSLOT ? 0 : cI + 1;
#pragma unroll
      for (int q = 0; q < C_::NREC; ++q)
        dma16_p2(a.REC + (size_t)j * C_::REC_DBL + q * 128 + 2 * lane, dst + q * 128);
#pragma unroll
      for (int q = 0; q < C_::NU; ++q) {
        // piece q: 128 doubles = two rows of 64 lanes; a lane copies two consecutive trajectories of one row
        const int e = q * 128 + 2 * lane, row = e / 64, t2 = e % 64;   // row = sample * NC + control
        dma16_p2(a.u + ((size_t)(2 * D * j + 1) * NC + row) * B + bw + t2, dst + C_::REC_DBL + q * 128);   // (whole tiles)
      }
    };
    auto prepare = [&](int j) OCS_INLINE {   // block j has landed: the node before its first step, for the objective waves
#pragma unroll
      for (int c = 0; c < NC; ++c)
        ufirst[j & 3][c][lane] = j > 0 ? (&inp[0][0] + cP * C_::SLOT)[C_::REC_DBL + ((2 * D - 1) * NC + c) * 64 + lane]
                                       : a.u[(size_t)c * B + b];
      cP = cP + 1 == NSLOT ? 0 : cP + 1;
    };
    for (int j = 0; j <= Q && j < nb; ++j) issue(j);
    {
      const int last = (nb - 1) < Q ? (nb - 1) : Q;
      wait_blocks_p2<C_::LPB, Q>(last - (nb > 1 ? 1 : 0));
      prepare(0);
      if (nb > 1) prepare(1);
    }
    for (int k = -1; k <= nb + 1; ++k) {
      lds_barrier_p2_();
      if (k < 0) continue;
      if (k + 1 + Q < nb) issue(k + 1 + Q);
      if (k + 2 < nb) {
        const int youngest = (k + 1 + Q) < (nb - 1) ? (k + 1 + Q) : (nb - 1);
        wait_blocks_p2<C_::LPB, Q>(youngest - (k + 2));
        prepare(k + 2);
      }
    }
  } else if (wave == 1) {
    // ---------------- S: the recursion ----------------
    chain_wave_priority();
    const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
    double y[NS], uprev[NC];
#pragma unroll
    for (int k = 0; k < NS; ++k) y[k] = a.x0[(size_t)k * B + b];
#pragma unroll
    for (int c = 0; c < NC; ++c) uprev[c] = a.u[(size_t)c * B + b];
    int cS = 0;
    for (int k = -1; k <= nb + 1; ++k) {
      lds_barrier_p2_();
      if (k >= 0 && k < nb) {
        const double* rec = &inp[0][0] + cS * C_::SLOT;
        cS = cS + 1 == NSLOT ? 0 : cS + 1;
        const double* us = rec + C_::REC_DBL + lane;
#pragma unroll
        for (int s = 0; s < D; ++s) {
          double uM[NC], uB[NC];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            uM[c] = us[((2 * s) * NC + c) * 64];
            uB[c] = us[((2 * s + 1) * NC + c) * 64];
          }
          const double h = rec[RS * s], hh = rec[RS * s + 1], h6 = rec[RS * s + 2];
          const double tA = rec[RS * s + 4], tM = rec[RS * s + 5], tB = rec[RS * s + 6];
#pragma unroll
          for (int r = 0; r < NS; ++r) yb[k & 1][s][r][lane] = y[r];
          double F1[NS], F2[NS], F3[NS], F4[NS], Y[NS];
          P::Fx(&tA, y, uprev, p, F1);                                           // :39
#pragma unroll
          for (int r = 0; r < NS; ++r) Y[r] = __builtin_fma(hh, F1[r], y[r]);    // :40
          P::Fx(&tM, Y, uM, p, F2);                                              // :42
#pragma unroll
          for (int r = 0; r < NS; ++r) Y[r] = __builtin_fma(hh, F2[r], y[r]);    // :43
          P::Fx(&tM, Y, uM, p, F3);                                              // :45
#pragma unroll
          for (int r = 0; r < NS; ++r) Y[r] = __builtin_fma(h, F3[r], y[r]);     // :46
          P::Fx(&tB, Y, uB, p, F4);                                              // :48
#pragma unroll
          for (int r = 0; r < NS; ++r)                                           // :50-51
            y[r] = __builtin_fma(h6, __builtin_fma(2.0, F3[r], __builtin_fma(2.0, F2[r], F1[r])) + F4[r], y[r]);
#pragma unroll
          for (int c = 0; c < NC; ++c) uprev[c] = uB[c];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (OUT_X && valid && !(a.frozen != nullptr && a.frozen[b] != 0)) {
#pragma unroll
      for (int r = 0; r < NS; ++r) a.x[((size_t)a.N * NAUG + r) * B + b] = y[r];   // x(t_N); the other nodes come from C
    }
  } else if (wave < 2 + NCW) {
    // ---------------- C: objective increments and the stores of the trajectory ----------------
    const int cw = wave - 2;
    const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
    const unsigned B8 = (unsigned)(B * 8), col8 = (unsigned)(colB * 8);
    const bool fz = a.frozen != nullptr && a.frozen[b] != 0;   // (fb_sweep: a converged instance stores nothing)
    const unsigned vx = (valid && !fz) ? (unsigned)((size_t)b * 8) : kDropP2;
    int cC = 0;
    for (int k = -1; k <= nb + 1; ++k) {
      lds_barrier_p2_();
      if (k >= 1 && k <= nb) {
        const int j = k - 1;
        const double* rec = &inp[0][0] + cC * C_::SLOT;
        cC = cC + 1 == NSLOT ? 0 : cC + 1;
        const double* us = rec + C_::REC_DBL + lane;
        const BufP2 bx = BufP2::make(a.x + (size_t)(j * D) * colB);
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
          const int s = cw * SPW + q;   // wave-uniform step of the block
          double y[NS], uA[NC], uM[NC], uB[NC];
#pragma unroll
          for (int r = 0; r < NS; ++r) y[r] = yb[j & 1][s][r][lane];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            uA[c] = s > 0 ? us[((2 * s - 1) * NC + c) * 64] : ufirst[j & 3][c][lane];
            uM[c] = us[((2 * s) * NC + c) * 64];
            uB[c] = us[((2 * s + 1) * NC + c) * 64];
          }
          const double h = rec[RS * s], hh = rec[RS * s + 1], h6 = rec[RS * s + 2];
          const double tA = rec[RS * s + 4], tM = rec[RS * s + 5], tB = rec[RS * s + 6];
          double F1[NAUG], F2[NAUG], F3[NAUG], F4[NAUG], Y[NS];
          P::F(&tA, y, uA, p, F1);
#pragma unroll
          for (int r = 0; r < NS; ++r) Y[r] = __builtin_fma(hh, F1[r], y[r]);
          P::F(&tM, Y, uM, p, F2);
#pragma unroll
          for (int r = 0; r < NS; ++r) Y[r] = __builtin_fma(hh, F2[r], y[r]);
          P::F(&tM, Y, uM, p, F3);
#pragma unroll
          for (int r = 0; r < NS; ++r) Y[r] = __builtin_fma(h, F3[r], y[r]);
          P::F(&tB, Y, uB, p, F4);
          dd[j & 1][s][lane] = h6 * (__builtin_fma(2.0, F3[NS], __builtin_fma(2.0, F2[NS], F1[NS])) + F4[NS]);   // :50, row end
          if (OUT_X) {
#pragma unroll
            for (int r = 0; r < NS; ++r) bx.st(y[r], vx, (unsigned)s * col8 + (unsigned)r * B8);   // x(r, t_i), i = j D + s
          }
        }
      }
    }
  } else {
    // ---------------- J: running objective ----------------
    const bool wc = OUT_X && !a.nocost;
    const bool fz = a.frozen != nullptr && a.frozen[b] != 0;
    const unsigned col8 = (unsigned)(colB * 8);
    const unsigned vj = (valid && wc && !fz) ? (unsigned)(((size_t)NS * B + b) * 8) + col8 : kDropP2;
    double carry = 0.0;
    if (wc && valid && !fz) a.x[(size_t)NS * B + b] = 0.0;
    for (int k = -1; k <= nb + 1; ++k) {
      lds_barrier_p2_();
      if (k >= 2) {
        const int j = k - 2;
        const BufP2 bx = BufP2::make(a.x + (size_t)(j * D) * colB);
#pragma unroll
        for (int s = 0; s < D; ++s) {
          carry += dd[j & 1][s][lane];
          bx.st_nt(carry, vj, (unsigned)s * col8);   // objective at node j D + s + 1
        }
      }
    }
    if (valid && !fz) a.J[b] = carry;
  }
}

}  // namespace ocs
