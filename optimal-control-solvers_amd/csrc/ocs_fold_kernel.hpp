// ocs_fold_kernel.hpp -- state pass of the forward-backward sweep with the control update folded in (sweep 1: a flag puts
// the default start u0 = lower bound in the place of ControlChar of a costate that does not exist yet).
//
// fb_sweep.m:79-87 alternates  [x, lam] = compute_x_lam(u)  and  u = ControlChar(t, x(t), lam(t))  on the grid.  For the
// problems of the wave-specialised kernels (the registry's logistic family; hipRTC problems given as row functions that
// declare it, OCS_USER_CC_NOX in ocs_user_functor.hpp) ControlChar does not read x, so the control a state pass integrates is a
// function of the PREVIOUS sweep's costate alone: u_k(t) = ControlChar(t, lam_{k-1}(t)), with lam at the half steps by
// pchip (compute_x_lam.m:11-14 / vectorInterpolant).  Writing those 2N+1 samples per instance to memory and reading
// them back was the largest kernel of a sweep (k_control_grid: 24 B per instance and step written, 8 + 8 read);
// here the state pass reads the costate rows instead (8 B) and forms its control samples on the way:
//
//   wave M    streams the step records and the node rows of lam of a block of D = 8 steps HBM -> LDS (LDS-DMA), Q blocks
//             ahead; for nS > 1 also the pchip interval records and the ControlChar time coefficients (at nS = 1 a
//             control wave's step is the same for all lanes and it reads those with scalar loads);
//   waves U   (D / G of them) take one step of a block each, lane (step, trajectory), in two stages one interval apart:
//             the pchip slope at the step's right node (block k+3; each slope is formed once and handed on through
//             LDS), then lam at the half step from the two slopes, ControlChar at the half step and at the right
//             node (block k+2) -> ubuf (the layout the control samples have when they come from memory);
//   waves P, S, C, J   exactly as in k_forward_p2 (ocs_pipeline2_kernel.hpp), reading ubuf instead of a DMA slot.
//
// Interval k (between barriers k and k+1), k = -3 .. nb+1:
//   M: issues block k+5+Q, waits for block k+5     U: slopes of block k+3, samples of block k+2     P: block k+1
//   S: block k     C: block k-1     J: block k-2
// A block's slot is read from interval j-4 (the slopes of block j-1 take its first two nodes) to j+1 (C): NSLOT = Q + 7.
// The arithmetic of S, C and J is k_forward_p2's; the control samples agree with k_control_grid's to round-off (pchip
// slopes by reciprocal + Newton step, the cubic at the middle of an interval in closed form, 1/(2c) of ControlChar
// hoisted: ocs_device_common.hpp, LogisticK::control_char_pre).
#pragma once
#include "ocs_pipeline2_kernel.hpp"

namespace ocs {

#ifndef OCS_FOLD_G1_LDS
#define OCS_FOLD_G1_LDS 0   // tuning builds: 1 = one-row problems take the interval records through LDS as well
#endif
template <int G>
struct FoldCfg {
  static constexpr bool PRL = G > 1 || OCS_FOLD_G1_LDS;      // interval records and ControlChar coefficients through LDS
  static constexpr int D = 8, TPW = 64 / G;
  static constexpr int Q = 5;
  static constexpr int NSLOT = Q + 7;
  static constexpr int RS = rec_stride(1), SCO = rec_sc_offset(1);
  static constexpr int REC_DBL = D * RS;                       // step records
  static constexpr int LAM_DBL = D * 64, NLAM = LAM_DBL / 128; // node rows of lam, [node][row][trajectory]
  // pchip interval records and ControlChar coefficients (grid points 2jD .. 2jD+31: a DMA instruction with lanes 0..15
  // only) of a block -- G > 1 only: on G == 1 the control waves read them with scalar loads from the tables
  static constexpr int PR_DBL = PRL ? D * kPRec : 0;
  static constexpr int TU_DBL = PRL ? 32 : 0;
  static constexpr int LOFF = REC_DBL, POFF = LOFF + LAM_DBL, TOFF = POFF + PR_DBL;
  static constexpr int SLOT = TOFF + TU_DBL;
  static constexpr int LPB = REC_DBL / 128 + NLAM + (PRL ? 2 : 0);
  static constexpr int KHEAD = -3;                             // first interval
  static constexpr int U_DBL = 2 * D * TPW;
  static constexpr int NCW = (G == 4) ? 2 : 4;
  static constexpr int SPW = D / NCW;
  static constexpr int NPASS = SPW / G > 0 ? SPW / G : 1;
  static constexpr int NUW = D / G;                            // control waves: G steps per wave
  static constexpr int NWAVE = 4 + NCW + NUW;
  static_assert(REC_DBL == 128 && (!PRL || PR_DBL == 128) && Q * LPB <= 63, "block shapes");
  // wave -> role: the recursion wave shares its SIMD (waves w, w+4, w+8, w+12) with the light roles only
  enum Role { M_ = 0, S_ = 1, P_ = 2, J_ = 3, C_ = 4, U_ = 5 };
#ifndef OCS_FOLD_ROLEMAP
#define OCS_FOLD_ROLEMAP 0   // tuning builds: 1 = a control wave on the recursion wave's SIMD in the place of M, 2 = two (M and J out)
#endif
  __device__ static constexpr int role(int w) {
    if (G == 1 && OCS_FOLD_ROLEMAP == 1)
      return w == 1 ? S_ : w == 5 ? P_ : w == 9 ? J_ : w == 14 ? M_ : (w == 0 || (w >= 2 && w <= 4)) ? C_ : U_;
    if (G == 1 && OCS_FOLD_ROLEMAP == 2)
      return w == 1 ? S_ : w == 5 ? P_ : w == 10 ? J_ : w == 14 ? M_ : (w == 0 || (w >= 2 && w <= 4)) ? C_ : U_;
    return G == 1 ? (w == 1 ? S_ : w == 5 ? P_ : w == 9 ? J_ : w == 13 ? M_ : (w == 0 || (w >= 2 && w <= 4)) ? C_ : U_)
         : G == 2 ? (w == 0 ? M_ : w == 1 ? S_ : w == 5 ? P_ : w == 9 ? J_ : (w == 2 || w == 3 || w == 4 || w == 6) ? C_ : U_)
                  : (w == 0 ? M_ : w == 1 ? S_ : w == 5 ? P_ : w == 4 ? J_ : (w == 2 || w == 3) ? C_ : U_);
  }
};

struct FwdArgsCC {
  int N, batch;
  const double* REC;
  const double* PR;     // [N][kPRec]
  const double* TU;     // [2N+1 (+32 readable)] ControlChar-side time coefficients (NTU = 1)
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* lb;     // [NC]
  const double* ub;
  const double* x0;
  const double* lam;    // [N+1][G][B] costate of the sweep before
  double* x;
  double* J;
  const int* frozen;
  int nocost;
  const int* gate;
  int first;            // first sweep: the control is the lower bound (u0 = ControlBounds(:,1), fb_sweep.m:23); lam is not used
};

template <class P, bool UNI>
__global__ __launch_bounds__(FoldCfg<P::NS>::NWAVE * 64) void k_forward_cc(const FwdArgsCC a) {
  constexpr int G = P::NS, NAUG = P::NAUG;
  static_assert(P::NC == 1 && P::NTC == 1 && P::NTU == 1 && P::ROW_SEPARABLE && !P::CC_READS_X, "fold: one control, uncoupled rows, ControlChar of the costate alone");
  using C_ = FoldCfg<G>;
  constexpr int D = C_::D, TPW = C_::TPW, Q = C_::Q, NSLOT = C_::NSLOT, RS = C_::RS, SCO = C_::SCO;
  constexpr int NCW = C_::NCW, SPW = C_::SPW;
  __shared__ __attribute__((aligned(16))) double inp[NSLOT][C_::SLOT];
  __shared__ __attribute__((aligned(16))) double ubuf[4][C_::U_DBL];      // control samples of a block: [2s | 2s+1][trajectory]
  __shared__ __attribute__((aligned(16))) double zb[2][D][64];
  __shared__ double ufirst[4][TPW];
  __shared__ double ufirst0[TPW];                                         // u(t_0)
  __shared__ __attribute__((aligned(16))) double dsl[4][D][64];           // pchip slope of lam at the right node of a step
  __shared__ double dnode0[64];                                           // ... at t_0
  __shared__ __attribute__((aligned(16))) double2 prep[2][D][64];
  __shared__ double dd[2][D][TPW];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t B = (size_t)a.batch;
  const int N = a.N, nb = N / D;
  const int bw = tile_base(blockIdx.x, TPW, a.batch);
  if (a.gate && *a.gate == 0) return;
  const uniform_ptr PS = as_uniform(a.ps);
  const size_t colB = (size_t)NAUG * B;
  const int role = C_::role(wave);

  if (role == C_::M_) {
    // ---------------- M: HBM -> LDS ----------------
    // blocks 0 .. nb; block nb is the node t_N alone (its other rows repeat it, its tables are block nb-1's)
    int cI = 0;   // ring position of the next block to issue (blocks are issued in order)
    auto issue = [&](int j) OCS_INLINE {
      double* dst = &inp[0][0] + cI * C_::SLOT;
      cI = cI + 1 == NSLOT ? 0 : cI + 1;
      const int jt = j < nb ? j : nb - 1;
      dma16_p2(a.REC + (size_t)jt * C_::REC_DBL + 2 * lane, dst);
#pragma unroll
      for (int q = 0; q < C_::NLAM; ++q) {
        const int e = q * 128 + 2 * lane, st = e / 64, rr = (e % 64) / TPW, t2 = e % TPW;
        const int node = j * D + st < N ? j * D + st : N;
        dma16_p2(a.lam + ((size_t)node * G + rr) * B + bw + t2, dst + C_::LOFF + q * 128);
      }
      if constexpr (C_::PRL) {
        dma16_p2(a.PR + (size_t)jt * C_::PR_DBL + 2 * lane, dst + C_::POFF);
        if (lane < C_::TU_DBL / 2) dma16_p2(a.TU + (size_t)jt * 2 * D + 2 * lane, dst + C_::TOFF);
      }
    };
    // before barrier k the blocks <= k+4 have landed
    const int youngest0 = (1 + Q) < nb ? (1 + Q) : nb;
    for (int j = 0; j <= youngest0; ++j) issue(j);
    wait_blocks_p2<C_::LPB, Q>(youngest0 - 1);
    P2_BEGIN();
    for (int k = C_::KHEAD; k <= nb + 1; ++k) {
      P2_BARRIER();
      if (k + 5 + Q <= nb) issue(k + 5 + Q);
      if (k + 5 <= nb) {
        const int youngest = (k + 5 + Q) < nb ? (k + 5 + Q) : nb;
        wait_blocks_p2<C_::LPB, Q>(youngest - (k + 5));
      }
    }
    P2_END(wave);
  } else if (role == C_::U_) {
    // ---------------- U: the control samples of block k+2 from the costate of the sweep before ----------------
    const int sub = lane / TPW, tl = lane % TPW, b = bw + tl;
    const typename P::Par par = P::load(ParamSrc{PS, a.pb, a.pmask, B, b});
    const typename P::CCPre ccp = P::cc_pre(par);
    const double lb = a.lb[0], ub = a.ub[0];
    int uwi = 0;   // this wave's index among the U waves (wave-uniform)
    for (int v = 0; v < wave; ++v) uwi += C_::role(v) == C_::U_;
    const int s = uwi * G + sub;              // this lane's step of every block
    // The two stages of an interval are independent (the samples use the slopes of the interval before).  The LDS pipe
    // is the contended unit of this kernel, so whatever the samples stage needs that the slopes stage of the interval
    // before has already read or formed -- the two node values of the step, its own slope, the spacing -- is carried in
    // registers, and on G == 1, where a wave's step is the same for all lanes, the interval records and the ControlChar
    // coefficients are scalar loads from the tables in memory (constant cache) instead of LDS broadcasts.
    const int rowA = s * 64;                                    // node at the left of the step, in its block's slot
    const bool inB = s + 1 < D, inC = s + 2 < D;                // the next two nodes: same slot, or the first rows of the next
    const int rowB = (inB ? s + 1 : s + 1 - D) * 64, rowC = (inC ? s + 2 : s + 2 - D) * 64;
    // ring positions of the blocks k+3, k+4, carried along (a wave's time goes into its instruction count, and the
    // remainders of a ring of 12 were a large part of it); a block outside the horizon still has a position: what is
    // read there is not used
    int c3 = 0, c4 = 1;   // k = KHEAD = -3: blocks 0, 1
    static_assert(C_::KHEAD == -3, "ring positions at the first interval");
    const uniform_ptr PRu = as_uniform(a.PR), TUu = as_uniform(a.TU);
    double cw1[G], cw2[G], cd1[G];          // carried: nodes i, i+1 and the slope at i+1 of the step whose samples come next
    double csv = 0.0, ctuM = 0.0, ctuB = 0.0, ctu0 = 0.0;   // csv: h/8 of that step
#pragma unroll
    for (int r = 0; r < G; ++r) cw1[r] = cw2[r] = cd1[r] = 0.0;
    P2_BEGIN();
    for (int k = C_::KHEAD; k <= nb + 1; ++k) {
      P2_BARRIER();
      const int js = k + 3, jm = k + 2;
      const bool vs = js < nb, vm = (unsigned)jm < (unsigned)nb;
      // ---- slopes: node n = js D + s + 1, the right node of this lane's step of block js ----
      const double* slotS = &inp[0][0] + c3 * C_::SLOT;
      const double* nxtS = &inp[0][0] + c4 * C_::SLOT;
      c3 = c4;
      c4 = c4 + 1 == NSLOT ? 0 : c4 + 1;
      double ih0, ih1, W1, W2, svn, tuMn, tuBn, tu0n, hE0 = 0.0, hE1 = 0.0, hE2 = 0.0, ihE = 0.0;
      const bool ends = vs && (js == 0 || js == nb - 1);   // a block with an end of the grid: the three-point formulas
      if constexpr (!C_::PRL) {
        const int i = vs ? js * D + s : 0;                  // (wave-uniform)
        const uniform_ptr q = PRu + (size_t)i * kPRec;
        ih0 = q[4]; ih1 = q[5]; W1 = q[8]; W2 = q[9]; svn = q[11];
        tu0n = TUu[2 * i]; tuMn = TUu[2 * i + 1]; tuBn = TUu[2 * i + 2];
        if (ends) { hE0 = q[0]; hE1 = q[1]; hE2 = q[2]; ihE = q[3]; }
      } else {
        const double* prS = slotS + C_::POFF + s * kPRec;   // record of interval n-1
        ih0 = prS[4]; ih1 = prS[5]; W1 = prS[8]; W2 = prS[9]; svn = prS[11];
        tu0n = slotS[C_::TOFF + 2 * s]; tuMn = slotS[C_::TOFF + 2 * s + 1]; tuBn = slotS[C_::TOFF + 2 * s + 2];
        if (ends) { hE0 = prS[0]; hE1 = prS[1]; hE2 = prS[2]; ihE = prS[3]; }
      }
      const bool first = jm == 0 && s == 0;
      double wa[G], wb[G], wc[G], d0[G];
#pragma unroll
      for (int r = 0; r < G; ++r) {
        const int o = C_::LOFF + r * TPW + tl;
        wa[r] = slotS[o + rowA];
        wb[r] = (inB ? slotS : nxtS)[o + rowB];
        wc[r] = (inC ? slotS : nxtS)[o + rowC];   // (beyond t_N: a valid address, the value is not used)
        // the slope at the left node of the samples' step: the neighbouring wave's, of the interval before
        d0[r] = (s > 0 ? dsl[jm & 3][s > 0 ? s - 1 : 0] : dsl[(jm + 3) & 3][D - 1])[r * TPW + tl];
      }
      if (jm == 0) {   // (wave-uniform)
#pragma unroll
        for (int r = 0; r < G; ++r) d0[r] = first ? dnode0[r * TPW + tl] : d0[r];
      }
      double dn[G], lmid[G];
#pragma unroll
      for (int r = 0; r < G; ++r) {
        const double sa = (wb[r] - wa[r]) * ih0, sb = (wc[r] - wb[r]) * ih1;
        dn[r] = pchip_interior_f(sa, sb, W1, W2);
        lmid[r] = __builtin_fma(csv, d0[r] - cd1[r], 0.5 * (cw1[r] + cw2[r]));   // the cubic at the middle of its interval
      }
      const bool u0lb = a.first != 0;   // (whatever lam holds then -- possibly NaN -- is dropped by the selects)
      const double uMc = P::control_char_pre(ctuM, lmid, ccp, lb, ub), uBc = P::control_char_pre(ctuB, cw2, ccp, lb, ub);
      const double uM = u0lb ? lb : uMc, uB = u0lb ? lb : uBc;
      if (vm) {
        double* us = &ubuf[jm & 3][tl];
        us[(2 * s) * TPW] = uM;
        us[(2 * s + 1) * TPW] = uB;
        if (first) {   // the first node of the horizon
          const double u0c = P::control_char_pre(ctu0, cw1, ccp, lb, ub);
          ufirst0[tl] = u0lb ? lb : u0c;
        }
      }
      if (ends) {
        const int n = js * D + s + 1;
#pragma unroll
        for (int r = 0; r < G; ++r) {
          const double sa = (wb[r] - wa[r]) * ih0, sb = (wc[r] - wb[r]) * ih1;
          if (n == N) {
            const double wz = slotS[C_::LOFF + r * TPW + tl + (D - 2) * 64];   // node N-2 (n == N: the last step of the last block)
            dn[r] = pchip_end_pl(hE1, hE0, sa, (wa[r] - wz) * ihE);
          }
          if (n == 1) dnode0[r * TPW + tl] = pchip_end_pl(hE1, hE2, sa, sb);   // the slope at t_0
        }
      }
#pragma unroll
      for (int r = 0; r < G; ++r) {
        if (vs) dsl[js & 3][s][r * TPW + tl] = dn[r];
        cw1[r] = wa[r];
        cw2[r] = wb[r];
        cd1[r] = dn[r];
      }
      csv = svn;
      ctuM = tuMn;
      ctuB = tuBn;
      ctu0 = tu0n;
    }
    P2_END(wave);
  } else if (role == C_::S_) {
    // ---------------- S: the recursion (k_forward_p2's) ----------------
    chain_wave_priority();
    const int r = lane / TPW, tl = lane % TPW, b = bw + tl;
    const typename P::RowPar rp = P::load_row(ParamSrc{PS, a.pb, a.pmask, B, b}, r);
    const bool fz = a.frozen != nullptr && a.frozen[b] != 0;
    const double mh = P::row_shift(rp);
    double z = a.x0[(size_t)r * B + b] - mh;
    double cprev = 0.0;
    const uniform_ptr R0 = as_uniform(a.REC);
    const double hU = R0[0], hhU = R0[1], h6U = R0[2];
    P2_BEGIN();
    int cS = 0;   // ring position of block k
    for (int k = C_::KHEAD; k <= nb + 1; ++k) {
      P2_BARRIER();
      if (k == -1) cprev = P::row_vertex(mh, ufirst0[tl]);   // written by U in interval -2
      if (k >= 0 && k < nb) {
        const double* rec = &inp[0][0] + cS * C_::SLOT;
        cS = cS + 1 == NSLOT ? 0 : cS + 1;
        const double2* pw = &prep[k & 1][0][lane];
        double* zw = &zb[k & 1][0][lane];
        struct In { double2 c; double h, hh, h6; };
        auto fetch = [&](int s) OCS_INLINE {
          In v;
          v.c = pw[s * 64];
#ifndef OCS_FOLD_SREC
#define OCS_FOLD_SREC 0   // tuning builds: 1 = the recursion wave takes the step sizes with scalar loads from the record table
#endif
          if (!UNI && OCS_FOLD_SREC) {
            const uniform_ptr rq = R0 + (size_t)(k * D + s) * RS;
            v.h = rq[0];
            v.hh = rq[1];
            v.h6 = rq[2];
          } else if (!UNI) {
            v.h = rec[RS * s];
            v.hh = rec[RS * s + 1];
            v.h6 = rec[RS * s + 2];
          } else {
            v.h = hU; v.hh = hhU; v.h6 = h6U;
          }
          return v;
        };
#ifndef OCS_FOLD_SPF
#define OCS_FOLD_SPF 1   // steps the recursion wave reads its inputs ahead of their use (tuning builds: 2)
#endif
        In nxt = fetch(0);
        In nx2 = nxt;
        if (OCS_FOLD_SPF == 2) nx2 = fetch(1);
        if constexpr (P::HAS_SHIFT) {
#pragma unroll
          for (int s = 0; s < D; ++s) {
            const In c = nxt;
            if (OCS_FOLD_SPF == 2) {
              nxt = nx2;
              if (s + 2 < D) nx2 = fetch(s + 2);
            } else if (s + 1 < D) {
              nxt = fetch(s + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            const double cM = c.c.x, cB = c.c.y;
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
        } else {
          // row functions (user problems): no shifted form -- wave P hands the samples on as they are (row_vertex is the
          // identity there), the row function gets the stage's time coefficient from the step record
#pragma unroll
          for (int s = 0; s < D; ++s) {
            const In c = nxt;
            if (s + 1 < D) nxt = fetch(s + 1);
            const double tA = rec[RS * s + 4], tM = rec[RS * s + 5], tB = rec[RS * s + 6];
            __builtin_amdgcn_sched_barrier(0);
            const double uM = c.c.x, uB = c.c.y;
            zw[s * 64] = z;
            const double F1 = P::g_row_f(z, cprev, tA, rp);
            double Y = __builtin_fma(c.hh, F1, z);
            const double F2 = P::g_row_f(Y, uM, tM, rp);
            Y = __builtin_fma(c.hh, F2, z);
            const double F3 = P::g_row_f(Y, uM, tM, rp);
            Y = __builtin_fma(c.h, F3, z);
            const double F4 = P::g_row_f(Y, uB, tB, rp);
            z = __builtin_fma(c.h6, F4, __builtin_fma(c.h6, __builtin_fma(2.0, F3, __builtin_fma(2.0, F2, F1)), z));
            cprev = uB;
          }
        }
      }
    }
    P2_END(wave);
    if (!fz) a.x[((size_t)N * NAUG + r) * B + b] = z + mh;
  } else if (role == C_::C_) {
    // ---------------- C: objective increments and the stores of the trajectory (k_forward_p2's) ----------------
    int cw = 0;
    for (int v = 0; v < wave; ++v) cw += C_::role(v) == C_::C_;
    const int csub = lane / TPW, ctl = lane % TPW, b = bw + ctl;
    typename P::RowPar rpr[G];
    double mhr[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
      rpr[q] = P::load_row(ParamSrc{PS, a.pb, a.pmask, B, b}, q);
      mhr[q] = P::row_shift(rpr[q]);
    }
    const bool fz = a.frozen != nullptr && a.frozen[b] != 0;
    const unsigned B8 = (unsigned)(B * 8), col8 = (unsigned)(colB * 8);
    const unsigned vx = fz ? kDropP2 : (unsigned)((size_t)b * 8) + (unsigned)csub * col8;
    const uniform_ptr R0 = as_uniform(a.REC);
    const double hU = R0[0], hhU = R0[1];
    int cC = 0;   // ring position of block k-1
    P2_BEGIN();
    for (int k = C_::KHEAD; k <= nb + 1; ++k) {
      P2_BARRIER();
      if (k >= 1 && k <= nb) {
        const int j = k - 1;
        const double* rec = &inp[0][0] + cC * C_::SLOT;
        cC = cC + 1 == NSLOT ? 0 : cC + 1;
        const double* us = &ubuf[j & 3][ctl];
        const double* zr = &zb[j & 1][0][0];
        const double ublk = ufirst[j & 3][ctl];
        const BufP2 bx = BufP2::make(a.x + (size_t)(j * D) * colB);
#pragma unroll
        for (int p = 0; p < C_::NPASS; ++p) {
          const int s0 = cw * SPW + p * G;
          const int s = s0 + csub;
          const double wA = rec[RS * s + SCO + 3], wM = rec[RS * s + SCO + 4], wB = rec[RS * s + SCO + 5];
          const double h = UNI ? hU : rec[RS * s], hh = UNI ? hhU : rec[RS * s + 1];
          const double uM = us[(2 * s) * TPW], uB = us[(2 * s + 1) * TPW];
          const double uAl = us[(s > 0 ? 2 * s - 1 : 0) * TPW];
          const double uA = s > 0 ? uAl : ublk;
          double zq[G];
#pragma unroll
          for (int q = 0; q < G; ++q) zq[q] = zr[s * 64 + q * TPW + ctl];
          double d;
          if constexpr (P::HAS_SHIFT) {
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
              bx.st(y1, vx, (unsigned)s0 * col8 + (unsigned)q * B8);
            }
            d = __builtin_fma(wA, q1, __builtin_fma(wM, q2 + q3, wB * q4));
          } else {
            // row functions: the integrand at the four stage states as the reference sums it (RK4Integrator.m:50),
            // k_forward_p2's generic branch
            const double h6 = rec[RS * s + 2], tA = rec[RS * s + 4], tM = rec[RS * s + 5], tB = rec[RS * s + 6];
            (void)wA; (void)wM; (void)wB;
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
              bx.st(y, vx, (unsigned)s0 * col8 + (unsigned)q * B8);
            }
            d = h6 * (__builtin_fma(2.0, q3, __builtin_fma(2.0, q2, q1)) + q4);
          }
          if (csub < G) dd[j & 1][s][ctl] = d;
        }
      }
    }
    P2_END(wave);
  } else if (role == C_::J_) {
    // ---------------- J: running objective (k_forward_p2's) ----------------
    constexpr int SPJ = D / G;
    const int sg = lane / TPW, tl = lane % TPW, b = bw + tl;
    const bool fz = a.frozen != nullptr && a.frozen[b] != 0;
    const bool wc = !a.nocost;
    const unsigned col8 = (unsigned)(colB * 8);
    const unsigned vj = (fz || !wc) ? kDropP2 : (unsigned)(((size_t)G * B + b) * 8) + (unsigned)(sg * SPJ + 1) * col8;
    double carry = 0.0;
    if (wc && !fz && sg == 0) a.x[(size_t)G * B + b] = 0.0;
    P2_BEGIN();
    for (int k = C_::KHEAD; k <= nb + 1; ++k) {
      P2_BARRIER();
      if (k >= 2) {
        const int j = k - 2;
        double pre[SPJ];
#pragma unroll
        for (int q = 0; q < SPJ; ++q) pre[q] = dd[j & 1][sg * SPJ + q][tl];
#pragma unroll
        for (int q = 1; q < SPJ; ++q) pre[q] += pre[q - 1];
        const double tot = pre[SPJ - 1];
        double excl = 0.0;
        if (G >= 2) {
          const int below = (lane + 64 - TPW) & 63;
          const double t1 = __shfl(tot, below);
          double inc = tot + (sg >= 1 ? t1 : 0.0);
          if (G == 4) {
            const double t2 = __shfl(inc, (lane + 64 - 2 * TPW) & 63);
            inc += (sg >= 2 ? t2 : 0.0);
          }
          const double e = __shfl(inc, below);
          excl = sg >= 1 ? e : 0.0;
        }
        const double base = carry + excl;
        const BufP2 bx = BufP2::make(a.x + (size_t)(j * D) * colB);
#pragma unroll
        for (int q = 0; q < SPJ; ++q) bx.st_nt(base + pre[q], vj, (unsigned)q * col8);
        const double lastv = base + pre[SPJ - 1];
        carry = (G == 1) ? lastv : __shfl(lastv, (G - 1) * TPW + tl);
      }
    }
    P2_END(wave);
    if (!fz && sg == 0) a.J[b] = carry;
  } else {
    // ---------------- P: the control terms of the next block for S, and the node before a block for C ----------------
    const int r = lane / TPW, tl = lane % TPW, b = bw + tl;
    const typename P::RowPar rp = P::load_row(ParamSrc{PS, a.pb, a.pmask, B, b}, r);
    const double mh = P::row_shift(rp);
    P2_BEGIN();
    for (int k = C_::KHEAD; k <= nb + 1; ++k) {
      P2_BARRIER();
      const int j = k + 1;   // read by S in interval k+1; its samples were written by U in interval k-1
      if (j < 0 || j >= nb) continue;
      const double* us = &ubuf[j & 3][tl];
      double2* w = &prep[j & 1][0][lane];
#pragma unroll
      for (int s = 0; s < D; ++s)
        w[s * 64] = double2{P::row_vertex(mh, us[(2 * s) * TPW]), P::row_vertex(mh, us[(2 * s + 1) * TPW])};
      if (lane < TPW) ufirst[j & 3][lane] = j > 0 ? ubuf[(j - 1) & 3][(2 * D - 1) * TPW + lane] : ufirst0[lane];
    }
    P2_END(wave);
  }
}

}  // namespace ocs
