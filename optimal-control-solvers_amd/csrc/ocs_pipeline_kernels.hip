// ocs_pipeline_kernels.hip -- wave-specialised ("pipeline") mapping of the RK4 state / adjoint
// passes for row-separable problems.
//
// What bounds the passes at the BASELINE batch (4096 trajectories) is not HBM bandwidth but two
// latencies of a nearly empty chip: the fp64 issue rate of ONE wave (an instruction every ~5.4
// cycles) and the memory-level parallelism of a few waves (bytes in flight per CU over a ~1 us round
// trip; in-order vmcnt makes every short-distance load wait behind the long-distance prefetches).
// This mapping attacks both.  A workgroup owns 64/G trajectories (G = nS lanes per trajectory, one
// state row per lane as in ocs_rowsplit_kernels.hip) and splits a step's WORK over its waves, which
// run concurrently on the SIMDs of one CU and hand data to each other through LDS in blocks of D
// steps (one LDS-only barrier per block):
//
//   (roles are cut where the hand-off is narrow: LDS stores cost a lone wave ~13 cycles per 512 B, as
//    much as two fp64 instructions, so R hands over 3 values per step and A 3, not 8 and 4)
//   forward   wave M: memory wave.  Streams control samples and step records HBM -> LDS with LDS-DMA
//                     (global_load_lds_dwordx4, 1 KiB per instruction, no VGPR staging), Q blocks
//                     ahead of the compute waves; it is the only wave that waits on loads.
//             wave S: the state recursion (F1..F4, Y2..Y4, y_{i+1}) out of LDS; stores the state
//                     rows; publishes the four stage states of every step.
//             wave C: integrates the objective from the published stage states, reduces it over
//                     the rows of a trajectory (DPP) and stores the cost row / J.
//   backward  wave M: streams checkpoints, control samples and records.
//             wave R: recomputes the stage states Y2..Y4 from the checkpoints, publishes them.
//             wave A: the adjoint recursion (dJdk, lam); stores lam; publishes k1, k2+k3, k4.
//             wave D: assembles the dJdu columns from k1..k4, reduces over rows, stores them.
//
// The serial critical path per step shrinks to the longest role, and no compute wave ever waits for
// HBM.  Arithmetic inside the roles is the row-split kernels' (same formulas, same association), so
// the two mappings agree bit for bit; layouts and semantics are those of k_forward / k_backward
// (RK4Integrator.m:28-121).  Restrictions (the launcher falls back to row-split otherwise):
// nSTEPS a multiple of D, batch a multiple of 64/G.
#include "ocs_device_common.hpp"
#include "ocs_internal.hpp"
#include "ocs_problems.hpp"
#include <cstdlib>
#ifdef OCS_PL_STAMPS
#include <cstdio>
#include <vector>
#endif

namespace ocs {

static inline int hip_rc5(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

template <int CTRL>
__device__ static inline double dpp_quad_pl(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  const int hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi2, lo2);
}
template <int G>
__device__ static inline double group_sum_pl(double v) {
  static_assert(G == 1 || G == 2 || G == 4, "group size");
  if (G >= 2) v += dpp_quad_pl<0xB1>(v);
  if (G == 4) v += dpp_quad_pl<0x4E>(v);
  return v;
}

// Hand-off barrier: only LDS traffic has to be complete.  __syncthreads() would also drain vmcnt,
// i.e. every global store still in flight.
__device__ static inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-byte-per-lane LDS-DMA: lane l copies src_l[0..1] to lds_base[2l..2l+1]
__device__ static inline void dma16(const double* src, double* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
// wait until at most `blocks` * LPB of this wave's vector-memory operations are outstanding, i.e.
// until every DMA except those of the youngest `blocks` blocks has landed (blocks is wave-uniform)
template <int LPB>
__device__ static inline void wait_blocks(int blocks) {
  static_assert(2 * LPB <= 63, "vmcnt is a 6-bit counter");
  if (blocks <= 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (blocks == 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPB) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPB) : "memory");
}

template <int G, bool BWD>
struct PLCfg {
  static constexpr int D = 8;                          // steps per hand-off block (BASELINE N = 1000 = 125 blocks)
  static constexpr int TPW = 64 / G;                   // trajectories per workgroup
  static constexpr int Q = 3;                          // blocks the memory wave runs ahead (wait_blocks: <= 2 younger)
  static constexpr int LAG = BWD ? 2 : 1;              // intervals between the first and the last reader of a slot
  static constexpr int NSLOT = Q + LAG + 1;            // input ring slots (see the schedules below)
  static constexpr int RS = rec_stride(1);             // doubles per step record (NTC = 1)
  static constexpr int SCO = rec_sc_offset(1);         // offset of the step constants in a record
  static constexpr int REC_DBL = D * RS;               // records of a block
  static constexpr int NREC = REC_DBL / 128;           // DMA instructions for them
  static constexpr int U_DBL = 2 * D * TPW;            // control samples of a block
  static constexpr int NU = U_DBL / 128;
  static constexpr int X_DBL = BWD ? D * 64 : 0;       // checkpoint rows of a block (backward only)
  static constexpr int NX = X_DBL / 128;
  static constexpr int SLOT = REC_DBL + U_DBL + X_DBL;
  static constexpr int LPB = NREC + NU + NX;           // DMA instructions per block
  static_assert(REC_DBL % 128 == 0 && U_DBL % 128 == 0, "blocks must be whole DMA instructions");
};

#ifdef OCS_PL_STAMPS
#define PL_T() __builtin_amdgcn_s_memtime()
#else
#define PL_T() 0LL
#endif

struct FwdArgsPL {
  int N, batch;
  const double* REC;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* x0;
  const double* u;
  double* x;
  double* J;
  long long* dbg;  // diagnostic build (-DOCS_PL_STAMPS) only: per-workgroup cycle sums; nullptr otherwise
  const int* frozen;  // optional [B]: trajectories with frozen[b] != 0 store nothing (as in FwdArgs)
  double* dump;       // [B] scratch for their stores
  int ld;             // row distance of the arrays when the launch covers a window of a larger batch; 0 = batch
  int nocost;         // leave the running-objective row of x unwritten (J only)
  const int* gate;    // optional: the launch does nothing if *gate == 0 (a sweep enqueued before the previous one's
                      // count of active instances is known, fb_sweep)
};

// ---------------------------------------------------------------------------------------
// forward.  nb = N / D blocks.  barrier_k (k = 0..nb) separates interval k-1 from interval k and is
// reached by M only once block k has landed in LDS.  In interval k
//   M issues the DMA of block k+Q into input slot (k+Q) % NSLOT, then waits for block k+1;
//   S processes block k     (inputs: slot k % NSLOT;        writes stage buffer k & 1);
//   C processes block k-1   (inputs: slot (k-1) % NSLOT;    reads stage buffer (k-1) & 1).
// The slot M overwrites in interval k last served block k+Q-NSLOT = k-2, read by C in interval k-1.
// ---------------------------------------------------------------------------------------
template <class P, bool OUT_X, bool FRZ>
__global__ __launch_bounds__(192) void k_forward_pl(const FwdArgsPL a) {
  constexpr int G = P::NS, NAUG = P::NAUG;
  static_assert(P::NC == 1 && P::NTC == 1, "pipeline kernels are written for one control and one time coefficient");
  using C_ = PLCfg<G, false>;
  constexpr int D = C_::D, TPW = C_::TPW, Q = C_::Q, NSLOT = C_::NSLOT, RS = C_::RS, SCO = C_::SCO;
  // [buffer][Y1..Y4][step][lane], the step stride padded so that the objective wave, whose lanes of a trajectory
  // take consecutive steps, reads G rows of a step per lane without bank conflicts
  // (pairs (Y1, Y2), (Y3, Y4): two 16-byte LDS instructions per step on the writing side; with 65 elements per step
  //  the G lanes of a trajectory in C, one step apart, sit 16 bytes apart in the banks)
  constexpr int SS = (G == 1) ? 64 : 65;
  __shared__ __attribute__((aligned(16))) double2 stage[2][2][D][SS];
  __shared__ double ulast[2][64];  // control sample at the first node of a block (S -> C)
  __shared__ __attribute__((aligned(16))) double inp[NSLOT][C_::SLOT];  // [slot]{records | u}
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t B = (size_t)(a.ld ? a.ld : a.batch);
  const int nb = a.N / D;
  const int bw = blockIdx.x * TPW;  // first trajectory of this workgroup
  if (a.gate && *a.gate == 0) return;

  if (wave == 0) {
    // ---------------- M: HBM -> LDS ----------------
    auto issue = [&](int j) OCS_INLINE {
      double* dst = &inp[j % NSLOT][0];
#pragma unroll
      for (int q = 0; q < C_::NREC; ++q)
        dma16(a.REC + (size_t)j * C_::REC_DBL + q * 128 + 2 * lane, dst + q * 128);
#pragma unroll
      for (int q = 0; q < C_::NU; ++q) {
        const int e = q * 128 + 2 * lane, row = e / TPW, tl = e % TPW;
        dma16(a.u + ((size_t)(2 * D * j + 1 + row)) * B + bw + tl, dst + C_::REC_DBL + q * 128);
      }
    };
    for (int j = 0; j < Q && j < nb; ++j) issue(j);
    long long tw = 0, tb = 0;
    const long long t00 = PL_T();
    for (int k = 0; k <= nb; ++k) {
      const long long t0 = PL_T();
      if (k < nb) {
        const int behind = (nb - 1 - k) < (Q - 1) ? (nb - 1 - k) : (Q - 1);  // younger blocks in flight
        wait_blocks<C_::LPB>(behind);
      }
      const long long t1 = PL_T();
      lds_barrier();
      const long long t2 = PL_T();
      tw += t1 - t0;
      tb += t2 - t1;
      if (k + Q < nb) issue(k + Q);
    }
#ifdef OCS_PL_STAMPS
    if (a.dbg && lane == 0) {
      a.dbg[blockIdx.x * 16 + 0] = tw;
      a.dbg[blockIdx.x * 16 + 1] = tb;
      a.dbg[blockIdx.x * 16 + 2] = PL_T() - t00;
    }
#endif
    (void)tw; (void)tb; (void)t00;
  } else if (G == 1) {
    // One row per trajectory: nothing to reduce across lanes and no redundant store; C sums the objective lane by
    // lane and, having slack, also stores the state row (from Y1), so that S only marches.
    const int r = lane % G;
    const int tl = lane / G;
    const int b = bw + tl;
    const uniform_ptr PS = as_uniform(a.ps);
    const typename P::RowPar rp = P::load_row([&](int k) OCS_INLINE {
      return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + b] : PS[k];
    }, r);
    // a frozen trajectory writes every value to one scratch double (pointer stride 0): no branch around stores
    // (FRZ is a template parameter: without frozen lanes the column stride stays a scalar)
    // frozen trajectories skip their stores (a scratch address written a thousand times by the same lane was
    // measured to cost up to 60 % of the kernel once most instances are frozen)
    const bool fz = FRZ && a.frozen[b] != 0;
    const size_t colB = (size_t)NAUG * B;
    const double u0 = a.u[b];
    if (wave == 1) {
      // ---------------- S: state recursion ----------------
      // the recursion runs on z = y - m_r/2 (P::row_f_shifted: two dependent operations per stage, not three)
      const double mh = P::row_shift(rp);
      const double y0 = a.x0[(size_t)r * B + b];
      double y = y0 - mh;
      double cprev = P::row_vertex(mh, u0);
      long long tb = 0, tc = 0;
      for (int k = 0; k <= nb; ++k) {
        const long long t0 = PL_T();
        lds_barrier();
        const long long t1 = PL_T();
        tb += t1 - t0;
        if (k < nb) {
          const double* rec = &inp[k % NSLOT][0];
          const double* us = rec + C_::REC_DBL + tl;
          double2* w = &stage[k & 1][0][0][lane];
          // LDS reads of step s+1 are issued before step s is computed (LDS latency ~100 cycles would
          // otherwise sit on every step: the scheduler keeps loads next to their uses)
          struct In { double h, hh, h6, uM, uB; };
          auto fetch = [&](int s) OCS_INLINE {
            In v;
            v.h = rec[RS * s];
            v.hh = rec[RS * s + 1];
            v.h6 = rec[RS * s + 2];
            v.uM = us[(2 * s) * TPW];
            v.uB = us[(2 * s + 1) * TPW];
            return v;
          };
          In nxt = fetch(0);
#pragma unroll
          for (int s = 0; s < D; ++s) {
            const In c = nxt;
            if (s + 1 < D) nxt = fetch(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            const double cM = P::row_vertex(mh, c.uM), cB = P::row_vertex(mh, c.uB);
            const double F1 = P::row_f_shifted(y, cprev);
            double Y = __builtin_fma(c.hh, F1, y);
            w[s * SS] = double2{y, Y};                     // (Y1, Y2) - m/2
            const double F2 = P::row_f_shifted(Y, cM);
            Y = __builtin_fma(c.hh, F2, y);
            const double Y3 = Y;
            const double F3 = P::row_f_shifted(Y, cM);
            Y = __builtin_fma(c.h, F3, y);
            w[(D + s) * SS] = double2{Y3, Y};              // (Y3, Y4) - m/2
            const double F4 = P::row_f_shifted(Y, cB);
            y = __builtin_fma(c.h6, F4, __builtin_fma(c.h6, __builtin_fma(2.0, F3, __builtin_fma(2.0, F2, F1)), y));  // (F4 joins last:
            // one dependent operation after it instead of two)
            cprev = cB;
          }
        }
        tc += PL_T() - t1;
      }
#ifdef OCS_PL_STAMPS
      if (a.dbg && lane == 0) {
        a.dbg[blockIdx.x * 16 + 4] = tb;
        a.dbg[blockIdx.x * 16 + 5] = tc;
      }
#endif
      (void)tb; (void)tc;
      if (OUT_X && !fz) a.x[((size_t)a.N * NAUG + r) * B + b] = y + mh;  // x(t_N); C stores the other nodes
    } else {
      // ---------------- C: objective ----------------
      // pc += W_A q1 + W_M (q2 + q3) + W_B q4 with the quadrature weights of the record table
      // (W_A = h/6 e^{-r t_A}, ...): the same sum as h/6 (F1 + 2 F2 + 2 F3 + F4) of the cost row.
      const double mh = P::row_shift(rp);  // the stage values arrive as Y - m_r/2
      double pc = 0.0, uprev2 = u0 * u0;
      double* xc = a.x + (size_t)G * B + b;
      double* xr = a.x + (size_t)r * B + b;
      const bool wc = !fz && !a.nocost;  // the objective row is written
      if (OUT_X && wc) *xc = 0.0;
      long long tb = 0, tc = 0;
      for (int k = 0; k <= nb; ++k) {
        const long long t0 = PL_T();
        lds_barrier();
        const long long t1 = PL_T();
        tb += t1 - t0;
        if (k >= 1) {
          const int j = k - 1;
          const double* rec = &inp[j % NSLOT][0];
          const double* us = rec + C_::REC_DBL + tl;
          const double2* w = &stage[j & 1][0][0][lane];
          struct In { double wA, wM, wB, uM, uB, Y1, Y2, Y3, Y4; };
          auto fetch = [&](int s) OCS_INLINE {
            In v;
            v.wA = rec[RS * s + SCO + 3];
            v.wM = rec[RS * s + SCO + 4];
            v.wB = rec[RS * s + SCO + 5];
            v.uM = us[(2 * s) * TPW];
            v.uB = us[(2 * s + 1) * TPW];
            const double2 p12 = w[s * SS], p34 = w[(D + s) * SS];
            v.Y1 = p12.x;
            v.Y2 = p12.y;
            v.Y3 = p34.x;
            v.Y4 = p34.y;
            return v;
          };
          In nxt = fetch(0);
#pragma unroll
          for (int s = 0; s < D; ++s) {
            const In c = nxt;
            if (s + 1 < D) nxt = fetch(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            const double uM2 = c.uM * c.uM, uB2 = c.uB * c.uB;
            const double y1 = c.Y1 + mh;
            if (OUT_X) {
              if (!fz) *xr = y1;   // x(t_i)
              xr += colB;
            }
            const double q1 = P::row_q(y1, uprev2, rp), q2 = P::row_q(c.Y2 + mh, uM2, rp);
            const double q3 = P::row_q(c.Y3 + mh, uM2, rp), q4 = P::row_q(c.Y4 + mh, uB2, rp);
            pc = __builtin_fma(c.wA, q1, __builtin_fma(c.wM, q2 + q3, __builtin_fma(c.wB, q4, pc)));
            if (OUT_X) {
              xc += colB;
              const double pcs = group_sum_pl<G>(pc);
              if (wc) *xc = pcs;
            }
            uprev2 = uB2;
          }
        }
        tc += PL_T() - t1;
      }
#ifdef OCS_PL_STAMPS
      if (a.dbg && lane == 0) {
        a.dbg[blockIdx.x * 16 + 8] = tb;
        a.dbg[blockIdx.x * 16 + 9] = tc;
      }
#endif
      (void)tb; (void)tc;
      const double Jt = group_sum_pl<G>(pc);
      if (!fz) a.J[b] = Jt;
    }
  } else if (wave == 1) {
    // ---------------- S: state recursion ----------------
    // Only the recursion: the stage states go to LDS, C stores the trajectory from there.  The recursion runs on
    // z = y - m_r/2 (P::row_f_shifted: two dependent operations per stage instead of three).
    const int r = lane % G;
    const int tl = lane / G;
    const int b = bw + tl;
    const uniform_ptr PS = as_uniform(a.ps);
    const typename P::RowPar rp = P::load_row([&](int k) OCS_INLINE {
      return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + b] : PS[k];
    }, r);
    const bool fz = FRZ && a.frozen[b] != 0;
    const double mh = P::row_shift(rp);
    double uprev = a.u[b];
    double y = a.x0[(size_t)r * B + b] - mh;
    double cprev = P::row_vertex(mh, uprev);
    long long tb = 0, tc = 0;
    for (int k = 0; k <= nb; ++k) {
      const long long t0 = PL_T();
      lds_barrier();
      const long long t1 = PL_T();
      tb += t1 - t0;
      if (k < nb) {
        const double* rec = &inp[k % NSLOT][0];
        const double* us = rec + C_::REC_DBL + tl;
        double2* w = &stage[k & 1][0][0][lane];
        ulast[k & 1][lane] = uprev;
        // LDS reads of step s+1 are issued before step s is computed (LDS latency ~100 cycles would
        // otherwise sit on every step: the scheduler keeps loads next to their uses)
        struct In { double h, hh, h6, uM, uB; };
        auto fetch = [&](int s) OCS_INLINE {
          In v;
          v.h = rec[RS * s];
          v.hh = rec[RS * s + 1];
          v.h6 = rec[RS * s + 2];
          v.uM = us[(2 * s) * TPW];
          v.uB = us[(2 * s + 1) * TPW];
          return v;
        };
        In nxt = fetch(0);
#pragma unroll
        for (int s = 0; s < D; ++s) {
          const In c = nxt;
          if (s + 1 < D) nxt = fetch(s + 1);
          __builtin_amdgcn_sched_barrier(0);
          const double cM = P::row_vertex(mh, c.uM), cB = P::row_vertex(mh, c.uB);
          const double F1 = P::row_f_shifted(y, cprev);
          double Y = __builtin_fma(c.hh, F1, y);
          w[s * SS] = double2{y, Y};                     // (Y1, Y2) - m/2
          const double F2 = P::row_f_shifted(Y, cM);
          Y = __builtin_fma(c.hh, F2, y);
          const double Y3 = Y;
          const double F3 = P::row_f_shifted(Y, cM);
          Y = __builtin_fma(c.h, F3, y);
          w[(D + s) * SS] = double2{Y3, Y};              // (Y3, Y4) - m/2
          const double F4 = P::row_f_shifted(Y, cB);
          y = __builtin_fma(c.h6, F4, __builtin_fma(c.h6, __builtin_fma(2.0, F3, __builtin_fma(2.0, F2, F1)), y));  // (F4 joins last:
            // one dependent operation after it instead of two)
          cprev = cB;
          uprev = c.uB;
        }
      }
      tc += PL_T() - t1;
    }
#ifdef OCS_PL_STAMPS
    if (a.dbg && lane == 0) {
      a.dbg[blockIdx.x * 16 + 4] = tb;
      a.dbg[blockIdx.x * 16 + 5] = tc;
    }
#endif
    (void)tb; (void)tc;
    if (OUT_X && !fz) a.x[((size_t)a.N * NAUG + r) * B + b] = y + mh;  // x(t_N); the other nodes are stored by C
  } else {
    // ---------------- C: objective and the stores of the trajectory ----------------
    // The G lanes of a trajectory take G consecutive steps (not the G rows of one step): a lane reads all rows of its
    // step, so the sum over the rows needs no cross-lane reduction, the running objective is a prefix sum over the
    // lanes of a quad, and every value is stored once.
    //   d_i = W_A q1 + W_M (q2 + q3) + W_B q4,  q_j = sum_r row_q(Y_j,r), with the quadrature weights of the record
    //   table (W_A = h/6 e^{-r t_A}, ...): the same sum as h/6 (F1 + 2 F2 + 2 F3 + F4) of the cost row.
    const int csub = lane % G;
    const int ctl = lane / G;
    const int b = bw + ctl;
    const uniform_ptr PS = as_uniform(a.ps);
    typename P::RowPar rpr[G];
    double mhr[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
      rpr[q] = P::load_row([&](int k) OCS_INLINE {
        return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + b] : PS[k];
      }, q);
      mhr[q] = P::row_shift(rpr[q]);
    }
    const bool fz = FRZ && a.frozen[b] != 0;
    const size_t colB = (size_t)NAUG * B;
    double carry = 0.0;  // running objective at the node before this lane's step (OUT_X) / this lane's share (else)
    const bool wc = !fz && !a.nocost;  // the objective row is written
    if (OUT_X && wc && csub == 0) a.x[(size_t)G * B + b] = 0.0;
    long long tb = 0, tc = 0;
    for (int k = 0; k <= nb; ++k) {
      const long long t0 = PL_T();
      lds_barrier();
      const long long t1 = PL_T();
      tb += t1 - t0;
      if (k >= 1) {
        const int j = k - 1;
        const double* rec = &inp[j % NSLOT][0];
        const double* us = rec + C_::REC_DBL + ctl;
        const double2* st = &stage[j & 1][0][0][ctl * G];
        const double ublk = ulast[j & 1][ctl * G];
        // all LDS reads of the block first (the passes below then wait for their own only)
        struct In { double wA, wM, wB, uA, uM, uB, Y[4][G]; };
        In in[D / G];
#pragma unroll
        for (int p0 = 0; p0 < D; p0 += G) {
          const int s = p0 + csub;
          In& v = in[p0 / G];
          v.wA = rec[RS * s + SCO + 3];
          v.wM = rec[RS * s + SCO + 4];
          v.wB = rec[RS * s + SCO + 5];
          v.uM = us[(2 * s) * TPW];
          v.uB = us[(2 * s + 1) * TPW];
          v.uA = us[(s > 0 ? 2 * s - 1 : 0) * TPW];
#pragma unroll
          for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int q = 0; q < G; ++q) {
              const double2 pr = st[(pp * D + s) * SS + q];
              v.Y[2 * pp][q] = pr.x;
              v.Y[2 * pp + 1][q] = pr.y;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p0 = 0; p0 < D; p0 += G) {
          const int s = p0 + csub;
          const In& c = in[p0 / G];
          const double uA = s > 0 ? c.uA : ublk;
          const double uA2 = uA * uA, uM2 = c.uM * c.uM, uB2 = c.uB * c.uB;
          double Y1[G];
          const double cqM = P::control_q(uM2, rpr[0]);
          double q1 = P::control_q(uA2, rpr[0]), q2 = cqM, q3 = cqM, q4 = P::control_q(uB2, rpr[0]);
#pragma unroll
          for (int q = 0; q < G; ++q) {
            Y1[q] = c.Y[0][q] + mhr[q];
            q1 = P::state_q_acc(Y1[q], q1);
            q2 = P::state_q_acc(c.Y[1][q] + mhr[q], q2);
            q3 = P::state_q_acc(c.Y[2][q] + mhr[q], q3);
            q4 = P::state_q_acc(c.Y[3][q] + mhr[q], q4);
          }
          const double d = __builtin_fma(c.wA, q1, __builtin_fma(c.wM, q2 + q3, c.wB * q4));
          if (OUT_X) {
            // inclusive prefix over the G steps of the pass, on top of the running objective
            double pre = d;
            if (G >= 2) {
              const double t = dpp_quad_pl<(G == 4) ? 0x90 : 0xA0>(pre);  // lane <- lane - 1
              pre += (csub >= 1) ? t : 0.0;
            }
            if (G == 4) {
              const double t = dpp_quad_pl<0x44>(pre);                    // lane <- lane - 2
              pre += (csub >= 2) ? t : 0.0;
            }
            const double tot = carry + pre;
            if (!fz) {
              double* xn = a.x + (size_t)(j * D + s) * colB + b;  // node i = j D + s
#pragma unroll
              for (int q = 0; q < G; ++q) xn[(size_t)q * B] = Y1[q];
              if (wc) xn[colB + (size_t)G * B] = tot;              // objective at node i + 1
            }
            carry = (G == 1) ? tot : dpp_quad_pl<(G == 4) ? 0xFF : 0xF5>(tot);  // the pass's last lane
          } else {
            carry += d;
          }
        }
      }
      tc += PL_T() - t1;
    }
#ifdef OCS_PL_STAMPS
    if (a.dbg && lane == 0) {
      a.dbg[blockIdx.x * 16 + 8] = tb;
      a.dbg[blockIdx.x * 16 + 9] = tc;
    }
#endif
    (void)tb; (void)tc;
    const double Jt = OUT_X ? carry : group_sum_pl<G>(carry);
    if (!fz) a.J[b] = Jt;
  }
}

// ---------------------------------------------------------------------------------------
// backward.  Blocks are numbered from the END of the horizon: block j covers steps
// i = N-1-j*D-s, s = 0..D-1 (the order they are processed in).  barrier_k, k = 0..nb+1.  In interval k
//   M issues the DMA of block k+Q, then waits for block k+1;
//   R processes block k     (slot k % NSLOT: checkpoints, controls, records; writes RA[k & 1]);
//   A processes block k-1   (records of slot (k-1) % NSLOT; reads RA[(k-1) & 1]; writes AD[(k-1) & 1]);
//   D processes block k-2   (controls, records of slot (k-2) % NSLOT; reads AD[(k-2) & 1]).
// The slot M overwrites in interval k last served block k+Q-NSLOT = k-3, read by D in interval k-1.
// ---------------------------------------------------------------------------------------
struct BwdArgsPL {
  int N, batch;
  const double* REC;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* xck;
  const double* u;
  const double* lamT;
  double* lam;   // state rows by the adjoint wave A, the constant cost row by R
  double* dJdu;
  double* lam0;
  long long* dbg;  // diagnostic build only
  const double* pend0;  // optional [B]: the k1 half of the last column (2N) when the steps above N were done by
                        // another kernel (a pass split at a multiple of the block length); default 0
};

// LT: a terminal costate was given; without one the cost row of lam is exactly 1 and the products with it drop out
template <class P, bool OUT_LAM, bool OUT_DJDU, bool LT>
__global__ __launch_bounds__(256) void k_backward_pl(const BwdArgsPL a) {
  constexpr int G = P::NS, NAUG = P::NAUG;
  static_assert(P::NC == 1 && P::NTC == 1, "pipeline kernels are written for one control and one time coefficient");
  using C_ = PLCfg<G, true>;
  constexpr int D = C_::D, TPW = C_::TPW, Q = C_::Q, NSLOT = C_::NSLOT, RS = C_::RS, SCO = C_::SCO;
  constexpr int UOFF = C_::REC_DBL, XOFF = C_::REC_DBL + C_::U_DBL;
  // R -> A: the four stage states as two 16-byte pairs (Y1, Y2), (Y3, Y4): two LDS instructions on either side
  __shared__ __attribute__((aligned(16))) double2 ra[2][2][D][64];
  // A -> D: k1, k2 + k3, k4 and lam(t_i).  D takes G consecutive steps of a trajectory per lane, i.e. reads the G rows
  // of a step: the padded step stride keeps those 16-byte reads of the lanes of a quad in different banks
  constexpr int KS = (G == 1) ? 64 : 72;
  __shared__ __attribute__((aligned(16))) double ad[2][4][D][KS];
  __shared__ __attribute__((aligned(16))) double inp[NSLOT][C_::SLOT]; // {records | u | checkpoints}
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t B = (size_t)a.batch;
  const int N = a.N, nb = N / D;
  const int bw = blockIdx.x * TPW;

  if (wave == 0) {
    // ---------------- M: HBM -> LDS ----------------
    auto issue = [&](int j) OCS_INLINE {
      double* dst = &inp[j % NSLOT][0];
      const int iLo = N - (j + 1) * D;  // lowest step of the block; LDS holds ascending steps
#pragma unroll
      for (int q = 0; q < C_::NREC; ++q)
        dma16(a.REC + (size_t)iLo * RS + q * 128 + 2 * lane, dst + q * 128);
#pragma unroll
      for (int q = 0; q < C_::NU; ++q) {  // samples 2*iLo .. 2*iLo + 2D - 1
        const int e = q * 128 + 2 * lane, row = e / TPW, tl = e % TPW;
        dma16(a.u + ((size_t)(2 * iLo + row)) * B + bw + tl, dst + UOFF + q * 128);
      }
#pragma unroll
      for (int q = 0; q < C_::NX; ++q) {  // checkpoint rows x(r, i), LDS layout [step][r][tl]
        const int e = q * 128 + 2 * lane, st = e / 64, rr = (e % 64) / TPW, tl = e % TPW;
        dma16(a.xck + ((size_t)(iLo + st) * NAUG + rr) * B + bw + tl, dst + XOFF + q * 128);
      }
    };
    for (int j = 0; j < Q && j < nb; ++j) issue(j);
    for (int k = 0; k <= nb + 1; ++k) {
      if (k < nb) {
        const int behind = (nb - 1 - k) < (Q - 1) ? (nb - 1 - k) : (Q - 1);
        wait_blocks<C_::LPB>(behind);
      }
      lds_barrier();
      if (k + Q < nb) issue(k + Q);
    }
    return;
  }
  const int r = lane % G;
  const int tl = lane / G;
  const int b = bw + tl;
  const uniform_ptr PS = as_uniform(a.ps);
  const typename P::RowPar rp = P::load_row([&](int k) OCS_INLINE {
    return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + b] : PS[k];
  }, r);
  const double lamc = a.lamT ? a.lamT[(size_t)G * B + b] : 1.0;
  const size_t colB = (size_t)NAUG * B;

  if (wave == 1) {
    // ---------------- R: stage states; also the constant cost row of lam ----------------
    // R is the compute wave with slack (151 of 252 cycles per step): it writes lam(end, :) = lamT(end) for the
    // columns of its block, D / (64 / TPW) stores per lane and block (a separate fill kernel used to: 8 us + a launch)
    const int tl2 = lane % TPW, so = lane / TPW;  // this lane's trajectory and first step offset for those stores
    const double lamc2 = a.lamT ? a.lamT[(size_t)G * B + bw + tl2] : 1.0;
    double* lc = a.lam + (size_t)G * B + bw + tl2;
    if (OUT_LAM && so == 0) lc[(size_t)N * colB] = lamc2;
    long long tb = 0, tc = 0;
    for (int k = 0; k <= nb + 1; ++k) {
      const long long t0 = PL_T();
      lds_barrier();
      const long long t1 = PL_T();
      tb += t1 - t0;
      if (OUT_LAM && k < nb) {
        const int itop = N - 1 - k * D;
#pragma unroll
        for (int s2 = 0; s2 < D; s2 += 64 / TPW) __builtin_nontemporal_store(lamc2, &lc[(size_t)(itop - s2 - so) * colB]);
      }
      if (k < nb) {
        const double* slot = &inp[k % NSLOT][0];
        const double* us = slot + UOFF + tl;
        const double* xs = slot + XOFF + r * TPW + tl;
        double2* w = &ra[k & 1][0][0][lane];
        struct In { double h, hh, xi, uA, uM; };
        auto fetch = [&](int s) OCS_INLINE {  // s-th step processed = local ascending index D-1-s
          const int l = D - 1 - s;
          In v;
          v.h = slot[RS * l];
          v.hh = slot[RS * l + 1];
          v.xi = xs[l * 64];
          v.uA = us[(2 * l) * TPW];
          v.uM = us[(2 * l + 1) * TPW];
          return v;
        };
        In nxt = fetch(0);
#pragma unroll
        for (int s = 0; s < D; ++s) {
          const In c = nxt;
          if (s + 1 < D) nxt = fetch(s + 1);
          __builtin_amdgcn_sched_barrier(0);
          double f = P::row_f(c.xi, c.uA, rp);
          const double Y2 = __builtin_fma(c.hh, f, c.xi);
          f = P::row_f(Y2, c.uM, rp);
          const double Y3 = __builtin_fma(c.hh, f, c.xi);
          f = P::row_f(Y3, c.uM, rp);
          const double Y4 = __builtin_fma(c.h, f, c.xi);
          w[(0 * D + s) * 64] = double2{c.xi, Y2};
          w[(1 * D + s) * 64] = double2{Y3, Y4};
        }
      }
      tc += PL_T() - t1;
    }
#ifdef OCS_PL_STAMPS
    if (a.dbg && lane == 0) {
      a.dbg[blockIdx.x * 16 + 0] = tb;
      a.dbg[blockIdx.x * 16 + 1] = tc;
    }
#endif
    (void)tb; (void)tc;
  } else if (wave == 2) {
    // ---------------- A: adjoint recursion ----------------
    double lam = a.lamT ? a.lamT[(size_t)r * B + b] : 0.0;
    if (OUT_LAM) a.lam[(size_t)N * colB + (size_t)r * B + b] = lam;
    long long tb = 0, tc = 0;
    for (int k = 0; k <= nb + 1; ++k) {
      const long long t0 = PL_T();
      lds_barrier();
      const long long t1 = PL_T();
      tb += t1 - t0;
      if (k >= 1 && k <= nb) {
        const int j = k - 1;
        const double* slot = &inp[j % NSLOT][0];
        const double2* w = &ra[j & 1][0][0][lane];
        double* kw = &ad[j & 1][0][0][lane];
        struct In { double h, hh, h6, h3, e4, e3, e1, Y1, Y2, Y3, Y4; };
        auto fetch = [&](int s) OCS_INLINE {
          const int l = D - 1 - s;
          In v;
          v.h = slot[RS * l];
          v.hh = slot[RS * l + 1];
          v.h6 = slot[RS * l + 2];
          v.h3 = slot[RS * l + 3];
          v.e4 = slot[RS * l + SCO];
          v.e3 = slot[RS * l + SCO + 1];
          v.e1 = slot[RS * l + SCO + 2];
          const double2 p12 = w[(0 * D + s) * 64], p34 = w[(1 * D + s) * 64];
          v.Y1 = p12.x;
          v.Y2 = p12.y;
          v.Y3 = p34.x;
          v.Y4 = p34.y;
          return v;
        };
        In nxt = fetch(0);
#pragma unroll
        for (int s = 0; s < D; ++s) {
          const In c = nxt;
          if (s + 1 < D) nxt = fetch(s + 1);
          __builtin_amdgcn_sched_barrier(0);
          const double ev4 = LT ? c.e4 * lamc : c.e4, ev3 = LT ? c.e3 * lamc : c.e3, ev1 = LT ? c.e1 * lamc : c.e1;
          const double h6l = c.h6 * lam, h3l = c.h3 * lam;
          const double k4 = h6l;                                   // :73
          const double g3 = P::row_dfdx(c.Y4, k4, ev4, rp);        // :74-75
          double acc = lam + g3;                                   // :86-88, each term as soon as it exists: only
          const double k3 = __builtin_fma(c.h, g3, h3l);           // :77     the last add stays on the dependent chain
          const double g2 = P::row_dfdx(c.Y3, k3, ev3, rp);        // :78-79
          acc += g2;
          const double k2 = __builtin_fma(c.hh, g2, h3l);          // :81
          const double g1 = P::row_dfdx(c.Y2, k2, ev3, rp);        // :82-83
          acc += g1;
          const double k1 = __builtin_fma(c.hh, g1, h6l);          // :85
          const double g0 = P::row_dfdx(c.Y1, k1, ev1, rp);        // :87-88
          lam = acc + g0;
          if (OUT_DJDU) {
            kw[(0 * D + s) * KS] = k1;
            kw[(1 * D + s) * KS] = k2 + k3;
            kw[(2 * D + s) * KS] = k4;
          }
          if (OUT_LAM) kw[(3 * D + s) * KS] = lam;  // stored by D
        }
      }
      tc += PL_T() - t1;
    }
#ifdef OCS_PL_STAMPS
    if (a.dbg && lane == 0) {
      a.dbg[blockIdx.x * 16 + 4] = tb;
      a.dbg[blockIdx.x * 16 + 5] = tc;
    }
#endif
    (void)tb; (void)tc;
    if (a.lam0) {
      a.lam0[(size_t)r * B + b] = lam;
      a.lam0[(size_t)G * B + b] = lamc;
    }
  } else {
    // ---------------- D: dJdu columns and the stores of lam ----------------
    // Lane (trajectory ctl, step csub of a pass of G steps; processed order, i.e. descending time) adds the rows itself:
    //   column 2i+2 = B'k1 of step i+1 + B'k4 of step i,  column 2i+1 = B'(k2 + k3) at the midpoint   (:97-121)
    //   with sum_r row_dfdu(cw_r u, k_r, ev) = row_dfdu(cws u, sum_r k_r, ev)  (linear in both)
    // The two halves of a node column meet through a lane shift; every column and every lam value is stored once.
    const int csub = lane % G, ctl = lane / G;
    const int bx = bw + ctl;
    double cws = 0.0;
#pragma unroll
    for (int q = 0; q < G; ++q)
      cws += P::load_row([&](int k) OCS_INLINE {
        return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + bx] : PS[k];
      }, q).cw;
    const double lamcx = a.lamT ? a.lamT[(size_t)G * B + bx] : 1.0;
    // carried from pass to pass by the lane of the last (lowest) step: its B'k1 and its node control
    double pend = a.pend0 ? a.pend0[bx] : 0.0;
    double ucar = OUT_DJDU ? a.u[(size_t)(2 * N) * B + bx] : 0.0;
    long long tb = 0, tc = 0;
    for (int k = 0; k <= nb + 1; ++k) {
      const long long t0 = PL_T();
      lds_barrier();
      const long long t1 = PL_T();
      tb += t1 - t0;
      if (k >= 2) {
        const int j = k - 2;
        const double* slot = &inp[j % NSLOT][0];
        const double* us = slot + UOFF + ctl;
        const double* kw = &ad[j & 1][0][0][ctl * G];
        // every LDS read of the block first, then the arithmetic
        struct In { double e4, e3, e1, uA, uM, k1[G], k23[G], k4[G], lm[G]; };
        In in[D / G];
#pragma unroll
        for (int p0 = 0; p0 < D; p0 += G) {
          const int s = p0 + csub, l = D - 1 - s;
          In& v = in[p0 / G];
          if (OUT_DJDU) {
            v.e4 = slot[RS * l + SCO];
            v.e3 = slot[RS * l + SCO + 1];
            v.e1 = slot[RS * l + SCO + 2];
            v.uA = us[(2 * l) * TPW];
            v.uM = us[(2 * l + 1) * TPW];
          }
#pragma unroll
          for (int q = 0; q < G; ++q) {
            if (OUT_DJDU) {
              v.k1[q] = kw[(0 * D + s) * KS + q];
              v.k23[q] = kw[(1 * D + s) * KS + q];
              v.k4[q] = kw[(2 * D + s) * KS + q];
            }
            if (OUT_LAM) v.lm[q] = kw[(3 * D + s) * KS + q];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int i0 = N - 1 - j * D - csub;  // this lane's step in the first pass
#pragma unroll
        for (int p0 = 0; p0 < D; p0 += G) {
          const In& c = in[p0 / G];
          const int i = i0 - p0;
          if (OUT_LAM) {
            double* lp = a.lam + (size_t)i * colB + bx;
#pragma unroll
            for (int q = 0; q < G; ++q) __builtin_nontemporal_store(c.lm[q], &lp[(size_t)q * B]);
          }
          if (OUT_DJDU) {
            const double ev4 = LT ? c.e4 * lamcx : c.e4, ev3 = LT ? c.e3 * lamcx : c.e3, ev1 = LT ? c.e1 * lamcx : c.e1;
            double S1 = c.k1[0], S23 = c.k23[0], S4 = c.k4[0];
#pragma unroll
            for (int q = 1; q < G; ++q) {
              S1 += c.k1[q];
              S23 += c.k23[q];
              S4 += c.k4[q];
            }
            // the node above this step: its control and its B'k1 sit in the lane of the step processed just before
            const double un_s = (G == 1) ? ucar : dpp_quad_pl<(G == 4) ? 0x90 : 0xA0>(c.uA);
            const double unext = csub ? un_s : ucar;
            const double p1 = P::row_dfdu(cws * c.uA, S1, ev1);
            const double p4 = P::row_dfdu(cws * unext, S4, ev4);
            const double cuM = cws * c.uM;
            const double p23 = P::row_dfdu(cuM + cuM, S23, ev3);  // (cuM ev3 - k2) + (cuM ev3 - k3)
            const double p1_s = (G == 1) ? pend : dpp_quad_pl<(G == 4) ? 0x90 : 0xA0>(p1);
            const double pabove = csub ? p1_s : pend;
            double* dp = a.dJdu + (size_t)(2 * i + 2) * B + bx;
            __builtin_nontemporal_store(pabove + p4, dp);   // column 2i+2 (non-temporal: see ocs_scan_kernels.hip)
            __builtin_nontemporal_store(p23, dp - B);       // column 2i+1
            pend = (G == 1) ? p1 : dpp_quad_pl<(G == 4) ? 0xFF : 0xF5>(p1);
            ucar = (G == 1) ? c.uA : dpp_quad_pl<(G == 4) ? 0xFF : 0xF5>(c.uA);
          }
        }
      }
      tc += PL_T() - t1;
    }
#ifdef OCS_PL_STAMPS
    if (a.dbg && lane == 0) {
      a.dbg[blockIdx.x * 16 + 8] = tb;
      a.dbg[blockIdx.x * 16 + 9] = tc;
    }
#endif
    (void)tb; (void)tc;
    if (OUT_DJDU && csub == 0) a.dJdu[bx] = pend;  // left end point :101-102
  }
}


// ---------------------------------------------------------------------------------------
// costate pass of the forward-backward sweep (compute_x_lam.m:11-14 on the node grid): lam' = adjointRHS(t, x(t),
// lam, u(t)), lam(TF) = 0, classical RK4 from TF down to T0 with x at the nodes and the pchip midpoints.
// Two waves: M streams the step records, x(t_i) and x(tmid_i) of a block into LDS with LDS-DMA, Q blocks ahead;
// L runs the recursion out of LDS and stores lam.  For the row-separable registry problems the adjoint right-hand
// side does not read u, so no control samples are moved.  Blocks are numbered from TF: block j covers steps
// i = N-1-j*D-s, s = 0..D-1.  barrier_k, k = 0..nb: M waits for block k, then both waves meet; L processes block k
// in interval k while M issues block k+Q.  Slot (k+Q) % NSLOT was last read by L in interval k+Q-NSLOT = k-1.
// ---------------------------------------------------------------------------------------
bool pipeline_supported(Functor f, int nS, int nC);
template <int G>
struct CostateCfg {
  static constexpr int D = 8, TPW = 64 / G, Q = 3, NSLOT = Q + 1;
  static constexpr int RS = rec_stride(1), SCO = rec_sc_offset(1);
  static constexpr int REC_DBL = D * RS, NREC = REC_DBL / 128;
  static constexpr int X_DBL = D * 64, NX = X_DBL / 128;          // x(t_i) rows of a block, [step][row][traj]
  static constexpr int SLOT = REC_DBL + 2 * X_DBL;               // + the midpoint rows
  static constexpr int LPB = NREC + 2 * NX;
};
struct CostateArgsPL {
  int N, batch, ld;       // ld: row distance of the arrays (window of a larger batch) or 0
  const double* REC;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* x;        // [N+1][ldx][B]
  int ldx;
  const double* xmid;     // [N][G][B]
  const int* frozen;
  double* dump;
  double* lam;            // [N+1][G][B]
};

template <class P, bool FRZ>
__global__ __launch_bounds__(128) void k_costate_pl(const CostateArgsPL a) {
  constexpr int G = P::NS;
  static_assert(P::NC == 1 && P::NTC == 1, "pipeline kernels are written for one control and one time coefficient");
  using C_ = CostateCfg<G>;
  constexpr int D = C_::D, TPW = C_::TPW, Q = C_::Q, NSLOT = C_::NSLOT, RS = C_::RS;
  constexpr int XOFF = C_::REC_DBL, MOFF = C_::REC_DBL + C_::X_DBL;
  __shared__ __attribute__((aligned(16))) double inp[NSLOT][C_::SLOT];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t B = (size_t)(a.ld ? a.ld : a.batch);
  const int N = a.N, nb = N / D;
  const int bw = tile_base(blockIdx.x, TPW, a.batch);
  if (wave == 0) {
    // ---------------- M: HBM -> LDS ----------------
    auto issue = [&](int j) OCS_INLINE {
      double* dst = &inp[j % NSLOT][0];
      const int iLo = N - (j + 1) * D;  // lowest step of the block; LDS holds ascending steps
#pragma unroll
      for (int q = 0; q < C_::NREC; ++q) dma16(a.REC + (size_t)iLo * RS + q * 128 + 2 * lane, dst + q * 128);
#pragma unroll
      for (int q = 0; q < C_::NX; ++q) {
        const int e = q * 128 + 2 * lane, st = e / 64, rr = (e % 64) / TPW, tl = e % TPW;
        dma16(a.x + ((size_t)(iLo + st) * a.ldx + rr) * B + bw + tl, dst + XOFF + q * 128);
        dma16(a.xmid + ((size_t)(iLo + st) * G + rr) * B + bw + tl, dst + MOFF + q * 128);
      }
    };
    for (int j = 0; j < Q && j < nb; ++j) issue(j);
    for (int k = 0; k <= nb; ++k) {
      if (k < nb) {
        const int behind = (nb - 1 - k) < (Q - 1) ? (nb - 1 - k) : (Q - 1);
        wait_blocks<C_::LPB>(behind);
      }
      lds_barrier();
      if (k + Q < nb) issue(k + Q);
    }
    return;
  }
  // ---------------- L: costate recursion ----------------
  chain_wave_priority();
  const int r = lane % G, tl = lane / G;
  const int b = bw + tl;
  const uniform_ptr PS = as_uniform(a.ps);
  const typename P::RowPar rp = P::load_row([&](int k) OCS_INLINE {
    return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + b] : PS[k];
  }, r);
  const bool fz = FRZ && a.frozen[b] != 0;  // frozen instances skip their stores
  const size_t colB = (size_t)G * B;
  double* ls = a.lam + (size_t)N * G * B + (size_t)r * B + b;
  double l = 0.0;                                           // lam0 = 0*x0   compute_x_lam.m:4
  double xB = a.x[((size_t)N * a.ldx + r) * B + b];         // x(t_N)
  double aB, bB;                                             // at x(t_N), with 2 e^{-r t_N}
  P::row_dfdx_pre(xB, 2.0 * a.REC[(size_t)(N - 1) * RS + 6], rp, aB, bB);
  if (!fz) *ls = 0.0;
  for (int k = 0; k <= nb; ++k) {
    lds_barrier();
    if (k < nb) {
      const double* slot = &inp[k % NSLOT][0];
      const double* xs = slot + XOFF + r * TPW + tl;
      const double* ms = slot + MOFF + r * TPW + tl;
      struct In { double h, hh, h6, eA, eM, xA, xM; };
      auto fetch = [&](int s) OCS_INLINE {  // s-th step processed = local ascending index D-1-s
        const int q = D - 1 - s;
        In v;
        v.h = slot[RS * q];
        v.hh = slot[RS * q + 1];
        v.h6 = slot[RS * q + 2];
        v.eA = slot[RS * q + rec_sc_offset(1) + 6];  // 2 e^{-r t} at the left node and at the midpoint (step constants); the
        v.eM = slot[RS * q + rec_sc_offset(1) + 7];  // right node's is the left node's of the step processed before
        v.xA = xs[q * 64];
        v.xM = ms[q * 64];
        return v;
      };
      In nxt = fetch(0);
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const In c = nxt;
        if (s + 1 < D) nxt = fetch(s + 1);
        __builtin_amdgcn_sched_barrier(0);
        // adjointRHS(t, x, lam) = -dFdx_times_vec(t, [x;0], u, [lam;1])(row r); ev = 2 e^{-rt} * 1
        // row of (dF/dy)' v in affine form a v + b (P::row_dfdx_pre); the pair of the right node is the left node's of
        // the step processed before
        double aM, bM, aA, bA;
        P::row_dfdx_pre(c.xM, c.eM, rp, aM, bM);
        P::row_dfdx_pre(c.xA, c.eA, rp, aA, bA);
        const double k1 = -__builtin_fma(aB, l, bB);
        double L = __builtin_fma(-c.hh, k1, l);
        const double k2 = -__builtin_fma(aM, L, bM);
        L = __builtin_fma(-c.hh, k2, l);
        const double k3 = -__builtin_fma(aM, L, bM);
        L = __builtin_fma(-c.h, k3, l);
        const double k4 = -__builtin_fma(aA, L, bA);
        l = __builtin_fma(-c.h6, k4, __builtin_fma(-c.h6, __builtin_fma(2.0, k3, __builtin_fma(2.0, k2, k1)), l));  // (k4 last)
        ls -= colB;
        if (!fz) *ls = l;
        aB = aA;
        bB = bA;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// The same costate pass forming the pchip midpoints x(tmid_i) itself: four more waves (H0..H3) evaluate them for
// the block the recursion wave takes next, out of the x(t_i) rows M has already put into LDS (each takes two
// consecutive intervals, so a node slope shared by two of them is computed once).  The separate midpoint kernel and
// the round trip of its output through memory disappear (waves 1, 2, 4, 5; the recursion wave is wave 3, alone on its SIMD).  Per-interval pchip constants come from a record table
// (PR: spacings, their reciprocals, slope weights, local midpoint) that M streams with the step records.
//   interval k (between barriers k and k+1):  M: issue block k+Q (block k+1 has landed);  H: block k-1;  L: block k-2
//   slot of block j is read in intervals j (one node, by H for block j-1), j+1 (H), j+2 (two nodes by H for block
//   j+1; L) and rewritten by M in interval j+NSLOT-Q  ->  NSLOT = Q + 3.
// ---------------------------------------------------------------------------------------
// (kPRec, the interval record layout, and pchip_end_pl are in ocs_device_common.hpp)
// MET: the pass also measures the change of the control its costate implies (see k_costate_plx): the costate rows of
// the sweep before and the ControlChar time coefficients of the block ride along in the slot, which lives one interval longer
template <int G, bool MET = false>
struct CostateXCfg {
  static constexpr int D = 8, TPW = 64 / G, Q = 3, NSLOT = Q + 3 + (MET ? 1 : 0);
  static constexpr int RS = rec_stride(1);
  static constexpr int REC_DBL = D * RS, NREC = REC_DBL / 128;
  static constexpr int PR_DBL = D * kPRec, NPR = PR_DBL / 128;
  static constexpr int X_DBL = D * 64, NX = X_DBL / 128;
  static constexpr int LOFF = REC_DBL + PR_DBL + X_DBL, TUOFF = LOFF + X_DBL;   // MET only
  static constexpr int SLOT = REC_DBL + PR_DBL + X_DBL + (MET ? X_DBL + 128 : 0);
  static constexpr int LPB = NREC + NPR + NX + (MET ? NX + 1 : 0);
  static_assert(PR_DBL % 128 == 0, "interval records of a block must be whole DMA instructions");
};
struct CostateXArgs {
  CostateArgsPL c;     // c.xmid is not used
  const double* PR;    // [N][kPRec]
  const int* gate;     // optional: the launch does nothing if *gate == 0
  // MET (fb_sweep with the control update folded into the state pass, ocs_fold_kernel.hpp): c.lam holds the costate of the
  // sweep before on entry.  The weighted change of the control at the nodes (fb_sweep.m:107), ControlChar(lam_new)
  // against ControlChar(lam_old), is taken while lam is replaced, and check_convergence + the loop bookkeeping
  // (:79-87, :99-115; k_fbs_advance) are done by the same wave at the end: c.frozen is the status array.
  const double* TU;    // [2N+1 (+128 readable)] ControlChar-side time coefficients (NTU = 1)
  const double* lb;
  const double* ub;
  double relTol, absTol;
  int sweep;           // sweep 1: c.lam holds nothing yet; the control before it is the lower bound (u0, fb_sweep.m:23)
  int* status;         // == c.frozen
  double* maxChange;   // [nSWEEPS][B]
  int* nactive;        // counter of the instances that continue
};


// one helper wave of k_costate_plx: the pchip midpoints of intervals q0 .. q0+RL-1 (ascending positions) of every block;
// ST: this wave also stores the costate values the recursion wave leaves in LDS (lr), so that the recursion wave
// carries neither the store pointer nor the exec masking of frozen instances
template <class C_, int RL, bool ST>
__device__ static inline void costate_midpoints(const double (*inp)[C_::SLOT], double (*xm)[C_::D][64], int q0, int N,
                                                int r, int tl, int lane, double xN, const double (*lr)[C_::D][64],
                                                double* lamp, size_t colB, bool fz) {
  constexpr int D = C_::D, TPW = C_::TPW, NSLOT = C_::NSLOT;
  constexpr int POFF = C_::REC_DBL, XOFF = C_::REC_DBL + C_::PR_DBL;
  const int nb = N / D;
  for (int k = 0; k <= nb + 2; ++k) {
    lds_barrier();
    if (ST && k >= 3 && !fz) {  // lam of the block the recursion wave finished in the interval before: its stores
      const int jl = k - 3;
      double* lp = lamp + (size_t)(N - 1 - jl * D) * colB;
#pragma unroll
      for (int s = 0; s < D; ++s) lp[-(ptrdiff_t)((size_t)s * colB)] = lr[jl & 1][s][lane];
    }
    const int j = k - 1;
    if (j < 0 || j >= nb) continue;
    const double* slot = &inp[0][0] + C_::SLOT * (j % NSLOT);
    const int iLo = N - (j + 1) * D;
    // Straight-line on purpose (no branch between the LDS reads, all slopes as interior ones behind an opaque
    // barrier): the three divisions then overlap; written with branches this wave took longer than the recursion.
    // records of the run
    double pr[RL][12];
#pragma unroll
    for (int c = 0; c < RL; ++c)
#pragma unroll
      for (int e = 0; e < 12; ++e) pr[c][e] = slot[POFF + (q0 + c) * kPRec + e];
    // nodes iLo+q0-1 .. iLo+q0+RL+1 of this lane's row (ascending): own block, one node of the block below
    // (j+1), two of the block above (j-1) or x(t_N); a node outside the grid reads some valid address instead
    double w[RL + 3];
#pragma unroll
    for (int t = 0; t < RL + 3; ++t) {
      const int q = q0 - 1 + t;  // position relative to this block
      const int jj = q < 0 ? (j + 1 < nb ? j + 1 : j) : (q < D ? j : (j > 0 ? j - 1 : j));
      const int qq = q < 0 ? D + q : (q < D ? q : q - D);
      const double v = (&inp[0][0] + C_::SLOT * (jj % NSLOT) + XOFF + r * TPW + tl)[qq * 64];
      w[t] = (q >= D && j == 0) ? xN : v;
    }
    double sec[RL + 2];  // secants of intervals iLo+q0-1 .. iLo+q0+RL
#pragma unroll
    for (int t = 0; t < RL + 2; ++t) {
      const double ih = t == 0 ? pr[0][3] : (t == RL + 1 ? pr[RL - 1][5] : pr[t - 1][4]);
      sec[t] = (w[t + 1] - w[t]) * ih;
    }
    double d[RL + 1];  // slopes at nodes iLo+q0 .. iLo+q0+RL
#pragma unroll
    for (int c = 0; c < RL + 1; ++c) {
      d[c] = pchip_interior_f(sec[c], sec[c + 1], c < RL ? pr[c][6] : pr[RL - 1][8], c < RL ? pr[c][7] : pr[RL - 1][9]);
    }
    if (iLo + q0 == 0) d[0] = pchip_end_pl(pr[0][1], pr[0][2], sec[1], sec[2]);
    if (iLo + q0 + RL == N) d[RL] = pchip_end_pl(pr[RL - 1][1], pr[RL - 1][0], sec[RL], sec[RL - 1]);
    double* out = &xm[j & 1][0][lane];
#pragma unroll
    for (int c = 0; c < RL; ++c)   // the cubic at the middle of its interval (ocs_device_common.hpp, kPRec)
      out[(q0 + c) * 64] = __builtin_fma(pr[c][11], d[c] - d[c + 1], 0.5 * (w[c + 1] + w[c + 2]));
  }
}

template <class P, bool FRZ, bool MET = false>
__global__ __launch_bounds__(MET ? 576 : 320) void k_costate_plx(const CostateXArgs aa) {
  constexpr int G = P::NS;
  static_assert(P::NC == 1 && P::NTC == 1, "pipeline kernels are written for one control and one time coefficient");
  using C_ = CostateXCfg<G, MET>;
  constexpr int D = C_::D, TPW = C_::TPW, Q = C_::Q, NSLOT = C_::NSLOT, RS = C_::RS;
  constexpr int POFF = C_::REC_DBL, XOFF = C_::REC_DBL + C_::PR_DBL;
  const CostateArgsPL& a = aa.c;
  __shared__ __attribute__((aligned(16))) double inp[NSLOT][C_::SLOT];
  __shared__ __attribute__((aligned(16))) double xm[2][D][64];
  __shared__ __attribute__((aligned(16))) double lr[2][D][64];  // L -> H2: lam of a block, for the stores
  __shared__ double xres[MET ? 3 : 1][3][64];                   // MET: the partial maxima of the X waves
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t B = (size_t)(a.ld ? a.ld : a.batch);
  const int N = a.N, nb = N / D;
  const int bw = tile_base(blockIdx.x, TPW, a.batch);
  if (aa.gate && *aa.gate == 0) return;
  if (wave == 0) {
    // ---------------- M: HBM -> LDS ----------------
    auto issue = [&](int j) OCS_INLINE {
      double* dst = &inp[j % NSLOT][0];
      const int iLo = N - (j + 1) * D;
#pragma unroll
      for (int q = 0; q < C_::NREC; ++q) dma16(a.REC + (size_t)iLo * RS + q * 128 + 2 * lane, dst + q * 128);
#pragma unroll
      for (int q = 0; q < C_::NPR; ++q) dma16(aa.PR + (size_t)iLo * kPRec + q * 128 + 2 * lane, dst + POFF + q * 128);
#pragma unroll
      for (int q = 0; q < C_::NX; ++q) {
        const int e = q * 128 + 2 * lane, st = e / 64, rr = (e % 64) / TPW, tl = e % TPW;
        dma16(a.x + ((size_t)(iLo + st) * a.ldx + rr) * B + bw + tl, dst + XOFF + q * 128);
        // the costate of the sweep before, read Q blocks ahead of the stores that replace it (three intervals behind)
        if (MET) dma16(a.lam + ((size_t)(iLo + st) * G + rr) * B + bw + tl, dst + C_::LOFF + q * 128);
      }
      if (MET) dma16(aa.TU + (size_t)2 * iLo + 2 * lane, dst + C_::TUOFF);
    };
    for (int j = 0; j < Q && j < nb; ++j) issue(j);
    for (int k = 0; k <= nb + 2; ++k) {
      if (k < nb) {
        const int behind = (nb - 1 - k) < (Q - 1) ? (nb - 1 - k) : (Q - 1);
        wait_blocks<C_::LPB>(behind);
      }
      lds_barrier();
      if (k + Q < nb) issue(k + Q);
    }
    if (MET) lds_barrier();   // the X waves exchange their partial maxima
    return;
  }
  const int r = lane % G, tl = lane / G;
  const int b = bw + tl;
  if (MET && wave == 7) {   // keeps the recursion wave alone on its SIMD (waves w, w + 4, ..): takes part in the barriers only
    for (int k = 0; k <= nb + 3; ++k) lds_barrier();
    return;
  }
  if (MET && wave >= 5) {
    // ---------------- X0..X2: the change of the control at the nodes, then check_convergence for this trajectory -------
    // (waves 5, 6, 8: steps 0-2, 3-5, 6-7 of every block; one per SIMD beside the recursion wave's)
    // lane (trajectory tl, every G-th step of the wave's share): all rows of a node, new (lr, left by L) and old (the slot)
    const int xw = wave == 5 ? 0 : (wave == 6 ? 1 : 2);
    const int sLo = xw * 3, sCnt = xw == 2 ? 2 : 3;
    const uniform_ptr PS = as_uniform(a.ps);
    const typename P::CCPre ccp = P::cc_pre(P::load(ParamSrc{PS, a.pb, a.pmask, B, b}));
    const double lb = aa.lb[0], ub = aa.ub[0];
    double nmax = 0.0, dmax = 1.0;   // the largest weighted change as a fraction (k_control_grid's bookkeeping)
    bool any = false;
    auto take = [&](double un, double uo) OCS_INLINE {
      const double n = fabs(un - uo), d = aa.relTol * fabs(uo) + aa.absTol;
      const bool valid = (n + d) > 0.0;   // n / d is not NaN: neither is one, and they are not both zero (max() skips NaN, :108)
      const bool rep = valid & (!any | (n * dmax > nmax * d));
      nmax = rep ? n : nmax;
      dmax = rep ? d : dmax;
      any = any | valid;
    };
    const bool u0lb = aa.sweep == 1;   // (whatever lam held before the first sweep is dropped by the selects)
    if (r == 0 && xw == 0) {   // node t_N: lam = 0 before and after the pass (compute_x_lam.m:4)
      double z[G];
#pragma unroll
      for (int q = 0; q < G; ++q) z[q] = 0.0;
      const double uN = P::control_char_pre(aa.TU[(size_t)2 * N], z, ccp, lb, ub);
      take(uN, u0lb ? lb : uN);
    }
    for (int k = 0; k <= nb + 2; ++k) {
      lds_barrier();
      const int j = k - 3;
      if (j < 0) continue;
      const double* slot = &inp[j % NSLOT][0];
#pragma unroll
      for (int c0 = 0; c0 < 3; c0 += G) {
        if (c0 + r >= sCnt) continue;
        const int s = sLo + c0 + r, q = D - 1 - s;
        double ln[G], lo[G];
#pragma unroll
        for (int qq = 0; qq < G; ++qq) {
          ln[qq] = lr[j & 1][s][tl * G + qq];
          lo[qq] = slot[C_::LOFF + q * 64 + qq * TPW + tl];
        }
        const double tu = slot[C_::TUOFF + 2 * q];
        const double uo = P::control_char_pre(tu, lo, ccp, lb, ub);
        take(P::control_char_pre(tu, ln, ccp, lb, ub), u0lb ? lb : uo);
      }
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) {   // over the lanes of the trajectory
      const double on = __shfl_xor(nmax, m), od = __shfl_xor(dmax, m);
      const bool oa = __shfl_xor((int)any, m) != 0;
      if (oa && (!any || on * dmax > nmax * od)) {
        nmax = on;
        dmax = od;
      }
      any = any || oa;
    }
    if (xw > 0 && r == 0) {
      xres[xw][0][tl] = nmax;
      xres[xw][1][tl] = dmax;
      xres[xw][2][tl] = any ? 1.0 : 0.0;
    }
    lds_barrier();
    if (xw > 0) return;
#pragma unroll
    for (int w = 1; w < 3; ++w) {
      const double on = xres[w][0][tl], od = xres[w][1][tl];
      const bool oa = xres[w][2][tl] != 0.0;
      if (oa && (!any || on * dmax > nmax * od)) {
        nmax = on;
        dmax = od;
      }
      any = any || oa;
    }
    // fb_sweep.m:108-110 and :79-87 (k_fbs_advance): every other wave of this workgroup read the status when it started
    bool still = false;
    if (r == 0 && aa.status[b] == 0) {
      const double mx = any ? nmax / dmax : __builtin_nan("");
      aa.maxChange[(size_t)(aa.sweep - 1) * B + b] = mx;
      if (mx <= 1.0)
        aa.status[b] = aa.sweep;
      else
        still = true;
    }
    const unsigned long long mk = __ballot(still);
    if (lane == 0 && mk) atomicAdd(aa.nactive, __popcll(mk));
    return;
  }
  const double xN = a.x[((size_t)N * a.ldx + r) * B + b];  // x(t_N): the node above the first block
  if (wave != 3) {
    // ---------------- H0, H1, H2: pchip midpoints of intervals 0-2, 3-5, 6-7 of every block ----------------
    // (waves 1, 2, 4: one per SIMD beside the recursion wave's; two helpers on one SIMD were the bottleneck)
    if (wave == 1)
      costate_midpoints<C_, 3, false>(inp, xm, 0, N, r, tl, lane, xN, lr, nullptr, 0, true);
    else if (wave == 2)
      costate_midpoints<C_, 3, false>(inp, xm, 3, N, r, tl, lane, xN, lr, nullptr, 0, true);
    else  // the helper with the shortest run also stores lam
      costate_midpoints<C_, 2, true>(inp, xm, 6, N, r, tl, lane, xN, lr, a.lam + (size_t)r * B + b, (size_t)G * B,
                                     FRZ && a.frozen[b] != 0);
    if (MET) lds_barrier();
    return;
  }
  // ---------------- L: costate recursion ----------------
  chain_wave_priority();
  const uniform_ptr PS = as_uniform(a.ps);
  const typename P::RowPar rp = P::load_row([&](int k) OCS_INLINE {
    return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + b] : PS[k];
  }, r);
  const bool fz = FRZ && a.frozen[b] != 0;
  const size_t colB = (size_t)G * B;
  double l = 0.0;
  double xB = xN;
  double aB, bB;  // at x(t_N), with 2 e^{-r t_N}
  P::row_dfdx_pre(xB, 2.0 * a.REC[(size_t)(N - 1) * RS + 6], rp, aB, bB);
  if (!fz) a.lam[(size_t)N * colB + (size_t)r * B + b] = 0.0;
  for (int k = 0; k <= nb + 2; ++k) {
    lds_barrier();
    const int j = k - 2;
    if (j < 0 || j >= nb) continue;
    double* lw = &lr[j & 1][0][lane];
    const double* slot = &inp[j % NSLOT][0];
    const double* xs = slot + XOFF + r * TPW + tl;
    const double* ms = &xm[j & 1][0][lane];
    struct In { double h, hh, h6, eA, eM, xA, xM; };
    auto fetch = [&](int s) OCS_INLINE {
      const int q = D - 1 - s;
      In v;
      v.h = slot[RS * q];
      v.hh = slot[RS * q + 1];
      v.h6 = slot[RS * q + 2];
      v.eA = slot[RS * q + rec_sc_offset(1) + 6];  // 2 e^{-r t} at the left node and at the midpoint (step constants); the
      v.eM = slot[RS * q + rec_sc_offset(1) + 7];  // right node's is the left node's of the step processed before
      v.xA = xs[q * 64];
      v.xM = ms[q * 64];
      return v;
    };
    In nxt = fetch(0);
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const In c = nxt;
      if (s + 1 < D) nxt = fetch(s + 1);
      __builtin_amdgcn_sched_barrier(0);
      // row of (dF/dy)' v in affine form a v + b (P::row_dfdx_pre); the pair of the right node is the left node's of
      // the step processed before
      double aM, bM, aA, bA;
      P::row_dfdx_pre(c.xM, c.eM, rp, aM, bM);
      P::row_dfdx_pre(c.xA, c.eA, rp, aA, bA);
      const double k1 = -__builtin_fma(aB, l, bB);
      double L = __builtin_fma(-c.hh, k1, l);
      const double k2 = -__builtin_fma(aM, L, bM);
      L = __builtin_fma(-c.hh, k2, l);
      const double k3 = -__builtin_fma(aM, L, bM);
      L = __builtin_fma(-c.h, k3, l);
      const double k4 = -__builtin_fma(aA, L, bA);
      l = __builtin_fma(-c.h6, k4, __builtin_fma(-c.h6, __builtin_fma(2.0, k3, __builtin_fma(2.0, k2, k1)), l));  // (k4 last)
      lw[s * 64] = l;  // stored by the helper wave
      aB = aA;
      bB = bA;
    }
  }
  if (MET) lds_barrier();
}

template <class P>
static void run_costate_plx(const CostateXArgs& a, hipStream_t s) {
  const dim3 grid(tile_count(a.c.batch, 64 / P::NS)), block(320);
  if (a.c.frozen)
    k_costate_plx<P, true><<<grid, block, 0, s>>>(a);
  else
    k_costate_plx<P, false><<<grid, block, 0, s>>>(a);
}
int costate_prec() { return kPRec; }
bool costate_pl_ok(Functor f, int nS, int nC, int N, int batch);
// PR: [N][costate_prec()] interval records; the midpoints are formed inside (no xmid array)
int launch_costate_plx(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                       const int* frozen, double* dump, double* lam, int ld, hipStream_t s, const int* gate) {
  if (!costate_pl_ok(p.functor, p.nS, p.nC, g.N, batch) || (frozen && !dump) || !PR) return -1;
  if (ld == 0 && costate_scan_ok(p, g, batch)) return launch_costate_scan(p, g, batch, x, ldx, PR, frozen, lam, s, gate);
  CostateXArgs a{};   // (the MET fields stay zero: the plain costate pass)
  a.c = CostateArgsPL{g.N, batch, ld, g.REC, p.ps, p.pb, p.pmask, x, ldx, nullptr, frozen, dump, lam};
  a.PR = PR;
  a.gate = gate;
  if (p.nS == 1)
    run_costate_plx<LogisticK<1>>(a, s);
  else if (p.nS == 2)
    run_costate_plx<LogisticK<2>>(a, s);
  else if (p.nS == 4)
    run_costate_plx<LogisticK<4>>(a, s);
  else
    return -1;
  return hip_rc5(hipGetLastError());
}

// costate pass + change of the control + check_convergence (CostateXArgs, MET); lam holds the costate of the sweep before
int launch_costate_met(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                       const double* lb, const double* ub, double relTol, double absTol, int sweep, int* status,
                       double* maxChange, int* nactive, double* lam, hipStream_t s, const int* gate) {
  if (p.functor == Functor::User)   // hipRTC instance of the scan kernel (checks its own conditions)
    return launch_costate_scan_met(p, g, batch, x, ldx, PR, lb, ub, relTol, absTol, sweep, status, maxChange, nactive, lam, s, gate);
  if (!costate_pl_ok(p.functor, p.nS, p.nC, g.N, batch) || !PR || !g.TU || !status || !maxChange || !nactive) return -1;
  if (costate_scan_ok(p, g, batch))
    return launch_costate_scan_met(p, g, batch, x, ldx, PR, lb, ub, relTol, absTol, sweep, status, maxChange, nactive, lam, s, gate);
  const CostateXArgs a{CostateArgsPL{g.N, batch, 0, g.REC, p.ps, p.pb, p.pmask, x, ldx, nullptr, status, nullptr, lam},
                       PR, gate, g.TU, lb, ub, relTol, absTol, sweep, status, maxChange, nactive};
  const dim3 grid(tile_count(batch, 64 / p.nS)), block(576);
  if (p.nS == 1)
    k_costate_plx<LogisticK<1>, true, true><<<grid, block, 0, s>>>(a);
  else if (p.nS == 2)
    k_costate_plx<LogisticK<2>, true, true><<<grid, block, 0, s>>>(a);
  else if (p.nS == 4)
    k_costate_plx<LogisticK<4>, true, true><<<grid, block, 0, s>>>(a);
  else
    return -1;
  return hip_rc5(hipGetLastError());
}

// (k_costate_pl / k_costate_plx move no control samples: they are only instantiated for registry problems whose
//  dFdx_times_vec does not read u -- LogisticK -- which pipeline_supported() guarantees; a functor whose adjoint
//  right-hand side reads u must take launch_costate's lane kernel)
bool costate_pl_ok(Functor f, int nS, int nC, int N, int batch) {
  return pipeline_supported(f, nS, nC) && N >= 8 && N % 8 == 0 && tile_ok(batch, 64 / nS);
}
template <class P>
static void run_costate_pl(const CostateArgsPL& a, hipStream_t s) {
  const dim3 grid(tile_count(a.batch, 64 / P::NS)), block(128);
  if (a.frozen)
    k_costate_pl<P, true><<<grid, block, 0, s>>>(a);
  else
    k_costate_pl<P, false><<<grid, block, 0, s>>>(a);
}
int launch_costate_pl(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* xmid,
                      const int* frozen, double* dump, double* lam, int ld, hipStream_t s) {
  if (!costate_pl_ok(p.functor, p.nS, p.nC, g.N, batch) || (frozen && !dump)) return -1;
  const CostateArgsPL a{g.N, batch, ld, g.REC, p.ps, p.pb, p.pmask, x, ldx, xmid, frozen, dump, lam};
  if (p.nS == 1)
    run_costate_pl<LogisticK<1>>(a, s);
  else if (p.nS == 2)
    run_costate_pl<LogisticK<2>>(a, s);
  else if (p.nS == 4)
    run_costate_pl<LogisticK<4>>(a, s);
  else
    return -1;
  return hip_rc5(hipGetLastError());
}

// ---------------------------------------------------------------------------------------
bool pipeline_supported(Functor f, int nS, int nC) {
  return f == Functor::Logistic && (nS == 1 || nS == 2 || nS == 4) && nC == 1;
}
bool pipeline_shape_ok(int nS, int N, int batch, bool backward) {
  const int D = 8, TPW = 64 / nS;
  if (N < D || N % D != 0) return false;
  // whole tiles; the state pass (k_forward_p2) also takes a ragged last tile as a workgroup that overlaps its neighbour, given
  // at least one tile and an even row distance (ocs_pipeline2_kernel.hpp)
  return batch % TPW == 0 || (!backward && tile_ok(batch, TPW));
}
int pipeline_block_steps() { return 8; }

template <class P>
static void run_forward_pl(const FwdArgsPL& a, hipStream_t s) {
  constexpr int TPW = 64 / P::NS;
  const dim3 grid(a.batch / TPW), block(192);
  if (a.frozen) {
    if (a.x)
      k_forward_pl<P, true, true><<<grid, block, 0, s>>>(a);
    else
      k_forward_pl<P, false, true><<<grid, block, 0, s>>>(a);
  } else if (a.x) {
    k_forward_pl<P, true, false><<<grid, block, 0, s>>>(a);
  } else {
    k_forward_pl<P, false, false><<<grid, block, 0, s>>>(a);
  }
}
int launch_forward_pl(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                      double* x, double* J, const int* frozen, double* dump, int ld, hipStream_t s,
                      bool no_cost_row, const int* gate) {
  if (!pipeline_shape_ok(p.nS, g.N, batch, false) || (frozen && !dump)) return -1;
  static const bool v1 = getenv("OCS_FWD_V1") != nullptr;   // the previous kernel, for A/B timing
  if (!v1) return launch_forward_p2(p, g, batch, x0, u, x, J, frozen, ld, s, no_cost_row, gate);
  if (batch % (64 / p.nS) != 0) return -1;   // (the previous kernel: whole tiles only)
  FwdArgsPL a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, x0, u, x, J, nullptr, frozen, dump, ld, no_cost_row ? 1 : 0, gate};
#ifdef OCS_PL_STAMPS
  static long long* dbg = nullptr;
  const int nwg = batch / (64 / p.nS);
  if (!dbg) (void)hipMalloc((void**)&dbg, sizeof(long long) * 16 * 65536);
  (void)hipMemsetAsync(dbg, 0, sizeof(long long) * 16 * nwg, s);
  a.dbg = dbg;
#endif
  if (p.nS == 1)
    run_forward_pl<LogisticK<1>>(a, s);
  else if (p.nS == 2)
    run_forward_pl<LogisticK<2>>(a, s);
  else if (p.nS == 4)
    run_forward_pl<LogisticK<4>>(a, s);
  else
    return -1;
#ifdef OCS_PL_STAMPS
  {
    (void)hipStreamSynchronize(s);
    static int calls = 0;
    if (++calls % 8 == 0) {
      std::vector<long long> h(16 * nwg);
      (void)hipMemcpy(h.data(), dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
      double acc[16] = {0};
      for (int w = 0; w < nwg; ++w)
        for (int q = 0; q < 16; ++q) acc[q] += (double)h[16 * w + q] / nwg;
      fprintf(stderr, "[pl fwd nS=%d] cycles per wg: M wait %.0f barrier %.0f total %.0f | S barrier %.0f compute %.0f | C barrier %.0f compute %.0f\n",
              p.nS, acc[0], acc[1], acc[2], acc[4], acc[5], acc[8], acc[9]);
    }
  }
#endif
  return hip_rc5(hipGetLastError());
}

template <class P>
static void run_backward_pl(const BwdArgsPL& a, hipStream_t s) {
  constexpr int TPW = 64 / P::NS;
  const dim3 grid(a.batch / TPW), block(256);
  if (a.lamT) {
    if (a.lam && a.dJdu)
      k_backward_pl<P, true, true, true><<<grid, block, 0, s>>>(a);
    else if (a.lam)
      k_backward_pl<P, true, false, true><<<grid, block, 0, s>>>(a);
    else
      k_backward_pl<P, false, true, true><<<grid, block, 0, s>>>(a);
  } else if (a.lam && a.dJdu) {
    k_backward_pl<P, true, true, false><<<grid, block, 0, s>>>(a);
  } else if (a.lam) {
    k_backward_pl<P, true, false, false><<<grid, block, 0, s>>>(a);
  } else {
    k_backward_pl<P, false, true, false><<<grid, block, 0, s>>>(a);
  }
}
int launch_backward_pl(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                       const double* lamT, double* lam, double* dJdu, double* lam0, const double* pend0,
                       hipStream_t s) {
  if (!pipeline_shape_ok(p.nS, g.N, batch, true) || (!lam && !dJdu)) return -1;
  BwdArgsPL a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, xck, u, lamT, lam, dJdu, lam0, nullptr, pend0};
#ifdef OCS_PL_STAMPS
  static long long* dbgb = nullptr;
  const int nwg = batch / (64 / p.nS);
  if (!dbgb) (void)hipMalloc((void**)&dbgb, sizeof(long long) * 16 * 65536);
  (void)hipMemsetAsync(dbgb, 0, sizeof(long long) * 16 * nwg, s);
  a.dbg = dbgb;
#endif
  if (p.nS == 1)
    run_backward_pl<LogisticK<1>>(a, s);
  else if (p.nS == 2)
    run_backward_pl<LogisticK<2>>(a, s);
  else if (p.nS == 4)
    run_backward_pl<LogisticK<4>>(a, s);
  else
    return -1;
#ifdef OCS_PL_STAMPS
  {
    (void)hipStreamSynchronize(s);
    static int calls = 0;
    if (++calls % 8 == 0) {
      std::vector<long long> h(16 * nwg);
      (void)hipMemcpy(h.data(), dbgb, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
      double acc[16] = {0};
      for (int w = 0; w < nwg; ++w)
        for (int q = 0; q < 16; ++q) acc[q] += (double)h[16 * w + q] / nwg;
      fprintf(stderr, "[pl bwd nS=%d] cycles per wg: R barrier %.0f compute %.0f | A barrier %.0f compute %.0f | D barrier %.0f compute %.0f\n",
              p.nS, acc[0], acc[1], acc[4], acc[5], acc[8], acc[9]);
    }
  }
#endif
  return hip_rc5(hipGetLastError());
}

}  // namespace ocs
