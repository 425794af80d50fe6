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
//   forward   wave M: memory wave.  Streams control samples and step records HBM -> LDS with LDS-DMA
//                     (global_load_lds_dwordx4, 1 KiB per instruction, no VGPR staging), Q blocks
//                     ahead of the compute waves; it is the only wave that waits on loads.
//             wave S: the state recursion (F1..F4, Y2..Y4, y_{i+1}) out of LDS; stores the state
//                     rows; publishes the four stage states of every step.
//             wave C: integrates the objective from the published stage states, reduces it over
//                     the rows of a trajectory (DPP) and stores the cost row / J.
//   backward  wave M: streams checkpoints, control samples and records.
//             wave R: recomputes the stage states from the checkpoints, publishes them.
//             wave A: the adjoint recursion (dJdk, lam); stores lam; publishes k1..k4.
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

template <int G>
struct PLCfg {
  static constexpr int D = (G == 4) ? 16 : 8;   // steps per hand-off block
  static constexpr int TPW = 64 / G;            // trajectories per workgroup
  static constexpr int Q = 3;                   // blocks the memory wave runs ahead (wait_blocks handles <= 2 younger)
  static constexpr int NSLOT = Q + 2;           // input ring slots (see the schedule below)
  static constexpr int REC_DBL = D * 8;         // records of a block (NTC = 1: 8 doubles each)
  static constexpr int U_DBL = 2 * D * TPW;     // control samples of a block
  static constexpr int NU = U_DBL / 128;        // DMA instructions for them
  static_assert(U_DBL % 128 == 0, "a block of control samples must be whole DMA instructions");
};

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
};

// Schedule.  nb = N / D blocks.  barrier_k (k = 0..nb) separates interval k-1 from interval k and
// is reached by M only once block k has landed in LDS.  In interval k
//   M issues the DMA of block k+Q into input slot (k+Q) % NSLOT, then waits for block k+1;
//   S processes block k     (inputs: slot k % NSLOT;        writes stage buffer k & 1);
//   C processes block k-1   (inputs: slot (k-1) % NSLOT;    reads stage buffer (k-1) & 1).
// The slot M overwrites in interval k last served block k+Q-NSLOT = k-2, read by C in interval k-1.
template <class P, bool OUT_X>
__global__ __launch_bounds__(192) void k_forward_pl(const FwdArgsPL a) {
  constexpr int G = P::NS, NAUG = P::NAUG;
  static_assert(P::NC == 1 && P::NTC == 1, "pipeline kernels are written for one control and one time coefficient");
  using C_ = PLCfg<G>;
  constexpr int D = C_::D, TPW = C_::TPW, Q = C_::Q, NSLOT = C_::NSLOT;
  constexpr int SLOT = C_::REC_DBL + C_::U_DBL;
  constexpr int LPB = 1 + C_::NU;  // DMA instructions per block
  __shared__ __attribute__((aligned(16))) double stage[2][4][D][64];  // [buffer][Y1..Y4][step][lane]
  __shared__ __attribute__((aligned(16))) double inp[NSLOT][SLOT];    // [slot]{records | u}
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t B = (size_t)a.batch;
  const int nb = a.N / D;
  const int bw = blockIdx.x * TPW;  // first trajectory of this workgroup

  if (wave == 0) {
    // ---------------- M: HBM -> LDS ----------------
    const double* recsrc = a.REC + 2 * lane;                       // records of block j start at REC + j*REC_DBL
    const int e0 = 2 * lane;                                       // element pair handled by this lane
    auto issue = [&](int j) OCS_INLINE {
      double* dst = &inp[j % NSLOT][0];
      if (2 * lane < C_::REC_DBL) dma16(recsrc + (size_t)j * C_::REC_DBL, dst);
#pragma unroll
      for (int q = 0; q < C_::NU; ++q) {
        const int e = q * 128 + e0, row = e / TPW, tl = e % TPW;
        dma16(a.u + ((size_t)(2 * D * j + 1 + row)) * B + bw + tl, dst + C_::REC_DBL + q * 128);
      }
    };
    for (int j = 0; j < Q && j < nb; ++j) issue(j);
    for (int k = 0; k <= nb; ++k) {
      if (k < nb) {
        const int behind = (nb - 1 - k) < (Q - 1) ? (nb - 1 - k) : (Q - 1);  // younger blocks in flight
        wait_blocks<LPB>(behind);
      }
      lds_barrier();
      if (k + Q < nb) issue(k + Q);
    }
  } else {
    const int r = lane % G;
    const int tl = lane / G;
    const int b = bw + tl;
    const uniform_ptr PS = as_uniform(a.ps);
    const typename P::RowPar rp = P::load_row([&](int k) OCS_INLINE {
      return ((a.pmask >> k) & 1u) ? a.pb[(size_t)k * B + b] : PS[k];
    }, r);
    const size_t colB = (size_t)NAUG * B;
    const double u0 = a.u[b];
    if (wave == 1) {
      // ---------------- S: state recursion ----------------
      double y = a.x0[(size_t)r * B + b];
      double uprev = u0;
      double* xs = a.x + (size_t)r * B + b;
      if (OUT_X) *xs = y;
      for (int k = 0; k <= nb; ++k) {
        lds_barrier();
        if (k < nb) {
          const double* rec = &inp[k % NSLOT][0];
          const double* us = rec + C_::REC_DBL + tl;
          double* w = &stage[k & 1][0][0][lane];
          // LDS reads of step s+1 are issued before step s is computed (LDS latency ~100 cycles would
          // otherwise sit on every step: the scheduler keeps loads next to their uses)
          struct In { double h, hh, h6, uM, uB; };
          auto fetch = [&](int s) OCS_INLINE {
            In v;
            v.h = rec[8 * s];
            v.hh = rec[8 * s + 1];
            v.h6 = rec[8 * s + 2];
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
            w[s * 64] = y;                                 // Y1 = y_i
            const double F1 = P::row_f(y, uprev, rp);
            double Y = __builtin_fma(c.hh, F1, y);
            w[(D + s) * 64] = Y;                           // Y2
            const double F2 = P::row_f(Y, c.uM, rp);
            Y = __builtin_fma(c.hh, F2, y);
            w[(2 * D + s) * 64] = Y;                       // Y3
            const double F3 = P::row_f(Y, c.uM, rp);
            Y = __builtin_fma(c.h, F3, y);
            w[(3 * D + s) * 64] = Y;                       // Y4
            const double F4 = P::row_f(Y, c.uB, rp);
            y = __builtin_fma(c.h6, __builtin_fma(2.0, F3, __builtin_fma(2.0, F2, F1)) + F4, y);
            if (OUT_X) {
              xs += colB;
              *xs = y;
            }
            uprev = c.uB;
          }
        }
      }
    } else {
      // ---------------- C: objective ----------------
      double pc = 0.0, uprev2 = u0 * u0;
      double* xc = a.x + (size_t)G * B + b;
      if (OUT_X) *xc = 0.0;
      for (int k = 0; k <= nb; ++k) {
        lds_barrier();
        if (k >= 1) {
          const int j = k - 1;
          const double* rec = &inp[j % NSLOT][0];
          const double* us = rec + C_::REC_DBL + tl;
          const double* w = &stage[j & 1][0][0][lane];
          struct In { double h6, tcA, tcM, tcB, uM, uB, Y1, Y2, Y3, Y4; };
          auto fetch = [&](int s) OCS_INLINE {
            In v;
            v.h6 = rec[8 * s + 2];
            v.tcA = rec[8 * s + 4];
            v.tcM = rec[8 * s + 5];
            v.tcB = rec[8 * s + 6];
            v.uM = us[(2 * s) * TPW];
            v.uB = us[(2 * s + 1) * TPW];
            v.Y1 = w[s * 64];
            v.Y2 = w[(D + s) * 64];
            v.Y3 = w[(2 * D + s) * 64];
            v.Y4 = w[(3 * D + s) * 64];
            return v;
          };
          In nxt = fetch(0);
#pragma unroll
          for (int s = 0; s < D; ++s) {
            const In c = nxt;
            if (s + 1 < D) nxt = fetch(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            const double uM2 = c.uM * c.uM, uB2 = c.uB * c.uB;
            const double q1 = c.tcA * P::row_q(c.Y1, uprev2, rp);
            const double q2 = c.tcM * P::row_q(c.Y2, uM2, rp);
            const double q3 = c.tcM * P::row_q(c.Y3, uM2, rp);
            const double q4 = c.tcB * P::row_q(c.Y4, uB2, rp);
            pc = __builtin_fma(c.h6, __builtin_fma(2.0, q3, __builtin_fma(2.0, q2, q1)) + q4, pc);
            if (OUT_X) {
              xc += colB;
              *xc = group_sum_pl<G>(pc);
            }
            uprev2 = uB2;
          }
        }
      }
      a.J[b] = group_sum_pl<G>(pc);
    }
  }
}

// ---------------------------------------------------------------------------------------
bool pipeline_supported(Functor f, int nS, int nC) {
  return f == Functor::Logistic && (nS == 1 || nS == 2 || nS == 4) && nC == 1;
}
bool pipeline_shape_ok(int nS, int N, int batch) {
  const int D = (nS == 4) ? 16 : 8, TPW = 64 / nS;
  return N >= D && N % D == 0 && batch % TPW == 0;
}

template <class P>
static void run_forward_pl(const FwdArgsPL& a, hipStream_t s) {
  constexpr int TPW = 64 / P::NS;
  const dim3 grid(a.batch / TPW), block(192);
  if (a.x)
    k_forward_pl<P, true><<<grid, block, 0, s>>>(a);
  else
    k_forward_pl<P, false><<<grid, block, 0, s>>>(a);
}
int launch_forward_pl(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                      double* x, double* J, hipStream_t s) {
  if (!pipeline_shape_ok(p.nS, g.N, batch)) return -1;
  const FwdArgsPL a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, x0, u, x, J};
  if (p.nS == 1)
    run_forward_pl<LogisticK<1>>(a, s);
  else if (p.nS == 2)
    run_forward_pl<LogisticK<2>>(a, s);
  else if (p.nS == 4)
    run_forward_pl<LogisticK<4>>(a, s);
  else
    return -1;
  return hip_rc5(hipGetLastError());
}

}  // namespace ocs
