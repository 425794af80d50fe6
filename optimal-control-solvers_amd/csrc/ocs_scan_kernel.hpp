// ocs_scan_kernel.hpp -- the discrete-adjoint pass (RK4Integrator.m:59-121) as a scan over time, for
// row-separable problems.
//
// The adjoint recursion is linear in lam: for a problem whose state rows are uncoupled (row r of F reads y_r
// and u only, ocs_problems.hpp ROW_SEPARABLE) and whose cost row of lam is constant (last row of
// dFdx_times_vec is zero, OCProblem.m:14-15), every step is a scalar affine map per row,
//     lam_i = alpha_i lam_{i+1} + beta_i,
// whose coefficients depend on the stage states Y1..Y4 of step i only -- and those are recomputed from the
// checkpoint x(t_i) and the control samples, independently for every step.  So the 1000-step serial chain of
// the pass disappears:
//
//   phase 1  (time-parallel)  wave w of a workgroup takes a chunk of L consecutive steps of the workgroup's
//            64 (trajectory, row) lanes, recomputes Y2..Y4, forms (alpha_i, beta_i) by pushing the pair
//            (coefficient of lam, constant) through RK4Integrator.m:73-88, and composes them into the chunk
//            map (A_w, B_w):  lam_lo = A_w lam_hi+1 + B_w.
//   phase 2  (W-term scan)    the W chunk maps of a superblock (W chunks = W L steps) go through LDS; every
//            wave composes them in time order on top of the carry (lam at the top of the superblock), which
//            gives it lam at the top of its own chunk and the next carry.  One LDS barrier per superblock.
//   phase 3  (time-parallel)  with the true lam at the top of its chunk a wave runs lines :73-88 exactly as the
//            reference writes them (stage states recomputed once more; inside a chunk the arithmetic is the
//            serial kernels'), stores lam and
//            assembles the dJdu columns (:97-121).
//
// Superblocks are taken from the end of the horizon; the loads of superblock k+1 are issued before superblock k
// is processed (two register sets), so HBM latency lies under a whole superblock of arithmetic and the pass
// streams: per (trajectory, step) it reads x (nS doubles) and two new control samples and writes lam (nAug)
// and two dJdu columns.  Steps below 0 in the last superblock run as exact identity maps (records with
// h = 0 on clamped inputs), so any N works; stores are predicated, so any batch works.
//
// Differences to the serial kernels: lam at the chunk boundaries comes from composed maps, i.e. a different
// association of the same products and sums (round-off level; tolerance 1e-12 as for the other mappings).
// The composed products prod(alpha) must stay inside the fp64 range.
// This header holds the kernel template only (it is also compiled by hipRTC for user problems given as row functions,
// csrc/ocs_user_functor.hpp); the launchers are in ocs_scan_kernels.hip.  Functor interface: the g_* names of
// ocs_problems.hpp ("generic row-separable interface").
#pragma once
#include "ocs_device_common.hpp"

namespace ocs {

// Lane layout: lane = r * (64/G) + tl, the trajectory index fastest.  The 64/G lanes of a state row read and
// write one contiguous segment, and above all the four lanes of every quad touch ONE cache line: the texture
// addresser works through a wave's addresses a quad at a time, and with the row index fastest (lane = tl G + r,
// the layout of the row-split / pipeline kernels) every quad spans G lines -- measured 3.5 TB/s against 6 TB/s
// for the same bytes.  The price: sums over the rows of a trajectory cross lanes 16 or 32 apart, which DPP
// cannot reach; they use the row-swapping permlane instructions of gfx950 (swap_add16 / swap_add32 below).
// v_permlane16_swap(vdst, src): the odd 16-lane rows of vdst trade places with the even rows of src;
// v_permlane32_swap: the upper half of vdst with the lower half of src.  With (a, b) in, the two results added give
// per row [a0+a1, b0+b1, a2+a3, b2+b3] -- a transposing pair reduction in two instructions per 32-bit half, in the
// vector pipe (ds_bpermute, the alternative, parks the wave for an LDS round trip: 13 % of its cycles were measured).
__device__ static inline double swap_add16(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ static inline double swap_add32(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// sum over the G rows of a trajectory (lanes r TPW + tl), in every lane
template <int G>
__device__ static inline double group_sum_sc(double v) {
  static_assert(G == 1 || G == 2 || G == 4, "group size");
  if (G == 4) v = swap_add16(v, v);
  if (G >= 2) v = swap_add32(v, v);
  return v;
}
// two sums at once: lanes of even row r get sum_r a, lanes of odd row r get sum_r b
template <int G>
__device__ static inline double pair_sum_sc(double a, double b) {
  static_assert(G == 1 || G == 2 || G == 4, "group size");
  if (G == 1) return a;   // (not used: one lane per trajectory has nothing to sum)
  if (G == 4) {
    const double s = swap_add16(a, b);   // rows: [a0+a1, b0+b1, a2+a3, b2+b3]
    return swap_add32(s, s);             //       [sum a, sum b, sum a, sum b]
  }
  return swap_add32(a, b);               // halves (= rows): [a0+a1, b0+b1]
}
__device__ static inline void lds_barrier_sc() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct BwdArgsScan {
  int N, batch;          // N: a multiple of L (the launcher gives the remainder to the lane kernel)
  const double* RECS;    // step records (kScanRec doubles each), zero records around [0, N)
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* xck;
  const double* u;
  const double* lamT;
  double* lam;
  double* dJdu;
  double* lam0;
  const double* pend0;   // optional [B]: the k1 half of column 2N when the steps above N were done by another kernel
};

// Buffer addressing (raw SRSRC, 32-bit per-lane voffset + scalar soffset): a lane's part of every address
// ((row r, trajectory b) of a column) is fixed for the whole kernel and lives in one VGPR per array; the
// column (time) part is wave-uniform and goes through the scalar offset, so address arithmetic costs no vector
// registers or instructions.  A lane is switched off without a branch by an offset beyond num_records (the
// range check of a raw buffer compares the vector offset only and drops the access), a whole chunk by a
// descriptor with num_records = 0.
typedef unsigned v2u_sc __attribute__((ext_vector_type(2)));
constexpr unsigned kOffDrop = 0xFFFFFFF0u;  // >= num_records of every descriptor below
constexpr int kNumRec = 0x7FFFFFF0;
struct Buf {
  __amdgpu_buffer_rsrc_t r;
  __device__ static inline Buf make(const double* p, int nrec = kNumRec) {
    return Buf{__builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, nrec, 0x00020000)};
  }
  __device__ inline double ld(unsigned voff, unsigned soff) const {
#ifndef OCS_SCAN_LD_AUX
#define OCS_SCAN_LD_AUX 0
#endif
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, OCS_SCAN_LD_AUX));
  }
  __device__ inline void st0(double v, unsigned voff, unsigned soff) const {   // plain (cached) store
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_sc, v), r, voff, soff, 0);
  }
  __device__ inline void st(double v, unsigned voff, unsigned soff) const {
#ifndef OCS_SCAN_ST_AUX
#define OCS_SCAN_ST_AUX 2   // nt: see the note on non-temporal stores at the top of the file
#endif
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_sc, v), r, voff, soff, OCS_SCAN_ST_AUX);
  }
};
// 16-byte-per-lane LDS-DMA: lane l copies src_l[0..1] to lds_base[2l..2l+1]
__device__ static inline void dma16_sc(const double* src, double* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

constexpr int kScanRec = 16;       // doubles per record: {h, h/2, h/6, h/3 | s4, s3, s1, 0 | tcA, tcM, tcB, 0 | pad}
constexpr int kScanPadFront = 136; // zero records before step 0 (>= the steps of a superblock + 1: 16 x 4 in
                                   // k_backward_scan, 8 x 4 x 4 in k_backward_fcs)
constexpr int kScanPadBack = 8;    // and after step N-1 (a wave copies 8 records per chunk)
// Checkpoints of a chunk's upper L - 1 steps integrated again from the first instead of read (k_backward_scan): 8 nS / L
// bytes of checkpoint traffic per (trajectory, step) instead of 8 nS, for four more evaluations of the row's right-hand
// side.  Measured at BL-2 (batch 4096, nS = 4): buffers that rotate through HBM 94.5 -> 89.2 us, one buffer set re-used
// from the memory-side cache 79.9 -> 81.0 us; nS = 1: 70.3 -> 66.8 / 69.2 -> 66.7 us (profiles/r04f_xrc.log).
#ifndef OCS_SCAN_XRC
#define OCS_SCAN_XRC 1
#endif
constexpr bool kScanXRC = OCS_SCAN_XRC != 0;

// W waves per workgroup (chunks per superblock), L steps per chunk
// ABL (diagnostic builds, -DOCS_SCAN_ABL): 1 no stores, 2 no phase 3, 3 no phase 1, 4 no loads, 5 no barrier/phase 2
template <class P, int W, int L, bool OUT_LAM, bool OUT_DJDU, bool LT, int ABL = 0>
__global__ __launch_bounds__(W * 64) void k_backward_scan(const BwdArgsScan a) {
  constexpr int G = P::NS, NAUG = P::NAUG, TPW = 64 / G;
  static_assert(P::NC == 1 && P::NTC == 1 && P::ROW_SEPARABLE, "scan kernels: row-separable problems, one control");
  static_assert(L % G == 0 && W * L + 1 <= kScanPadFront && L + 1 <= 8, "chunk shape");
  typedef typename P::Stage Stage;
  __shared__ __attribute__((aligned(16))) double2 sm[2][W][64];      // chunk maps
  __shared__ double csm[2][64];                                      // lam at the bottom of a superblock
  __shared__ __attribute__((aligned(16))) double rcs[2][W][8 * kScanRec];   // records lo-1 .. lo+6 of a wave's chunk (1 KiB)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane / TPW, tl = lane % TPW;   // trajectory fastest (see group_sum_sc)
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  const int b0 = blockIdx.x * TPW + tl;
  const bool valid = b0 < a.batch;
  const int b = valid ? b0 : a.batch - 1;
  const uniform_ptr PS = as_uniform(a.ps);
  const typename P::RowPar rp = P::load_row(ParamSrc{PS, a.pb, a.pmask, B, b}, r);
  const double lamc = LT ? a.lamT[(size_t)G * B + b] : 1.0;
  const size_t colB = (size_t)NAUG * B;
  double carry = LT ? a.lamT[(size_t)r * B + b] : 0.0;   // lam(r, N+1)   :63-66
  if (OUT_LAM && wave == 0 && valid) {
    a.lam[(size_t)N * colB + (size_t)r * B + b] = carry;
    if (r == 0) a.lam[(size_t)N * colB + (size_t)G * B + b] = lamc;
  }
  const double pend_top = (OUT_DJDU && a.pend0 && r == 0) ? a.pend0[b] : 0.0;
  // per-lane parts of the addresses (bytes); the column parts are wave-uniform scalar offsets
  const unsigned col8 = (unsigned)(colB * 8), B8 = (unsigned)(B * 8);
  const unsigned vx = (unsigned)(((size_t)r * B + b) * 8);                 // x loads (clamped lane: valid memory)
  const unsigned vu = (unsigned)((size_t)b * 8);                           // u loads
  const unsigned vl = valid ? vx : kOffDrop;                               // lam stores
  const unsigned vc = valid ? (unsigned)(((size_t)G * B + b) * 8) + (unsigned)r * col8 : kOffDrop;  // cost row of lam:
                                                                           // lane (trajectory, r) takes column lo + r (+ G k)
  // dJdu stores of step i = lo + q at scalar offset 2 q B: lane r = 0 the midpoint column 2i+1, r = 1 (or the
  // second store, G = 1) the node column 2i+2, r = 2 (G = 4) the chunk's lowest node column 2 lo with q = 0
  const bool is0 = r == 0, is1 = r == 1;
  const unsigned vd_mid = valid ? vu + B8 : kOffDrop, vd_node = valid ? vu + 2 * B8 : kOffDrop;
  const unsigned vd_bot = valid ? vu : kOffDrop;
  const unsigned vd_r = (G == 1) ? vd_mid : is0 ? vd_mid : is1 ? vd_node : kOffDrop;  // steps q > 0
  const unsigned vd_r0 = (G == 4 && r == 2) ? vd_bot : vd_r;                          // step q = 0

  struct Ld {
    double x[L];          // x(r, lo+q)
    double u[2 * L + 1];  // u(2 lo + k)
    double xb, ub0, ub1;  // DFDU_READS_Y only: x(r, lo-1), u(2 lo - 2), u(2 lo - 1) (stage state 4 of the step below)
  };
  // chunk of wave `wave` in superblock sb: steps lo .. hi; below step 0 a chunk is dead as a whole (N % L == 0)
  auto chunk_lo = [&](int sb) OCS_INLINE { return N - (sb * W + wave + 1) * L; };
  // The loads of a chunk in L parts (part q: the records with q = 0, x(lo+q), u(2 lo + 2q), u(2 lo + 2q + 1),
  // and u(2 lo + 2L) with the last part), so that they can be issued between the steps of the previous superblock's
  // arithmetic: sixteen waves issuing 14 loads each back to back fill the address queue and stall at issue.
  auto load_part = [&](int sb, Ld& d, int slot, int q) OCS_INLINE {
    const int lo = chunk_lo(sb), lc = lo > 0 ? lo : 0;
    // the records first: older in the in-order queue than the loads the compiler waits for
    // (a dead chunk -- below step 0, possibly far below in a superblock past the horizon -- takes zero records
    //  from the front pad: identity maps)
    const int lr = lo >= 0 ? lo - 1 : -kScanPadFront;
    if (q == 0) dma16_sc(a.RECS + (long long)lr * kScanRec + 2 * lane, &rcs[slot][wave][0]);
    const Buf bx = Buf::make(a.xck + (size_t)lc * colB, ABL == 4 ? 0 : kNumRec), bu = Buf::make(a.u + (size_t)(2 * lc) * B, ABL == 4 ? 0 : kNumRec);
    if (!kScanXRC || q == 0) d.x[q] = bx.ld(vx, (unsigned)q * col8);
    d.u[2 * q] = bu.ld(vu, (unsigned)(2 * q) * B8);
    d.u[2 * q + 1] = bu.ld(vu, (unsigned)(2 * q + 1) * B8);
    if (q == L - 1) d.u[2 * L] = bu.ld(vu, (unsigned)(2 * L) * B8);
    if (P::DFDU_READS_Y && q == 0) {
      const int lb = lo > 0 ? lo - 1 : 0;   // (lo = 0: the values are multiplied by a zero record)
      const Buf bxb = Buf::make(a.xck + (size_t)lb * colB), bub = Buf::make(a.u + (size_t)(2 * lb) * B);
      d.xb = bxb.ld(vx, 0);
      d.ub0 = bub.ld(vu, 0);
      d.ub1 = bub.ld(vu, B8);
    }
  };
  auto load = [&](int sb, Ld& d, int slot) OCS_INLINE {
#pragma unroll
    for (int q = 0; q < L; ++q) load_part(sb, d, slot, q);
  };
  constexpr int NST = (OUT_LAM ? L + L / G : 0) + (OUT_DJDU ? (G == 1 ? 2 * L + 1 : (G == 2 ? L + 1 : L)) : 0);
  struct Rc { double h, hh, h6, h3, s4, s3, s1, tA, tM, tB; };
  auto rec_of = [&](const double* w, int q) OCS_INLINE {   // record of step lo + q (q = -1: the step below the chunk)
    const double2* p = reinterpret_cast<const double2*>(w + (q + 1) * kScanRec);
    const double2 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5];   // (unused fields cost nothing)
    return Rc{a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a4.x, a4.y, a5.x};
  };

  auto process = [&](int sb, const Ld& d, int slot, bool first, Ld& dn) OCS_INLINE {
    const int lo = chunk_lo(sb);
    const bool live = lo >= 0, topc = lo + L == N;
    // the records have landed once everything up to this superblock's loads has (in-order vmcnt); younger: the
    // previous superblock's stores (the loads of the next superblock are issued inside phase 1, after this wait)
    if (first)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    const double* rw = &rcs[slot][wave][0];
    // ---------------- phase 1: stage states and the chunk map ----------------
    // Steps in ascending order: (kScanXRC) only the checkpoint of the chunk's first step was read, the next one comes out
    // of the stage evaluations the step map needs anyway plus one more right-hand side (RK4Integrator.m:49-50 on this row)
    // -- 8 nS / L instead of 8 nS bytes of checkpoint traffic per (trajectory, step).  The chunk map is composed from the
    // bottom: lam_lo = A lam_(above the steps so far) + Bq.
    double xs[L];
    xs[0] = d.x[0];
    double A = 1.0, Bq = 0.0;
#pragma unroll
    for (int q = 0; q < L && ABL != 3; ++q) {
      const Rc c = rec_of(rw, q);
      const double xi = xs[q], uA = d.u[2 * q], uM = d.u[2 * q + 1], uB = d.u[2 * q + 2];
      const double F1 = P::g_row_f(xi, uA, c.tA, rp);           // compute_states :39-46, this row
      const double Y2 = __builtin_fma(c.hh, F1, xi);
      const double F2 = P::g_row_f(Y2, uM, c.tM, rp);
      const double Y3 = __builtin_fma(c.hh, F2, xi);
      const double F3 = P::g_row_f(Y3, uM, c.tM, rp);
      const double Y4 = __builtin_fma(c.h, F3, xi);
      if (q + 1 < L) {
        if (kScanXRC) {
          const double F4 = P::g_row_f(Y4, uB, c.tB, rp);
          xs[q + 1] = __builtin_fma(c.h6, ((F1 + 2.0 * F2) + 2.0 * F3) + F4, xi);   // :50
        } else {
          xs[q + 1] = d.x[q + 1];
        }
      }
      const Stage s4 = P::template stage<LT>(c.s4, c.h6, c.tB, lamc), s3 = P::template stage<LT>(c.s3, c.h3, c.tM, lamc),
                  s1 = P::template stage<LT>(c.s1, c.h6, c.tA, lamc);
      // row of (dF/dy)'v = a v_r + b; (p, q): the quantity is p lam_{i+1} + q          :73-88
      double a4, b4, a3, b3, a2, b2, a1, b1;
      P::g_row_dfdx_pre(Y4, uB, s4, rp, a4, b4);
      const double g3p = a4 * c.h6, g3q = b4;                                   // k4 = (h6, 0)
      const double k3p = __builtin_fma(c.h, g3p, c.h3), k3q = c.h * g3q;
      P::g_row_dfdx_pre(Y3, uM, s3, rp, a3, b3);
      const double g2p = a3 * k3p, g2q = __builtin_fma(a3, k3q, b3);
      const double k2p = __builtin_fma(c.hh, g2p, c.h3), k2q = c.hh * g2q;
      P::g_row_dfdx_pre(Y2, uM, s3, rp, a2, b2);
      const double g1p = a2 * k2p, g1q = __builtin_fma(a2, k2q, b2);
      const double k1p = __builtin_fma(c.hh, g1p, c.h6), k1q = c.hh * g1q;
      P::g_row_dfdx_pre(xi, uA, s1, rp, a1, b1);
      const double g0p = a1 * k1p, g0q = __builtin_fma(a1, k1q, b1);
      const double alpha = (((1.0 + g1p) + g2p) + g3p) + g0p;
      const double beta = ((g1q + g2q) + g3q) + g0q;
      Bq = __builtin_fma(A, beta, Bq);
      A = A * alpha;
      __builtin_amdgcn_sched_barrier(0);   // one step at a time: the temporaries of interleaved steps cost occupancy
      load_part(sb + 1, dn, slot ^ 1, q);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ABL == 3) {
#pragma unroll
      for (int q = 1; q < L; ++q) xs[q] = xs[0];
    }
    sm[sb & 1][wave][lane] = double2{A, Bq};
    if (ABL != 5) lds_barrier_sc();
    // ---------------- phase 2: lam at the top of this chunk ----------------
    // lam at the top of the superblock: lamT for the first one, afterwards what the LAST wave of the previous
    // superblock left at the bottom of its chunk (published to LDS behind its phase 3, i.e. before this superblock's
    // barrier) -- so every wave composes only the maps of the chunks above its own, in groups of four (all W at once
    // would hold 4 W registers), and superblock boundaries carry the serial recursion's own value.
    double lam = (sb == 0) ? carry : csm[(sb & 1) ^ 1][lane];
#pragma unroll
    for (int j0 = 0; j0 < W && ABL != 5; j0 += 4) {
      if (j0 < wave) {   // wave-uniform
        double2 ab[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ab[j] = sm[sb & 1][j0 + j][lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j0 + j < wave) {   // wave-uniform; the fence keeps it a branch (two selects per map otherwise)
            lam = __builtin_fma(ab[j].x, lam, ab[j].y);
            asm volatile("" : "+v"(lam));
          }
        }
      }
    }
    // ---------------- phase 3: the recursion inside the chunk, lam and dJdu stores ----------------
    const int lc = live ? lo : 0, nrec = (live && ABL != 1) ? kNumRec : 0;   // a dead chunk stores nothing
    const Buf bl = Buf::make(a.lam + (size_t)lc * colB, nrec), bd = Buf::make(a.dJdu + (size_t)(2 * lc) * B, nrec);
    if (OUT_LAM) {
      // the constant cost row of lam for the L columns of the chunk
#pragma unroll
      for (int q0 = 0; q0 < L; q0 += G) bl.st(lamc, vc, (unsigned)q0 * col8);
    }
    double pend = topc ? pend_top : 0.0;   // this row's B'k1 share of the node above
    if (ABL == 2) lam += xs[0] + d.u[0];
#pragma unroll
    for (int q = L - 1; q >= 0 && ABL != 2; --q) {
      const Rc c = rec_of(rw, q);
      // the stage states again (held registers are worth more than these nine operations: 4 waves per SIMD)
      const double xi = xs[q], uA = d.u[2 * q], uM = d.u[2 * q + 1], uB = d.u[2 * q + 2];
      double f = P::g_row_f(xi, uA, c.tA, rp);
      const double Y2 = __builtin_fma(c.hh, f, xi);
      f = P::g_row_f(Y2, uM, c.tM, rp);
      const double Y3 = __builtin_fma(c.hh, f, xi);
      f = P::g_row_f(Y3, uM, c.tM, rp);
      const double Y4 = __builtin_fma(c.h, f, xi);
      const Stage s4 = P::template stage<LT>(c.s4, c.h6, c.tB, lamc), s3 = P::template stage<LT>(c.s3, c.h3, c.tM, lamc),
                  s1 = P::template stage<LT>(c.s1, c.h6, c.tA, lamc);
      const double h6l = c.h6 * lam, h3l = c.h3 * lam;
      const double k4 = h6l;                                     // :73
      const double g3 = P::g_row_dfdx(Y4, uB, k4, s4, rp);       // :74-75
      const double k3 = __builtin_fma(c.h, g3, h3l);             // :77
      const double g2 = P::g_row_dfdx(Y3, uM, k3, s3, rp);       // :78-79
      const double k2 = __builtin_fma(c.hh, g2, h3l);            // :81
      const double g1 = P::g_row_dfdx(Y2, uM, k2, s3, rp);       // :82-83
      const double k1 = __builtin_fma(c.hh, g1, h6l);            // :85
      const double g0 = P::g_row_dfdx(xi, uA, k1, s1, rp);       // :87-88
      lam = (((lam + g1) + g2) + g3) + g0;                       // :86-88
      if (OUT_LAM) bl.st(lam, vl, (unsigned)q * col8);
      if (OUT_DJDU) {                                            // compute_dJdu :97-121
        const double p4 = P::g_row_dfdu(Y4, uB, k4, s4, rp);
        const double p23 = P::g_row_dfdu(Y3, uM, k3, s3, rp) + P::g_row_dfdu(Y2, uM, k2, s3, rp);
        // column 2i+1 (sum of p23 over the rows) and column 2i+2 (sum of pend + p4; it belongs to the chunk of step
        // i+1, except column 2N): lanes of row 0 end up with the first, lanes of row 1 with the second
        double cmid, cnode;
        if (G == 1) {
          cmid = p23;
          cnode = pend + p4;
        } else {
          cmid = cnode = pair_sum_sc<G>(p23, pend + p4);
        }
        pend = P::g_row_dfdu(xi, uA, k1, s1, rp);
        const unsigned so = (unsigned)(2 * q) * B8;
        double cbot = 0.0;
        if (q == 0) {
          // column 2 lo = B'k1 of step lo + B'k4 of step lo-1 (k4 = h/6 lam_lo); column 0 has the k1 half only
          // :101-102.  Where (dF/du)'v reads y, stage state 4 of the step below is recomputed from three extra loads.
          const Rc cb = rec_of(rw, -1);
          const Stage sb4 = P::template stage<LT>(cb.s4, cb.h6, cb.tB, lamc);
          double Y4b = 0.0;
          if (P::DFDU_READS_Y) {
            double fb = P::g_row_f(d.xb, d.ub0, cb.tA, rp);
            const double Y2b = __builtin_fma(cb.hh, fb, d.xb);
            fb = P::g_row_f(Y2b, d.ub1, cb.tM, rp);
            const double Y3b = __builtin_fma(cb.hh, fb, d.xb);
            fb = P::g_row_f(Y3b, d.ub1, cb.tM, rp);
            Y4b = __builtin_fma(cb.h, fb, d.xb);
          }
          cbot = group_sum_sc<G>(pend + P::g_row_dfdu(Y4b, uA, cb.h6 * lam, sb4, rp));
        }
        if (G == 1) {
          bd.st(cmid, vd_mid, so);
          if (q == L - 1)
            bd.st(cnode, topc ? vd_node : kOffDrop, so);
          else
            bd.st(cnode, vd_node, so);
          if (q == 0) bd.st(cbot, vd_bot, so);
        } else {
          const double val = is0 ? cmid : (is1 ? cnode : cbot);
          if (q == L - 1)
            bd.st(val, (topc | is0) ? vd_r : kOffDrop, so);
          else if (q == 0 && G == 4)
            bd.st(val, vd_r0, so);
          else
            bd.st(val, vd_r, so);
          if (q == 0 && G == 2) bd.st(cbot, is0 ? vd_bot : kOffDrop, so);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wave == W - 1) csm[sb & 1][lane] = lam;   // lam at the bottom of the superblock (a dead chunk passes it through)
    carry = lam;
  };

  // Superblocks in pairs (two register sets, no register moves); a superblock past the horizon loads clamped
  // addresses and processes identity maps without a store, so the loop needs no conditions.
  const int nsb = (N + W * L - 1) / (W * L);
  Ld d0, d1;
  load(0, d0, 0);
  for (int sb = 0; sb < nsb; sb += 2) {
    process(sb, d0, 0, sb == 0, d1);
    process(sb + 1, d1, 1, false, d0);
  }
  if (a.lam0 && wave == W - 1 && valid) {   // the last wave's chunk ends at (or, dead, passes through) step 0
    a.lam0[(size_t)r * B + b] = carry;
    if (r == 0) a.lam0[(size_t)G * B + b] = lamc;
  }
}

}  // namespace ocs
