// ocs_fused_control_kernels.hip -- the shooting objective and its gradient with the control basis
// inside the RK4 kernels ("fused-control mode"):
//
//   functions/single_shooting.m:137-150    u = compute_u(v);  [~,J] = compute_states(u);
//                                          [~,dJdu] = compute_adjoints(u);  dJdv = compute_dJdv(dJdu)
//   Control/ChebyshevControl.m:35-43       u = reshape(v,nC,[]) * B,   dJdv = dJdu * B'
//
// For a dense basis with few functions (Chebyshev: nBasis <= 32) neither u (2N+1 samples per trajectory)
// nor dJdu ever exists in memory: the forward kernel evaluates u(:,j) = sum_k v_k B(k,j) where it is
// consumed and the adjoint kernel folds every finished column of dJdu into 16 (or 32) running sums
// dJdv_k += dJdu(:,j) B(k,j).  HBM traffic per (trajectory, step) drops from 8 (2 nAug + 6 nC + basis
// kernels' 4 nC) bytes to the checkpoint write + read, 16 nS bytes.
//
// Mapping: lane per trajectory as in ocs_rk4_kernels.hpp (one wave = 64 trajectories).  A column
// B(:,j) of the basis is wave-uniform.  It lives in ONE vector register pair: lane l holds
// B(l & 15, j) (loaded coalesced from the transposed table, prefetched a chunk ahead like the control
// samples used to be), and `v_fmac_f64_dpp ... row_newbcast:k` multiplies by lane k's value of the own
// 16-lane row, i.e. by B(k,j) -- gfx950's only DPP mode for fp64, exactly a broadcast.  No scalar loads
// (they return out of order and would need ~100 SGPRs for the rows in flight), no LDS.
//
// Tried in round 3 and dropped: the two products with the basis matrix as v_mfma_f64_16x16x4_f64 tiles in this lane mapping
// (tiles transposed through wave-private LDS, the matrix instructions spread between the steps).  On gfx950 the fp64 matrix
// instruction occupies the SIMD's fp64 datapath for its 64 cycles -- it does not run beside fp64 vector instructions, of
// the same wave or of another (scripts/probe/mfma_valu_overlap.hip: 1 MFMA + 16 FMAs take 188-200 cycles, 64 + 104 apart)
// -- so the 1024 multiply-adds of a tile cost what sixteen v_fmac_f64_dpp cost, plus the transposition: 486 us against
// 426 us per evaluation at batch 64, 667 against 604 us at batch 65 536.
#include "ocs_device_common.hpp"
#include "ocs_internal.hpp"
#include "ocs_problems.hpp"

namespace ocs {

static inline int hip_rc_fc(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// One group of 16 basis functions.  Each product is `v_fmac_f64_dpp acc, brow, x row_newbcast:k`:
// acc += (lane k of the own 16-lane row of brow) * x = B(k,j) * x.  All 16 sit in ONE asm statement that starts
// with s_nop 1: the compiler's hazard recogniser does not look into inline assembly, and a DPP instruction must
// not read a VGPR written by the two VALU instructions before it.  EXEC is all ones wherever these run (lanes
// past the batch are clamped, not masked).
template <int NC>
struct FcRow16 {
  // u_c += sum_k B(k,j) v[k][c]
  __device__ static inline void dot(const double brow, const double (&vv)[16][NC], double (&u)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double acc = u[c];
      asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
          : "+v"(acc)
          : "v"(brow), "v"(vv[0][c]), "v"(vv[1][c]), "v"(vv[2][c]), "v"(vv[3][c]), "v"(vv[4][c]), "v"(vv[5][c]), "v"(vv[6][c]), "v"(vv[7][c]), "v"(vv[8][c]), "v"(vv[9][c]), "v"(vv[10][c]), "v"(vv[11][c]), "v"(vv[12][c]), "v"(vv[13][c]), "v"(vv[14][c]), "v"(vv[15][c]));
      u[c] = acc;
    }
  }
  // g[k][c] += B(k,j) d[c]
  __device__ static inline void axpy(const double brow, const double (&d)[NC], double (&g)[16][NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %16, %17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %16, %17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %16, %17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, %16, %17 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %5, %16, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %6, %16, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %7, %16, %17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %8, %16, %17 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %9, %16, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %10, %16, %17 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %11, %16, %17 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %12, %16, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %13, %16, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %14, %16, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %15, %16, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
          : "+v"(g[0][c]), "+v"(g[1][c]), "+v"(g[2][c]), "+v"(g[3][c]), "+v"(g[4][c]), "+v"(g[5][c]), "+v"(g[6][c]), "+v"(g[7][c]), "+v"(g[8][c]), "+v"(g[9][c]), "+v"(g[10][c]), "+v"(g[11][c]), "+v"(g[12][c]), "+v"(g[13][c]), "+v"(g[14][c]), "+v"(g[15][c])
          : "v"(brow), "v"(d[c]));
    }
  }
};

struct FcArgs {
  int N, batch, nBasis;
  const double* REC;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* BT;   // [2N+1][16 NG]: transposed basis, zero-padded to NG groups of 16 functions
  const double* v;    // [nBasis][nC][B]
  const double* x0;   // forward: [nS][B]
  double* ck;         // forward out / backward in: checkpoints [N+1][nAug][B] (state rows only are touched)
  double* J;          // forward out [B]
  double* dJdv;       // backward out [nBasis][nC][B]
  double* lam0;       // backward out, optional [nAug][B]: lam(:,1)  (single_shooting.m:149)
};

template <class P, int NG>
__device__ static inline void fc_load_v(const FcArgs& a, size_t B, int b, double (&vv)[NG][16][P::NC]) {
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int c = 0; c < P::NC; ++c) {
        const int kk = 16 * g + k;
        const double val = a.v[((size_t)(kk < a.nBasis ? kk : 0) * P::NC + c) * B + b];
        vv[g][k][c] = kk < a.nBasis ? val : 0.0;
      }
}

// ---------------------------------------------------------------------------------------
// J = x(end,end) of compute_states(u = v*B)   RK4Integrator.m:28-56, checkpoints y_i kept for the adjoint
// ---------------------------------------------------------------------------------------
template <class P, int CH, int PF, int NG>
__global__ __launch_bounds__(64) void k_forward_fc(const FcArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC, NAUG = P::NAUG, LD = 16 * NG;
  using Rec = StepRec<NTC>;
  const int lane = threadIdx.x;
  const int b0 = blockIdx.x * 64 + lane;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));

  double vv[NG][16][NC];
  fc_load_v<P, NG>(a, B, b, vv);
  auto u_of = [&](const double (&row)[NG], double (&u)[NC]) OCS_INLINE {
#pragma unroll
    for (int c = 0; c < NC; ++c) u[c] = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) FcRow16<NC>::dot(row[g], vv[g], u);
  };

  double y[NS], yc = 0.0;
#pragma unroll
  for (int k = 0; k < NS; ++k) y[k] = a.x0[(size_t)k * B + b];
  double* xo = a.ck + b;
#pragma unroll
  for (int k = 0; k < NS; ++k) xo[(size_t)k * B] = y[k];
  xo += (size_t)NAUG * B;

  const double* bp = a.BT + (lane & 15);  // this lane's entry of basis column j: bp[j * LD + 16 g]
  auto load_row = [&](double (&row)[NG]) OCS_INLINE {
#pragma unroll
    for (int g = 0; g < NG; ++g) row[g] = bp[16 * g];
    bp += LD;
  };
  double uprev[NC];
  {
    double r0[NG];
    load_row(r0);
    u_of(r0, uprev);
  }

  auto step = [&](const Rec& r, const double* uA, const double* uM, const double* uB) OCS_INLINE {
    double F1[NS + 1], F2[NS + 1], F3[NS + 1], F4[NS + 1], Y[NS];
    P::F(r.tcA, y, uA, p, F1);                                             // :39
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.hh, F1[k], y[k]);  // :40
    P::F(r.tcM, Y, uM, p, F2);                                             // :42
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.hh, F2[k], y[k]);  // :43
    P::F(r.tcM, Y, uM, p, F3);                                             // :45
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.h, F3[k], y[k]);   // :46
    P::F(r.tcB, Y, uB, p, F4);                                             // :48
#pragma unroll
    for (int k = 0; k < NS; ++k)                                           // :50-51
      y[k] = __builtin_fma(r.h6, __builtin_fma(2.0, F3[k], __builtin_fma(2.0, F2[k], F1[k])) + F4[k], y[k]);
    yc = __builtin_fma(r.h6, __builtin_fma(2.0, F3[NS], __builtin_fma(2.0, F2[NS], F1[NS])) + F4[NS], yc);
#pragma unroll
    for (int k = 0; k < NS; ++k) xo[(size_t)k * B] = y[k];  // the cost row of a checkpoint is never read
    xo += (size_t)NAUG * B;
  };

  static_assert(PF <= kRecPad, "record ring deeper than the table padding");
  Rec rq[PF];
  const double* recp = a.REC;
#pragma unroll
  for (int q = 0; q < PF; ++q) {
    rq[q] = load_rec<NTC>(recp);
    recp += rec_stride(NTC);
  }
  auto next_rec = [&]() OCS_INLINE {
    const Rec cur = rq[0];
#pragma unroll
    for (int q = 0; q + 1 < PF; ++q) rq[q] = rq[q + 1];
    rq[PF - 1] = load_rec<NTC>(recp);
    recp += rec_stride(NTC);
    return cur;
  };

  // basis columns are prefetched one chunk (CH steps = 2 CH columns) ahead, ping-pong in registers
  double rb0[2 * CH][NG], rb1[2 * CH][NG];
  auto load_chunk = [&](double (&dst)[2 * CH][NG]) OCS_INLINE {
#pragma unroll
    for (int s = 0; s < 2 * CH; ++s) load_row(dst[s]);
  };
  auto run_chunk = [&](const double (&src)[2 * CH][NG]) OCS_INLINE {
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      double uM[NC], uB[NC];
      u_of(src[2 * s], uM);
      u_of(src[2 * s + 1], uB);
      const Rec cur = next_rec();
      step(cur, uprev, uM, uB);
#pragma unroll
      for (int c = 0; c < NC; ++c) uprev[c] = uB[c];
    }
  };
  const int nch = N / CH;
  if (nch > 0) load_chunk(rb0);
  int c = 0;
  for (; c + 1 < nch; c += 2) {
    load_chunk(rb1);
    run_chunk(rb0);
    if (c + 2 < nch) load_chunk(rb0);
    run_chunk(rb1);
  }
  if (c < nch) run_chunk(rb0);
  for (int i = nch * CH; i < N; ++i) {  // remainder steps
    double rM[NG], rB[NG], uM[NC], uB[NC];
    load_row(rM);
    load_row(rB);
    u_of(rM, uM);
    u_of(rB, uB);
    const Rec cur = next_rec();
    step(cur, uprev, uM, uB);
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) uprev[cc] = uB[cc];
  }
  a.J[b] = yc;  // J = x(end,end)   :55
  if (warm == 1.234567e300) a.J[b] = warm;
}

// ---------------------------------------------------------------------------------------
// dJdv = compute_dJdv(compute_adjoints(u = v*B))   RK4Integrator.m:59-121, ChebyshevControl.m:41-43
// ---------------------------------------------------------------------------------------
template <class P, int CH, int PF, int NG>
__global__ __launch_bounds__(64) void k_backward_fc(const FcArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC, NAUG = P::NAUG, LD = 16 * NG;
  using Rec = StepRec<NTC>;
  const int lane = threadIdx.x;
  const int b0 = blockIdx.x * 64 + lane;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));

  double vv[NG][16][NC], gv[NG][16][NC];
  fc_load_v<P, NG>(a, B, b, vv);
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int c = 0; c < NC; ++c) gv[g][k][c] = 0.0;
  auto u_of = [&](const double (&row)[NG], double (&u)[NC]) OCS_INLINE {
#pragma unroll
    for (int c = 0; c < NC; ++c) u[c] = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) FcRow16<NC>::dot(row[g], vv[g], u);
  };
  auto fold = [&](const double (&row)[NG], const double (&d)[NC]) OCS_INLINE {  // dJdv += dJdu(:,j) B(:,j)'
#pragma unroll
    for (int g = 0; g < NG; ++g) FcRow16<NC>::axpy(row[g], d, gv[g]);
  };

  double lam[NS];  // lam(:,end) = e_last   :63-69; lam(end,:) stays 1
#pragma unroll
  for (int k = 0; k < NS; ++k) lam[k] = 0.0;
  const double lamc = 1.0;

  const double* bp = a.BT + (size_t)(2 * N + 1) * LD + (lane & 15);  // one column past the last
  auto load_row = [&](double (&row)[NG]) OCS_INLINE {                // walks the columns downwards
    bp -= LD;
#pragma unroll
    for (int g = 0; g < NG; ++g) row[g] = bp[16 * g];
  };
  const double* xp = a.ck + b + ((size_t)N * NAUG) * B;  // x(1,N+1)
  double rnext[NG], unext[NC], pend[NC];                  // column 2i+2 of B, u(:,2i+2), k1-term of step i+1
  load_row(rnext);
  u_of(rnext, unext);
#pragma unroll
  for (int c = 0; c < NC; ++c) pend[c] = 0.0;

  auto step = [&](const Rec& r, const double* xi, const double (&rA)[NG], const double (&rM)[NG]) OCS_INLINE {
    double uA[NC], uM[NC];
    u_of(rA, uA);
    u_of(rM, uM);
    const double* uB = unext;
    double f[NS], Y2[NS], Y3[NS], Y4[NS];
    P::Fx(r.tcA, xi, uA, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y2[k] = __builtin_fma(r.hh, f[k], xi[k]);
    P::Fx(r.tcM, Y2, uM, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y3[k] = __builtin_fma(r.hh, f[k], xi[k]);
    P::Fx(r.tcM, Y3, uM, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y4[k] = __builtin_fma(r.h, f[k], xi[k]);
    double k4[NAUG], k3[NAUG], k2[NAUG], k1[NAUG], g3[NS], g2[NS], g1[NS], g0[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) k4[k] = r.h6 * lam[k];                             // :73
    k4[NS] = r.h6 * lamc;
    P::dFdxT(r.tcB, Y4, uB, p, k4, g3);                                             // :74-75
#pragma unroll
    for (int k = 0; k < NS; ++k) k3[k] = __builtin_fma(r.h, g3[k], r.h3 * lam[k]);  // :77
    k3[NS] = r.h3 * lamc;
    P::dFdxT(r.tcM, Y3, uM, p, k3, g2);                                             // :78-79
#pragma unroll
    for (int k = 0; k < NS; ++k) k2[k] = __builtin_fma(r.hh, g2[k], r.h3 * lam[k]); // :81
    k2[NS] = r.h3 * lamc;
    P::dFdxT(r.tcM, Y2, uM, p, k2, g1);                                             // :82-83
#pragma unroll
    for (int k = 0; k < NS; ++k) k1[k] = __builtin_fma(r.hh, g1[k], r.h6 * lam[k]); // :85
    k1[NS] = r.h6 * lamc;
    P::dFdxT(r.tcA, xi, uA, p, k1, g0);                                             // :87-88
    // compute_dJdu :97-121: column 2i+2 pairs k4 of step i with k1 of step i+1; both columns are folded
    // into dJdv at once
    double d4[NC], d3[NC], d2[NC], dn[NC], dm[NC];
    P::dFduT(r.tcB, Y4, uB, p, k4, d4);
    P::dFduT(r.tcM, Y3, uM, p, k3, d3);
    P::dFduT(r.tcM, Y2, uM, p, k2, d2);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      dn[c] = pend[c] + d4[c];  // column 2i+2  :112-116 (:119-120 at i = N-1)
      dm[c] = d2[c] + d3[c];    // column 2i+1  :105-109
    }
    fold(rnext, dn);
    fold(rM, dm);
    P::dFduT(r.tcA, xi, uA, p, k1, pend);
#pragma unroll
    for (int k = 0; k < NS; ++k) lam[k] = (((lam[k] + g1[k]) + g2[k]) + g3[k]) + g0[k];  // :86-88
#pragma unroll
    for (int c = 0; c < NC; ++c) unext[c] = uA[c];
#pragma unroll
    for (int g = 0; g < NG; ++g) rnext[g] = rA[g];
  };

  static_assert(PF <= kRecPad, "record ring deeper than the table padding");
  Rec rq[PF];
  const double* recp = a.REC + (size_t)(N - 1) * rec_stride(NTC);
#pragma unroll
  for (int q = 0; q < PF; ++q) {
    rq[q] = load_rec<NTC>(recp);
    recp -= rec_stride(NTC);
  }
  auto next_rec = [&]() OCS_INLINE {
    const Rec cur = rq[0];
#pragma unroll
    for (int q = 0; q + 1 < PF; ++q) rq[q] = rq[q + 1];
    rq[PF - 1] = load_rec<NTC>(recp);
    recp -= rec_stride(NTC);
    return cur;
  };
  auto load_x = [&](double (&xi)[NS]) OCS_INLINE {
    xp -= (size_t)NAUG * B;
#pragma unroll
    for (int k = 0; k < NS; ++k) xi[k] = xp[(size_t)k * B];
  };

  const int nch = N / CH;
  for (int i = N - 1; i >= nch * CH; --i) {  // remainder steps at the top of the grid first
    double xi[NS], rM[NG], rA[NG];
    load_x(xi);
    load_row(rM);
    load_row(rA);
    const Rec cur = next_rec();
    step(cur, xi, rA, rM);
  }
  double xb0[CH][NS], xb1[CH][NS], rb0[2 * CH][NG], rb1[2 * CH][NG];
  auto load_chunk = [&](double (&xd)[CH][NS], double (&rd)[2 * CH][NG]) OCS_INLINE {
#pragma unroll
    for (int s = CH - 1; s >= 0; --s) load_x(xd[s]);
#pragma unroll
    for (int s = 2 * CH - 1; s >= 0; --s) load_row(rd[s]);
  };
  auto run_chunk = [&](const double (&xs)[CH][NS], const double (&rs)[2 * CH][NG]) OCS_INLINE {
#pragma unroll
    for (int s = CH - 1; s >= 0; --s) {
      const Rec cur = next_rec();
      step(cur, xs[s], rs[2 * s], rs[2 * s + 1]);
    }
  };
  int c = nch - 1;
  if (c >= 0) load_chunk(xb0, rb0);
  for (; c >= 1; c -= 2) {
    load_chunk(xb1, rb1);
    run_chunk(xb0, rb0);
    if (c >= 2) load_chunk(xb0, rb0);
    run_chunk(xb1, rb1);
  }
  if (c == 0) run_chunk(xb0, rb0);

  fold(rnext, pend);  // left end point :101-102 (rnext now holds column 1 of B)
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        const int kk = 16 * g + k;
        if (kk < a.nBasis) a.dJdv[((size_t)kk * NC + cc) * B + b] = gv[g][k][cc];
      }
  if (a.lam0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) a.lam0[(size_t)k * B + b] = lam[k];
    a.lam0[(size_t)NS * B + b] = lamc;
  }
  if (warm == 1.234567e300) a.dJdv[b] = warm;
}

// ---------------------------------------------------------------------------------------
// Two-role variants for full-chip batches (> 32 768 trajectories, where the wave-specialised tiles of
// ocs_fused_wave_kernels.hip are no longer selected): the same lane mapping, but the products with the basis move
// to a SECOND wave of the workgroup.  One lane kernel per 64 trajectories leaves one wave per SIMD at batch 65 536,
// and a lone wave is bound by its dependent chain (RK4 stages, the 16-deep v_fmac_f64_dpp accumulation) at about a
// third of the fp64 issue rate (scripts/probe/fp64_rate.hip).  Split in two, the integrator wave and the basis
// wave run side by side on the SIMDs; they meet at one LDS barrier per block of 8 steps:
//   forward   wave 1 evaluates the 16 samples u(:, 16 j + 1 .. 16 j + 16) of block j + 1 into LDS while wave 0
//             integrates block j from the samples written one barrier earlier;
//   backward  wave 1 evaluates the samples of the block below AND folds the 16 finished columns of dJdu of the
//             block above into dJdv, wave 0 runs the adjoint steps of the block between them.
// Arithmetic per lane is unchanged (same instructions, same order within each sum), so J, dJdv and lam0 are
// bit-identical to k_forward_fc / k_backward_fc.  Requires N % 8 == 0 and nBasis <= 16.
// ---------------------------------------------------------------------------------------
constexpr int kFc2Steps = 8;  // steps per block
__device__ static inline void fc2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class P>
__global__ __launch_bounds__(128) void k_forward_fc2(const FcArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC, NAUG = P::NAUG, LD = 16, BS = kFc2Steps, NR = 2 * BS;
  using Rec = StepRec<NTC>;
  __shared__ double ub[2][NR][NC][64];  // u(:, 16 j + 1 + s), block j in slot j & 1
  __shared__ double u0s[NC][64];        // u(:, 0)
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int b0 = blockIdx.x * 64 + lane;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N, nblk = N / BS;

  if (role == 1) {  // ---- basis wave: u = v B   ChebyshevControl.m:35-39
    double vv[1][16][NC];
    fc_load_v<P, 1>(a, B, b, vv);
    const double* bp = a.BT + (lane & 15);
    auto load_rows = [&](double (&dst)[NR]) OCS_INLINE {
#pragma unroll
      for (int s = 0; s < NR; ++s) dst[s] = bp[s * LD];
      bp += NR * LD;
    };
    auto expand = [&](const double (&src)[NR], const int slot) OCS_INLINE {
#pragma unroll
      for (int s = 0; s < NR; ++s) {
        double u[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) u[c] = 0.0;
        FcRow16<NC>::dot(src[s], vv[0], u);
#pragma unroll
        for (int c = 0; c < NC; ++c) ub[slot][s][c][lane] = u[c];
      }
    };
    {
      const double r0 = bp[0];
      bp += LD;
      double u[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) u[c] = 0.0;
      FcRow16<NC>::dot(r0, vv[0], u);
#pragma unroll
      for (int c = 0; c < NC; ++c) u0s[c][lane] = u[c];
    }
    double rb0[NR], rb1[NR];
    load_rows(rb0);
    if (nblk > 1) load_rows(rb1);
    expand(rb0, 0);
    fc2_barrier();
    for (int j = 0; j < nblk; j += 2) {
      if (j + 2 < nblk) load_rows(rb0);
      if (j + 1 < nblk) expand(rb1, 1);
      fc2_barrier();
      if (j + 1 < nblk) {
        if (j + 3 < nblk) load_rows(rb1);
        if (j + 2 < nblk) expand(rb0, 0);
        fc2_barrier();
      }
    }
    return;
  }

  // ---- integrator wave   RK4Integrator.m:28-56
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));
  double y[NS], yc = 0.0;
#pragma unroll
  for (int k = 0; k < NS; ++k) y[k] = a.x0[(size_t)k * B + b];
  double* xo = a.ck + b;
#pragma unroll
  for (int k = 0; k < NS; ++k) xo[(size_t)k * B] = y[k];
  xo += (size_t)NAUG * B;
  auto step = [&](const Rec& r, const double* uA, const double* uM, const double* uB) OCS_INLINE {
    double F1[NS + 1], F2[NS + 1], F3[NS + 1], F4[NS + 1], Y[NS];
    P::F(r.tcA, y, uA, p, F1);                                             // :39
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.hh, F1[k], y[k]);  // :40
    P::F(r.tcM, Y, uM, p, F2);                                             // :42
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.hh, F2[k], y[k]);  // :43
    P::F(r.tcM, Y, uM, p, F3);                                             // :45
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r.h, F3[k], y[k]);   // :46
    P::F(r.tcB, Y, uB, p, F4);                                             // :48
#pragma unroll
    for (int k = 0; k < NS; ++k)                                           // :50-51
      y[k] = __builtin_fma(r.h6, __builtin_fma(2.0, F3[k], __builtin_fma(2.0, F2[k], F1[k])) + F4[k], y[k]);
    yc = __builtin_fma(r.h6, __builtin_fma(2.0, F3[NS], __builtin_fma(2.0, F2[NS], F1[NS])) + F4[NS], yc);
#pragma unroll
    for (int k = 0; k < NS; ++k) xo[(size_t)k * B] = y[k];
    xo += (size_t)NAUG * B;
  };
  constexpr int PF = 4;
  static_assert(PF <= kRecPad, "record ring deeper than the table padding");
  Rec rq[PF];
  const double* recp = a.REC;
#pragma unroll
  for (int q = 0; q < PF; ++q) {
    rq[q] = load_rec<NTC>(recp);
    recp += rec_stride(NTC);
  }
  auto next_rec = [&]() OCS_INLINE {
    const Rec cur = rq[0];
#pragma unroll
    for (int q = 0; q + 1 < PF; ++q) rq[q] = rq[q + 1];
    rq[PF - 1] = load_rec<NTC>(recp);
    recp += rec_stride(NTC);
    return cur;
  };
  fc2_barrier();
  double uprev[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) uprev[c] = u0s[c][lane];
  for (int j = 0; j < nblk; ++j) {
    const int slot = j & 1;
    double us[NR][NC];  // all samples of the block at once: one LDS wait per block, the record ring stays ahead
#pragma unroll
    for (int s = 0; s < NR; ++s)
#pragma unroll
      for (int c = 0; c < NC; ++c) us[s][c] = ub[slot][s][c][lane];
#pragma unroll
    for (int s = 0; s < BS; ++s) {
      const Rec cur = next_rec();
      step(cur, uprev, us[2 * s], us[2 * s + 1]);
#pragma unroll
      for (int c = 0; c < NC; ++c) uprev[c] = us[2 * s + 1][c];
    }
    fc2_barrier();
  }
  if (b0 < a.batch) a.J[b] = yc;  // J = x(end,end)   :55
  if (warm == 1.234567e300) a.J[b] = warm;
}

template <class P>
__global__ __launch_bounds__(128) void k_backward_fc2(const FcArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC, NAUG = P::NAUG, LD = 16, BS = kFc2Steps, NR = 2 * BS;
  using Rec = StepRec<NTC>;
  __shared__ double ub[2][NR][NC][64];  // u(:, 16 blk + s)
  __shared__ double cs[2][NR][NC][64];  // dJdu(:, 16 blk + 1 + s)
  __shared__ double ends[2][NC][64];    // [0] u(:, 2N)   [1] dJdu(:, 0)
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int b0 = blockIdx.x * 64 + lane;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N, nblk = N / BS;
  // iteration t = 0 .. nblk-1 works on block blk = nblk-1-t (the grid is walked downwards), slot t & 1

  if (role == 1) {  // ---- basis wave: u = v B and dJdv = dJdu B'   ChebyshevControl.m:35-43
    double vv[1][16][NC], gv[1][16][NC];
    fc_load_v<P, 1>(a, B, b, vv);
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int c = 0; c < NC; ++c) gv[0][k][c] = 0.0;
    const double* bt = a.BT + (lane & 15);
    auto load_rows = [&](double (&dst)[NR], const int t, const int shift) OCS_INLINE {
      const double* q = bt + (size_t)(16 * (nblk - 1 - t) + shift) * LD;
#pragma unroll
      for (int s = 0; s < NR; ++s) dst[s] = q[s * LD];
    };
    auto expand = [&](const double (&src)[NR], const int slot) OCS_INLINE {
#pragma unroll
      for (int s = 0; s < NR; ++s) {
        double u[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) u[c] = 0.0;
        FcRow16<NC>::dot(src[s], vv[0], u);
#pragma unroll
        for (int c = 0; c < NC; ++c) ub[slot][s][c][lane] = u[c];
      }
    };
    auto contract = [&](const double (&src)[NR], const int slot) OCS_INLINE {  // the order of k_backward_fc: columns downwards
#pragma unroll
      for (int s = NR - 1; s >= 0; --s) {
        double d[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) d[c] = cs[slot][s][c][lane];
        FcRow16<NC>::axpy(src[s], d, gv[0]);
      }
    };
    {
      const double rt = bt[(size_t)(2 * N) * LD];
      double u[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) u[c] = 0.0;
      FcRow16<NC>::dot(rt, vv[0], u);
#pragma unroll
      for (int c = 0; c < NC; ++c) ends[0][c][lane] = u[c];
    }
    double E0[NR], E1[NR], C0[NR], C1[NR];
    load_rows(E0, 0, 0);
    if (nblk > 1) load_rows(E1, 1, 0);
    expand(E0, 0);
    fc2_barrier();
    for (int t = 0; t < nblk; t += 2) {
      if (t + 2 < nblk) load_rows(E0, t + 2, 0);
      load_rows(C0, t, 1);
      if (t + 1 < nblk) expand(E1, 1);
      if (t >= 1) contract(C1, 1);
      fc2_barrier();
      if (t + 1 < nblk) {
        if (t + 3 < nblk) load_rows(E1, t + 3, 0);
        load_rows(C1, t + 1, 1);
        if (t + 2 < nblk) expand(E0, 0);
        contract(C0, 0);
        fc2_barrier();
      }
    }
    if ((nblk - 1) & 1) contract(C1, 1); else contract(C0, 0);
    {
      const double r0 = bt[0];
      double d[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) d[c] = ends[1][c][lane];
      FcRow16<NC>::axpy(r0, d, gv[0]);  // left end point :101-102
    }
    if (b0 < a.batch) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c)
          if (k < a.nBasis) a.dJdv[((size_t)k * NC + c) * B + b] = gv[0][k][c];
    }
    return;
  }

  // ---- adjoint wave   RK4Integrator.m:59-121
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));
  double lam[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) lam[k] = 0.0;
  const double lamc = 1.0;
  const double* xp = a.ck + b + ((size_t)N * NAUG) * B;
  double unext[NC], pend[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) pend[c] = 0.0;

  auto step = [&](const Rec& r, const double* xi, const double* uA, const double* uM, double (&dn)[NC],
                  double (&dm)[NC]) OCS_INLINE {
    const double* uB = unext;
    double f[NS], Y2[NS], Y3[NS], Y4[NS];
    P::Fx(r.tcA, xi, uA, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y2[k] = __builtin_fma(r.hh, f[k], xi[k]);
    P::Fx(r.tcM, Y2, uM, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y3[k] = __builtin_fma(r.hh, f[k], xi[k]);
    P::Fx(r.tcM, Y3, uM, p, f);
#pragma unroll
    for (int k = 0; k < NS; ++k) Y4[k] = __builtin_fma(r.h, f[k], xi[k]);
    double k4[NAUG], k3[NAUG], k2[NAUG], k1[NAUG], g3[NS], g2[NS], g1[NS], g0[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) k4[k] = r.h6 * lam[k];                             // :73
    k4[NS] = r.h6 * lamc;
    P::dFdxT(r.tcB, Y4, uB, p, k4, g3);                                             // :74-75
#pragma unroll
    for (int k = 0; k < NS; ++k) k3[k] = __builtin_fma(r.h, g3[k], r.h3 * lam[k]);  // :77
    k3[NS] = r.h3 * lamc;
    P::dFdxT(r.tcM, Y3, uM, p, k3, g2);                                             // :78-79
#pragma unroll
    for (int k = 0; k < NS; ++k) k2[k] = __builtin_fma(r.hh, g2[k], r.h3 * lam[k]); // :81
    k2[NS] = r.h3 * lamc;
    P::dFdxT(r.tcM, Y2, uM, p, k2, g1);                                             // :82-83
#pragma unroll
    for (int k = 0; k < NS; ++k) k1[k] = __builtin_fma(r.hh, g1[k], r.h6 * lam[k]); // :85
    k1[NS] = r.h6 * lamc;
    P::dFdxT(r.tcA, xi, uA, p, k1, g0);                                             // :87-88
    double d4[NC], d3[NC], d2[NC];
    P::dFduT(r.tcB, Y4, uB, p, k4, d4);
    P::dFduT(r.tcM, Y3, uM, p, k3, d3);
    P::dFduT(r.tcM, Y2, uM, p, k2, d2);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      dn[c] = pend[c] + d4[c];  // column 2i+2  :112-116 (:119-120 at i = N-1)
      dm[c] = d2[c] + d3[c];    // column 2i+1  :105-109
    }
    P::dFduT(r.tcA, xi, uA, p, k1, pend);
#pragma unroll
    for (int k = 0; k < NS; ++k) lam[k] = (((lam[k] + g1[k]) + g2[k]) + g3[k]) + g0[k];  // :86-88
#pragma unroll
    for (int c = 0; c < NC; ++c) unext[c] = uA[c];
  };

  constexpr int PF = 4;
  static_assert(PF <= kRecPad, "record ring deeper than the table padding");
  Rec rq[PF];
  const double* recp = a.REC + (size_t)(N - 1) * rec_stride(NTC);
#pragma unroll
  for (int q = 0; q < PF; ++q) {
    rq[q] = load_rec<NTC>(recp);
    recp -= rec_stride(NTC);
  }
  auto next_rec = [&]() OCS_INLINE {
    const Rec cur = rq[0];
#pragma unroll
    for (int q = 0; q + 1 < PF; ++q) rq[q] = rq[q + 1];
    rq[PF - 1] = load_rec<NTC>(recp);
    recp -= rec_stride(NTC);
    return cur;
  };
  auto load_xs = [&](double (&xd)[BS][NS]) OCS_INLINE {  // the checkpoints of one block, top first
#pragma unroll
    for (int s = BS - 1; s >= 0; --s) {
      xp -= (size_t)NAUG * B;
#pragma unroll
      for (int k = 0; k < NS; ++k) xd[s][k] = xp[(size_t)k * B];
    }
  };
  auto run_block = [&](const double (&xs)[BS][NS], const int slot, const bool last) OCS_INLINE {
    double us[NR][NC];
#pragma unroll
    for (int s = 0; s < NR; ++s)
#pragma unroll
      for (int c = 0; c < NC; ++c) us[s][c] = ub[slot][s][c][lane];
#pragma unroll
    for (int s = BS - 1; s >= 0; --s) {
      const Rec cur = next_rec();
      double dn[NC], dm[NC];
      step(cur, xs[s], us[2 * s], us[2 * s + 1], dn, dm);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        cs[slot][2 * s + 1][c][lane] = dn[c];
        cs[slot][2 * s][c][lane] = dm[c];
      }
    }
    if (last) {
#pragma unroll
      for (int c = 0; c < NC; ++c) ends[1][c][lane] = pend[c];
    }
    fc2_barrier();
  };
  double xb0[BS][NS], xb1[BS][NS];
  load_xs(xb0);
  fc2_barrier();
#pragma unroll
  for (int c = 0; c < NC; ++c) unext[c] = ends[0][c][lane];
  for (int t = 0; t < nblk; t += 2) {
    if (t + 1 < nblk) load_xs(xb1);
    run_block(xb0, 0, t + 1 == nblk);
    if (t + 1 < nblk) {
      if (t + 2 < nblk) load_xs(xb0);
      run_block(xb1, 1, t + 2 == nblk);
    }
  }
  if (a.lam0 && b0 < a.batch) {
#pragma unroll
    for (int k = 0; k < NS; ++k) a.lam0[(size_t)k * B + b] = lam[k];
    a.lam0[(size_t)NS * B + b] = lamc;
  }
  if (warm == 1.234567e300) a.dJdv[b] = warm;
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
bool fused_control_supported(Functor f, int nS, int nC, int nBasis) {
  return f == Functor::Logistic && nS >= 1 && nS <= 4 && nC == 1 && nBasis >= 1 && nBasis <= 32;
}

constexpr int kFcChunk = 4;
// the two-role kernels: below 32 768 trajectories (SIMDs with no wave of their own left), nS <= 2 (register budget of
// the 8-step blocks); OCS_FC2=0 switches them off, OCS_FC2=2 selects them at every batch (A/B timing).
// Measured, fp64 Chebyshev-16, N = 1000, us per objective + gradient, one wave -> two roles:
//   batch 100: 454 -> 313    8192: 476 -> 326    32 768: 527 -> 497    49 152: 553 -> 577    65 536: 660 -> 661
// At the full chip the second wave per SIMD buys nothing: the 185 fp64 instructions per (trajectory, step) of the
// pair already issue at the rate the fp64 pipe sustains with every SIMD busy (37 of the ~45 TFLOP/s that
// scripts/probe/fp64_rate.hip reaches with independent FMAs), so the lane kernels there are issue-bound, not
// latency-bound.
static bool fc2_enabled(int batch) {
  static const int mode = [] {
    const char* e = getenv("OCS_FC2");
    return e ? atoi(e) : 1;
  }();
  return mode == 2 || (mode == 1 && batch < 32768);
}
template <class P, int NG>
static void run_fc(bool forward, const FcArgs& a, hipStream_t s) {
  constexpr int PF = P::NS <= 2 ? 4 : 3;
  const dim3 grid((a.batch + 63) / 64), block(64);
  if constexpr (NG == 1 && P::NS <= 2) {
    if (fc2_enabled(a.batch) && a.N % kFc2Steps == 0 && a.N >= kFc2Steps) {
      if (forward)
        k_forward_fc2<P><<<grid, dim3(128), 0, s>>>(a);
      else
        k_backward_fc2<P><<<grid, dim3(128), 0, s>>>(a);
      return;
    }
  }
  if (forward)
    k_forward_fc<P, kFcChunk, PF, NG><<<grid, block, 0, s>>>(a);
  else
    k_backward_fc<P, kFcChunk, PF, NG><<<grid, block, 0, s>>>(a);
}
static int launch_fc(bool forward, const ProblemDesc& p, const FcArgs& a, hipStream_t s) {
  if (!fused_control_supported(p.functor, p.nS, p.nC, a.nBasis)) return -1;
  const bool two = a.nBasis > 16;
  switch (p.nS) {
    case 1: two ? run_fc<LogisticK<1>, 2>(forward, a, s) : run_fc<LogisticK<1>, 1>(forward, a, s); break;
    case 2: two ? run_fc<LogisticK<2>, 2>(forward, a, s) : run_fc<LogisticK<2>, 1>(forward, a, s); break;
    case 3: two ? run_fc<LogisticK<3>, 2>(forward, a, s) : run_fc<LogisticK<3>, 1>(forward, a, s); break;
    case 4: two ? run_fc<LogisticK<4>, 2>(forward, a, s) : run_fc<LogisticK<4>, 1>(forward, a, s); break;
    default: return -1;
  }
  return hip_rc_fc(hipGetLastError());
}
// BT16: transposed basis [2N+1][16 or 32] zero-padded (16 when nBasis <= 16)
int launch_forward_fc(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, const double* BT16,
                      const double* v, const double* x0, double* ck, double* J, hipStream_t s) {
  FcArgs a{};
  a.N = g.N; a.batch = batch; a.nBasis = nBasis; a.REC = g.REC; a.ps = p.ps; a.pb = p.pb; a.pmask = p.pmask;
  a.BT = BT16; a.v = v; a.x0 = x0; a.ck = ck; a.J = J;
  return launch_fc(true, p, a, s);
}
int launch_backward_fc(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, const double* BT16,
                       const double* v, const double* ck, double* dJdv, double* lam0, hipStream_t s) {
  FcArgs a{};
  a.N = g.N; a.batch = batch; a.nBasis = nBasis; a.REC = g.REC; a.ps = p.ps; a.pb = p.pb; a.pmask = p.pmask;
  a.BT = BT16; a.v = v; a.ck = const_cast<double*>(ck); a.dJdv = dJdv; a.lam0 = lam0;
  return launch_fc(false, p, a, s);
}

}  // namespace ocs
