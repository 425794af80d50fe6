// ocs_fused_banded_kernels.hip -- the shooting objective and its gradient with a BANDED control basis inside the
// RK4 kernels: the reference's default parametrisations
//
//   Control/PWLinearControl.m:31-62     B(i,j) = max(0, 1 - |t_j - c_i| / dc): at most two consecutive non-zeros per column
//   Control/PWConstantControl.m:30-50   B(i,j) = [s_i <= t_j < s_{i+1}]: exactly one
//   functions/single_shooting.m:137-150 u = compute_u(v); J; dJdu; dJdv = compute_dJdv(dJdu)
//
// Like the dense case (ocs_fused_control_kernels.hip) neither u nor dJdu reaches memory.  Column j of B is the pair
// (w0_j, w1_j) on rows (r_j, r_j + 1) with r_j non-decreasing in steps of at most one (checked on the host, else the
// unfused kernels are used).  A lane (= trajectory) keeps the coefficient rows r, r+1 of v in registers and the row
// after them on its way; u(:,j) = w0 v_r + w1 v_{r+1}.  The adjoint pass walks the columns downwards with two
// cursors: one for the controls it has to rebuild (columns 2i, 2i+1 at the start of step i) and one for the columns of
// dJdu it finishes (2i+2, 2i+1 at the end of step i), folding them into the two live sums dJdv_r, dJdv_{r+1}; a sum is
// stored when its row leaves the band.  The column table is wave-uniform and read through scalar loads one step ahead
// (4 doubles per column: the 100-SGPR problem of the dense case does not arise).
#include "ocs_device_common.hpp"
#include "ocs_internal.hpp"
#include "ocs_problems.hpp"

namespace ocs {

static inline int hip_rc_fb(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// column table entry: {w0, w1, adv, pad}; adv = r_j - r_{j-1} (0 or 1; r_0 in the header of the args)
constexpr int kBandRec = 4;

struct FbArgs {
  int N, batch, nBasis, r0;  // r0: first row of column 0
  const double* REC;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* CT;   // [2N+1][kBandRec] column table
  const double* v;    // [nBasis][nC][B]
  const double* x0;
  double* ck;         // checkpoints [N+1][nAug][B] (state rows only are touched)
  double* J;
  double* dJdv;       // [nBasis][nC][B], zero-filled by the caller (rows outside every band stay 0)
  double* lam0;
};

struct BandCol {
  double w0, w1;
  int adv;
};
__device__ static inline BandCol band_col(uniform_ptr ct, int j) {
  BandCol c;
  c.w0 = ct[(size_t)j * kBandRec + 0];
  c.w1 = ct[(size_t)j * kBandRec + 1];
  c.adv = (int)ct[(size_t)j * kBandRec + 2];
  return c;
}

// ---------------------------------------------------------------------------------------
// forward: J = x(end,end) of compute_states(u = v*B)   RK4Integrator.m:28-56
// ---------------------------------------------------------------------------------------
template <class P, int PF>
__global__ __launch_bounds__(64) void k_forward_fb(const FbArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC, NAUG = P::NAUG;
  using Rec = StepRec<NTC>;
  const int b0 = blockIdx.x * 64 + threadIdx.x;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N, nB = a.nBasis;
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));
  const uniform_ptr CT = as_uniform(a.CT);

  auto load_row = [&](int r, double (&o)[NC]) OCS_INLINE {
    r = r < 0 ? 0 : (r > nB - 1 ? nB - 1 : r);
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = a.v[((size_t)r * NC + c) * B + b];
  };
  int r = a.r0;
  double va[NC], vb[NC], vn[NC];  // rows r, r+1 and (on its way) r+2
  load_row(r, va);
  load_row(r + 1, vb);
  load_row(r + 2, vn);
  auto control = [&](const BandCol& c, double (&u)[NC]) OCS_INLINE {
    if (c.adv) {  // uniform: the band moves up one row
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        va[k] = vb[k];
        vb[k] = vn[k];
      }
      ++r;
      load_row(r + 2, vn);
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) u[k] = __builtin_fma(c.w1, vb[k], c.w0 * va[k]);
  };

  double y[NS], yc = 0.0;
#pragma unroll
  for (int k = 0; k < NS; ++k) y[k] = a.x0[(size_t)k * B + b];
  double* xo = a.ck + b;
#pragma unroll
  for (int k = 0; k < NS; ++k) xo[(size_t)k * B] = y[k];
  xo += (size_t)NAUG * B;

  double uprev[NC];
  {
    BandCol c0 = band_col(CT, 0);
    c0.adv = 0;
    control(c0, uprev);
  }
  static_assert(PF <= kRecPad, "record ring deeper than the table padding");
  Rec rq[PF];
  const double* recp = a.REC;
#pragma unroll
  for (int q = 0; q < PF; ++q) {
    rq[q] = load_rec<NTC>(recp);
    recp += rec_stride(NTC);
  }
  BandCol cM = band_col(CT, 1), cB = band_col(CT, 2);
  for (int i = 0; i < N; ++i) {
    // the next step's two columns are requested now (scalar loads) and used an iteration later
    const int jn = 2 * i + 3 <= 2 * N - 1 ? 2 * i + 3 : 2 * N - 1;
    const BandCol nM = band_col(CT, jn), nB2 = band_col(CT, jn + 1);
    const Rec r_ = rq[0];
#pragma unroll
    for (int q = 0; q + 1 < PF; ++q) rq[q] = rq[q + 1];
    rq[PF - 1] = load_rec<NTC>(recp);
    recp += rec_stride(NTC);
    double uM[NC], uB[NC];
    control(cM, uM);
    control(cB, uB);
    double F1[NS + 1], F2[NS + 1], F3[NS + 1], F4[NS + 1], Y[NS];
    P::F(r_.tcA, y, uprev, p, F1);                                          // :39
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r_.hh, F1[k], y[k]);  // :40
    P::F(r_.tcM, Y, uM, p, F2);                                             // :42
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r_.hh, F2[k], y[k]);  // :43
    P::F(r_.tcM, Y, uM, p, F3);                                             // :45
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[k] = __builtin_fma(r_.h, F3[k], y[k]);   // :46
    P::F(r_.tcB, Y, uB, p, F4);                                             // :48
#pragma unroll
    for (int k = 0; k < NS; ++k)                                            // :50-51
      y[k] = __builtin_fma(r_.h6, __builtin_fma(2.0, F3[k], __builtin_fma(2.0, F2[k], F1[k])) + F4[k], y[k]);
    yc = __builtin_fma(r_.h6, __builtin_fma(2.0, F3[NS], __builtin_fma(2.0, F2[NS], F1[NS])) + F4[NS], yc);
#pragma unroll
    for (int k = 0; k < NS; ++k) xo[(size_t)k * B] = y[k];
    xo += (size_t)NAUG * B;
#pragma unroll
    for (int k = 0; k < NC; ++k) uprev[k] = uB[k];
    cM = nM;
    cB = nB2;
  }
  a.J[b] = yc;  // J = x(end,end)   :55
  if (warm == 1.234567e300) a.J[b] = warm;
}

// ---------------------------------------------------------------------------------------
// adjoint: dJdv = compute_dJdv(compute_adjoints(u = v*B))   RK4Integrator.m:59-121
// ---------------------------------------------------------------------------------------
template <class P, int PF>
__global__ __launch_bounds__(64) void k_backward_fb(const FbArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC, NAUG = P::NAUG;
  using Rec = StepRec<NTC>;
  const int b0 = blockIdx.x * 64 + threadIdx.x;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N, nB = a.nBasis;
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));
  const uniform_ptr CT = as_uniform(a.CT);

  auto load_row = [&](int r, double (&o)[NC]) OCS_INLINE {
    r = r < 0 ? 0 : (r > nB - 1 ? nB - 1 : r);
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = a.v[((size_t)r * NC + c) * B + b];
  };
  // first row of the last column: r0 + sum of adv (the host passes it in the pad slot of the last entry)
  const int rlast = (int)CT[(size_t)(2 * N) * kBandRec + 3];

  // cursor U (controls): starts on column 2N, walks down
  int ru = rlast;
  double va[NC], vb[NC], vp[NC];  // rows ru, ru+1 and (on its way) ru-1
  load_row(ru, va);
  load_row(ru + 1, vb);
  load_row(ru - 1, vp);
  // moving from column j to j-1: the band drops one row if adv_j == 1
  auto control_down = [&](int adv_above, double w0, double w1, double (&u)[NC]) OCS_INLINE {
    if (adv_above) {
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        vb[k] = va[k];
        va[k] = vp[k];
      }
      --ru;
      load_row(ru - 1, vp);
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) u[k] = __builtin_fma(w1, vb[k], w0 * va[k]);
  };
  // cursor G (gradient sums): rows rg, rg+1 live in ga, gb
  int rg = rlast;
  double ga[NC], gb[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) ga[k] = gb[k] = 0.0;
  auto flush_row = [&](int row, const double (&g)[NC]) OCS_INLINE {
    if (row >= 0 && row < nB) {
#pragma unroll
      for (int k = 0; k < NC; ++k) a.dJdv[((size_t)row * NC + k) * B + b] = g[k];
    }
  };
  // fold column j (value d) into the sums; `adv_above` = adv of column j+1 (the band dropped on the way down to j)
  auto fold = [&](int adv_above, double w0, double w1, const double (&d)[NC]) OCS_INLINE {
    if (adv_above) {
      flush_row(rg + 1, gb);
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        gb[k] = ga[k];
        ga[k] = 0.0;
      }
      --rg;
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      ga[k] = __builtin_fma(w0, d[k], ga[k]);
      gb[k] = __builtin_fma(w1, d[k], gb[k]);
    }
  };

  double lam[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) lam[k] = 0.0;
  const double lamc = 1.0;
  double unext[NC], pend[NC];
  BandCol cTop = band_col(CT, 2 * N);  // column 2i+2 of the step being processed
  control_down(0, cTop.w0, cTop.w1, unext);
#pragma unroll
  for (int c = 0; c < NC; ++c) pend[c] = 0.0;
  int adv_g = 0;  // adv of the column above the one cursor G folds next (0 for the very first column, 2N)

  static_assert(PF <= kRecPad, "record ring deeper than the table padding");
  Rec rq[PF];
  const double* recp = a.REC + (size_t)(N - 1) * rec_stride(NTC);
#pragma unroll
  for (int q = 0; q < PF; ++q) {
    rq[q] = load_rec<NTC>(recp);
    recp -= rec_stride(NTC);
  }
  BandCol cM = band_col(CT, 2 * N - 1), cA = band_col(CT, 2 * N - 2);
  // checkpoints are requested four steps ahead into a ring of four slots; the loop is unrolled by the ring so that
  // no freshly requested value is copied (a copy would be a wait)
  constexpr int RD = 4;
  auto load_x = [&](int i, double (&o)[NS]) OCS_INLINE {
    const double* q = a.ck + b + ((size_t)(i < 0 ? 0 : i) * NAUG) * B;
#pragma unroll
    for (int k = 0; k < NS; ++k) o[k] = q[(size_t)k * B];
  };
  double xr[RD][NS];
#pragma unroll
  for (int q = 0; q < RD; ++q) load_x(N - 1 - q, xr[q]);
  auto step = [&](int i, double (&xslot)[NS]) OCS_INLINE {
    // next step's two columns (scalar loads), used an iteration later
    const int jn = 2 * i - 1 >= 1 ? 2 * i - 1 : 1;
    const BandCol nM = band_col(CT, jn), nA = band_col(CT, jn - 1);
    double xi[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) xi[k] = xslot[k];
    load_x(i - RD, xslot);
    const Rec r = rq[0];
#pragma unroll
    for (int q = 0; q + 1 < PF; ++q) rq[q] = rq[q + 1];
    rq[PF - 1] = load_rec<NTC>(recp);
    recp -= rec_stride(NTC);

    double uA[NC], uM[NC];
    control_down(cTop.adv, cM.w0, cM.w1, uM);  // column 2i+1 (the band may drop between 2i+2 and 2i+1)
    control_down(cM.adv, cA.w0, cA.w1, uA);    // column 2i
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
    double d4[NC], d3[NC], d2[NC], dn[NC], dm[NC];
    P::dFduT(r.tcB, Y4, uB, p, k4, d4);
    P::dFduT(r.tcM, Y3, uM, p, k3, d3);
    P::dFduT(r.tcM, Y2, uM, p, k2, d2);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      dn[c] = pend[c] + d4[c];  // column 2i+2  :112-116 (:119-120 at i = N-1)
      dm[c] = d2[c] + d3[c];    // column 2i+1  :105-109
    }
    fold(adv_g, cTop.w0, cTop.w1, dn);
    fold(cTop.adv, cM.w0, cM.w1, dm);
    adv_g = cM.adv;
    P::dFduT(r.tcA, xi, uA, p, k1, pend);
#pragma unroll
    for (int k = 0; k < NS; ++k) lam[k] = (((lam[k] + g1[k]) + g2[k]) + g3[k]) + g0[k];  // :86-88
#pragma unroll
    for (int c = 0; c < NC; ++c) unext[c] = uA[c];
    cTop = cA;
    cM = nM;
    cA = nA;
  };
  int i = N - 1;
  for (; i >= RD - 1; i -= RD) {
    step(i, xr[0]);
    step(i - 1, xr[1]);
    step(i - 2, xr[2]);
    step(i - 3, xr[3]);
  }
  // the last (N mod 4) steps: the slots hold them in order
  if (i >= 0) step(i, xr[0]);
  if (i >= 1) step(i - 1, xr[1]);
  if (i >= 2) step(i - 2, xr[2]);
  fold(adv_g, cTop.w0, cTop.w1, pend);  // left end point :101-102 (cTop is column 0 now)
  flush_row(rg + 1, gb);
  flush_row(rg, ga);
  if (a.lam0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) a.lam0[(size_t)k * B + b] = lam[k];
    a.lam0[(size_t)NS * B + b] = lamc;
  }
  if (warm == 1.234567e300) a.dJdv[b] = warm;
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
bool fused_banded_supported(Functor f, int nS, int nC) { return f == Functor::Logistic && nS >= 1 && nS <= 4 && nC == 1; }
int fused_banded_rec() { return kBandRec; }

template <class P>
static void run_fb(bool forward, const FbArgs& a, hipStream_t s) {
  constexpr int PF = P::NS <= 2 ? 4 : 3;
  const dim3 grid((a.batch + 63) / 64), block(64);
  if (forward)
    k_forward_fb<P, PF><<<grid, block, 0, s>>>(a);
  else
    k_backward_fb<P, PF><<<grid, block, 0, s>>>(a);
}
static int launch_fb(bool forward, const ProblemDesc& p, const FbArgs& a, hipStream_t s) {
  if (!fused_banded_supported(p.functor, p.nS, p.nC) || a.N < 1) return -1;
  switch (p.nS) {
    case 1: run_fb<LogisticK<1>>(forward, a, s); break;
    case 2: run_fb<LogisticK<2>>(forward, a, s); break;
    case 3: run_fb<LogisticK<3>>(forward, a, s); break;
    case 4: run_fb<LogisticK<4>>(forward, a, s); break;
    default: return -1;
  }
  return hip_rc_fb(hipGetLastError());
}
int launch_forward_fb(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, int r0, const double* CT,
                      const double* v, const double* x0, double* ck, double* J, hipStream_t s) {
  FbArgs a{};
  a.N = g.N; a.batch = batch; a.nBasis = nBasis; a.r0 = r0; a.REC = g.REC; a.ps = p.ps; a.pb = p.pb; a.pmask = p.pmask;
  a.CT = CT; a.v = v; a.x0 = x0; a.ck = ck; a.J = J;
  return launch_fb(true, p, a, s);
}
int launch_backward_fb(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, int r0, const double* CT,
                       const double* v, const double* ck, double* dJdv, double* lam0, hipStream_t s) {
  FbArgs a{};
  a.N = g.N; a.batch = batch; a.nBasis = nBasis; a.r0 = r0; a.REC = g.REC; a.ps = p.ps; a.pb = p.pb; a.pmask = p.pmask;
  a.CT = CT; a.v = v; a.ck = const_cast<double*>(ck); a.dJdv = dJdv; a.lam0 = lam0;
  return launch_fb(false, p, a, s);
}

}  // namespace ocs
