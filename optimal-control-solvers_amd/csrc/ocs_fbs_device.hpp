// ocs_fbs_device.hpp -- kernel templates of the grid forward-backward sweep (pchip coupling, costate
// RK4, control update, convergence bookkeeping).  Shared by ocs_fbs_kernels.hip (built-in problems) and the
// hipRTC path for user problems; device code only.
#pragma once
#include "ocs_device_common.hpp"

#ifndef OCS_FBS_USTORE
#define OCS_FBS_USTORE(v, p) (*(p) = (v))
#endif
namespace ocs {

// ---------------------------------------------------------------------------------------
// pchip (Fritsch-Carlson slopes as in MATLAB pchip / Moler's pchiptx), per lane
// ---------------------------------------------------------------------------------------
__device__ static inline int dsgn(double v) { return (v > 0.0) - (v < 0.0); }

// interior node: weighted harmonic mean of the neighbouring secants, 0 at a local extremum
__device__ static inline double pchip_interior(double del0, double del1, double w1, double w2) {
  if (dsgn(del0) * dsgn(del1) <= 0) return 0.0;
  const double a0 = fabs(del0), a1 = fabs(del1);
  const double dmax = fmax(a0, a1), dmin = fmin(a0, a1);
  return dmin / (w1 * (del0 / dmax) + w2 * (del1 / dmax));
}
// end node: non-centred three-point formula with the two shape-preserving corrections
__device__ static inline double pchip_end(double h0, double h1, double del0, double del1) {
  double d = ((2.0 * h0 + h1) * del0 - h0 * del1) / (h0 + h1);
  if (dsgn(d) != dsgn(del0))
    d = 0.0;
  else if (dsgn(del0) != dsgn(del1) && fabs(d) > fabs(3.0 * del0))
    d = 3.0 * del0;
  return d;
}

// Node tables (uniform): TN[0..n-1] node times, HN[0..n-2] spacings, W1/W2[1..n-2] slope weights, IH = 1./HN.
struct PchipTab {
  int n;
  const double* TN;
  const double* HN;
  const double* W1;
  const double* W2;
  const double* IH;  // 1 ./ HN
};

// slope at node k of the samples v(k) = V[(k*ld + row)*B + b]
__device__ static inline double pchip_slope_at(const PchipTab& T, const double* V, size_t ldB, int k) {
  const int n = T.n;
  auto val = [&](int j) OCS_INLINE { return V[(size_t)j * ldB]; };
  if (n == 2) return (val(1) - val(0)) / T.HN[0];
  if (k == 0) {
    const double d0 = (val(1) - val(0)) / T.HN[0], d1 = (val(2) - val(1)) / T.HN[1];
    return pchip_end(T.HN[0], T.HN[1], d0, d1);
  }
  if (k == n - 1) {
    const double d0 = (val(n - 1) - val(n - 2)) / T.HN[n - 2], d1 = (val(n - 2) - val(n - 3)) / T.HN[n - 3];
    return pchip_end(T.HN[n - 2], T.HN[n - 3], d0, d1);
  }
  const double d0 = (val(k) - val(k - 1)) / T.HN[k - 1], d1 = (val(k + 1) - val(k)) / T.HN[k];
  return pchip_interior(d0, d1, T.W1[k], T.W2[k]);
}
// cubic Hermite piece on an interval of length h with end values v0, v1 and end slopes dk, dk1 (pwch + ppval)
__device__ static inline double pchip_piece(double v0, double v1, double dk, double dk1, double h, double s) {
  const double del = (v1 - v0) / h;
  const double dzzdx = (del - dk) / h, dzdxdx = (dk1 - del) / h;
  const double c3 = (dzdxdx - dzzdx) / h, c2 = 2.0 * dzzdx - dzdxdx;
  return v0 + s * (dk + s * (c2 + s * c3));
}
// interpolant in interval k at local coordinate s = q - TN[k]
__device__ static inline double pchip_eval(const PchipTab& T, const double* V, size_t ldB, int k, double s) {
  const double v0 = V[(size_t)k * ldB], v1 = V[(size_t)(k + 1) * ldB];
  const double dk = pchip_slope_at(T, V, ldB, k), dk1 = pchip_slope_at(T, V, ldB, k + 1);
  return pchip_piece(v0, v1, dk, dk1, T.HN[k], s);
}

// interior slope from the two neighbouring secants: pchip_interior with one division less -- of del0/dmax and
// del1/dmax one is exactly +-1 and the other is +-(dmin/dmax), so the denominator is bit-identical
// Branch-free: lanes whose secants do not have the same strict sign compute a discarded quotient (possibly 0/0)
// and select 0; divergent branches around two divisions cost more than the divisions.
__device__ static inline double pchip_interior2(double del0, double del1, double w1, double w2) {
  const bool same = (del0 > 0.0 && del1 > 0.0) || (del0 < 0.0 && del1 < 0.0);
  const double a0 = fabs(del0), a1 = fabs(del1);
  const double sg = del0 > 0.0 ? 1.0 : -1.0;
  const bool first = a0 >= a1;  // dmax = a0
  const double dmax = first ? a0 : a1, dmin = first ? a1 : a0;
  const double q = sg * (dmin / dmax);
  const double r0 = first ? sg : q, r1 = first ? q : sg;
  const double d = dmin / (w1 * r0 + w2 * r1);
  return same ? d : 0.0;
}

// One value in each of R consecutive intervals i0 .. i0+R-1 of one sample row, at local coordinates sv[c]:
// R+4 loads, each secant and each node slope once (not once per adjacent interval), R cubic pieces.  Divisions by
// the spacing use the reciprocal table IH (round-off level difference to pchip_eval's true divisions).  mid[c]
// is valid for i0 + c < n - 1.
constexpr int kPchipRun = 8;
// from a register window w[j] = v(i0 - 2 + j), j = 0 .. R+3 (entries outside the table are never used by a valid
// result; for R = 1 entry 0 is not used at all)
template <int R>
__device__ static inline void pchip_run_w(const PchipTab& T, const double (&w)[R + 4], int i0, const double (&sv)[R],
                                          double (&mid)[R]) {
  const int n = T.n;
  auto clampi = [&](int k, int hi) OCS_INLINE { return k < 0 ? 0 : (k > hi ? hi : k); };
  double sec[R + 3];  // sec[j] = secant of interval i0 - 2 + j
#pragma unroll
  for (int j = 0; j < R + 3; ++j) sec[j] = (w[j + 1] - w[j]) * T.IH[clampi(i0 - 2 + j, n - 2)];
  double d[R + 1];
#pragma unroll
  for (int c = 0; c <= R; ++c) {
    const int k = i0 + c;  // node; intervals k-1 and k are sec[c+1], sec[c+2]
    double dk;
    if (n == 2)
      dk = sec[2];  // N = 1: the only interval is sec[2] (i0 = 0)
    else if (k == 0)
      dk = pchip_end(T.HN[0], T.HN[1], sec[c + 2], sec[c + 3]);
    else if (k == n - 1)
      dk = pchip_end(T.HN[n - 2], T.HN[n - 3], sec[c + 1], sec[c]);
    else if (k < n - 1)
      dk = pchip_interior1(sec[c + 1], sec[c + 2], T.W1[k], T.W2[k]);
    else
      dk = 0.0;
    d[c] = dk;
  }
#pragma unroll
  for (int c = 0; c < R; ++c) {
    const int i = i0 + c;
    double m = 0.0;
    if (i < n - 1) {
      const double ih = T.IH[i], s = sv[c], del = sec[c + 2];
      const double dzzdx = (del - d[c]) * ih, dzdxdx = (d[c + 1] - del) * ih;
      const double c3 = (dzdxdx - dzzdx) * ih, c2 = 2.0 * dzzdx - dzdxdx;
      m = w[c + 2] + s * (d[c] + s * (c2 + s * c3));
    }
    mid[c] = m;
  }
}
template <int R>
__device__ static inline void pchip_run(const PchipTab& T, const double* V, size_t ldB, int i0, const double (&sv)[R],
                                        double (&mid)[R]) {
  const int n = T.n;
  double w[R + 4];
#pragma unroll
  for (int j = 0; j < R + 4; ++j) {
    int k = i0 - 2 + j;
    k = k < 0 ? 0 : (k > n - 1 ? n - 1 : k);
    w[j] = V[(size_t)k * ldB];
  }
  pchip_run_w<R>(T, w, i0, sv, mid);
}

// The same values one interval at a time: the window of one sample row while a thread walks up its intervals.  va = v(i),
// sp / sc / sn = secants of intervals i-1, i, i+1, di = slope at node i; step(i) returns the interpolant at local coordinate
// s of interval i and moves the window to i + 1.  Formulas and operands are pchip_run_w's (reciprocal spacings, pchip_interior1,
// pchip_end at the two ends of the table): bit-equal results.
// the tables behind constant-address-space pointers: with plain pointers every table entry read after a store of the kernel
// is a vector load again (the stores might alias it), waited for at once -- and with it every load that was issued ahead
struct PchipTabU {
  int n;
  uniform_ptr TN, HN, W1, W2, IH;
  __device__ explicit PchipTabU(const PchipTab& t)
      : n(t.n), TN(as_uniform(t.TN)), HN(as_uniform(t.HN)), W1(as_uniform(t.W1)), W2(as_uniform(t.W2)), IH(as_uniform(t.IH)) {}
};
struct PchipSlide {
  double va, vb, vc, sc, di;   // v(i), v(i+1), v(i+2), secant of interval i, slope at node i
  __device__ static inline int cl(int k, int hi) { return k < 0 ? 0 : (k > hi ? hi : k); }
  __device__ inline void init(const PchipTabU& T, const double* V, size_t ldB, int i) {
    const int n = T.n;
    const double vm = V[(size_t)cl(i - 1, n - 1) * ldB];
    va = V[(size_t)cl(i, n - 1) * ldB];
    vb = V[(size_t)cl(i + 1, n - 1) * ldB];
    vc = V[(size_t)cl(i + 2, n - 1) * ldB];
    const double sp = (va - vm) * T.IH[cl(i - 1, n - 2)], sn = (vc - vb) * T.IH[cl(i + 1, n - 2)];
    sc = (vb - va) * T.IH[cl(i, n - 2)];
    if (n == 2) di = sc;
    else if (i == 0) di = pchip_end(T.HN[0], T.HN[1], sc, sn);
    else if (i >= n - 1) di = 0.0;
    else di = pchip_interior1(sp, sc, T.W1[i], T.W2[i]);
  }
  // vd = v(min(i + 3, n - 1)): loaded by the caller, who can issue the loads of all rows ahead of the arithmetic
  __device__ static inline double next(const PchipTabU& T, const double* V, size_t ldB, int i) {
    return V[(size_t)cl(i + 3, T.n - 1) * ldB];
  }
  // the cubic of interval i as v0 + s (d0 + s (c2 + s c3)), then the window moves to i + 1 (several points per interval)
  __device__ inline void coef(const PchipTabU& T, const double* V, size_t ldB, int i, double vd, double& v0, double& d0, double& c2,
                              double& c3) {
    const int n = T.n, k = i + 1;
    const double sn = (vc - vb) * T.IH[cl(i + 1, n - 2)];
    double dn;
    if (n == 2) dn = sc;
    else if (k == n - 1) {
      const double sp = (va - V[(size_t)cl(i - 1, n - 1) * ldB]) * T.IH[cl(i - 1, n - 2)];
      dn = pchip_end(T.HN[n - 2], T.HN[n - 3], sc, sp);
    } else dn = pchip_interior1(sc, sn, T.W1[k], T.W2[k]);
    const double ih = T.IH[i], del = sc;
    const double dzzdx = (del - di) * ih, dzdxdx = (dn - del) * ih;
    c3 = (dzdxdx - dzzdx) * ih;
    c2 = 2.0 * dzzdx - dzdxdx;
    v0 = va;
    d0 = di;
    sc = sn;
    va = vb; vb = vc; vc = vd; di = dn;
  }
  __device__ inline double step(const PchipTabU& T, const double* V, size_t ldB, int i, double s, double vd) {   // i < n - 1
    const int n = T.n, k = i + 1;
    const double sn = (vc - vb) * T.IH[cl(i + 1, n - 2)];
    double dn;
    if (n == 2) dn = sc;
    else if (k == n - 1) {   // the last node of the table: the secant of the interval before this one once more (a load)
      const double sp = (va - V[(size_t)cl(i - 1, n - 1) * ldB]) * T.IH[cl(i - 1, n - 2)];
      dn = pchip_end(T.HN[n - 2], T.HN[n - 3], sc, sp);
    } else dn = pchip_interior1(sc, sn, T.W1[k], T.W2[k]);
    const double ih = T.IH[i], del = sc;
    const double dzzdx = (del - di) * ih, dzdxdx = (dn - del) * ih;
    const double c3 = (dzdxdx - dzzdx) * ih, c2 = 2.0 * dzzdx - dzdxdx;
    const double m = va + s * (di + s * (c2 + s * c3));
    sc = sn;
    va = vb; vb = vc; vc = vd; di = dn;
    return m;
  }
};

template <int R>
__device__ static inline void pchip_mid_run(const PchipTab& T, const double* V, size_t ldB, int i0,
                                            const double* __restrict__ TM, double (&mid)[R]) {
  double sv[R];
#pragma unroll
  for (int c = 0; c < R; ++c) {
    const int i = i0 + c < T.n - 1 ? i0 + c : T.n - 2;
    sv[c] = TM[i] - T.TN[i];
  }
  pchip_run<R>(T, V, ldB, i0, sv, mid);
}

// midpoint samples: out[i][r][b] = pchip(V(r,:))(t_mid_i)   for rows r < nrows of V [n][ld][B]
// blockIdx.y = run of kPchipRun intervals
__global__ __launch_bounds__(256) void k_pchip_mid(PchipTab T, int nrows, int ld, int batch,
                                                   const double* __restrict__ TM, const double* __restrict__ V,
                                                   double* __restrict__ out, int ldb, const int* __restrict__ gate) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int i0 = blockIdx.y * kPchipRun;
  if (gate && *gate == 0) return;
  if (b >= batch || i0 >= T.n - 1) return;
  const size_t B = (size_t)(ldb ? ldb : batch);  // row distance (a window of a larger batch) / trajectories here
  for (int r = 0; r < nrows; ++r) {
    double mid[kPchipRun];
    pchip_mid_run<kPchipRun>(T, V + (size_t)r * B + b, (size_t)ld * B, i0, TM, mid);
#pragma unroll
    for (int c = 0; c < kPchipRun; ++c)
      if (i0 + c < T.n - 1) out[((size_t)(i0 + c) * nrows + r) * B + b] = mid[c];
  }
}

// ---------------------------------------------------------------------------------------
// costate pass: lam' = adjointRHS(t, x(t), lam, u(t)), lam(TF) = 0, RK4 from TF down to T0
// (compute_x_lam.m:11-14 with odevr7 -> RK4 on the node grid)
// ---------------------------------------------------------------------------------------
struct CostateArgs {
  int N, batch;
  const double* REC;
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* x;     // [N+1][ldx][B] node states (rows 0..nS-1)
  int ldx;
  const double* xmid;  // [N][nS][B]
  const double* u;     // [2N+1][nC][B]
  const int* frozen;   // optional [B]: instances with frozen[b] != 0 (converged in an earlier sweep) store nothing
  double* dump;        // [B] scratch for their stores
  double* lam;         // [N+1][nS][B]
  int ld;              // row distance when the launch covers a window of a larger batch; 0 = batch
  const int* gate = nullptr;   // optional: the launch does nothing if *gate == 0
};

template <class P, int PF>
__global__ __launch_bounds__(64) void k_costate(const CostateArgs a) {
  if (a.gate && *a.gate == 0) return;
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC;
  using Rec = StepRec<NTC>;
  const int b0 = blockIdx.x * 64 + threadIdx.x;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)(a.ld ? a.ld : a.batch);
  const int N = a.N;
  const uniform_ptr PS = as_uniform(a.ps);
  const typename P::Par p = P::load(ParamSrc{PS, a.pb, a.pmask, B, b});
  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));

  const double* xp = a.x + b;
  const double* mp = a.xmid + b;
  const double* up = a.u + b;
  // lam is walked downwards with a per-lane row stride; a frozen instance has stride 0 on a scratch double
  const bool fz = a.frozen && a.frozen[b] != 0;
  const size_t lrow = fz ? 0 : B;
  double* lp = fz ? a.dump + b : a.lam + b;

  double l[NS], xB[NS], uB[NC];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    l[k] = 0.0;  // lam0 = 0*x0   compute_x_lam.m:4
    xB[k] = xp[((size_t)N * a.ldx + k) * B];
    lp[((size_t)N * NS + k) * lrow] = 0.0;
  }
  double* lq = lp + (size_t)N * NS * lrow;  // column i + 1 of lam; steps run i = N-1 .. 0
#pragma unroll
  for (int c = 0; c < NC; ++c) uB[c] = up[((size_t)(2 * N) * NC + c) * B];

  // adjointRHS(t, x, lam, u) = -dFdx_times_vec(t, [x;0], u, [lam;1])(1:nS)
  auto rhs = [&](const double* tc, const double* x, const double* lam, const double* u, double* out) OCS_INLINE {
    double v[NS + 1], g[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = lam[k];
    v[NS] = 1.0;
    P::dFdxT(tc, x, u, p, v, g);
#pragma unroll
    for (int k = 0; k < NS; ++k) out[k] = -g[k];
  };

  static_assert(PF <= kRecPad, "record ring deeper than the table padding");
  Rec rq[PF];
  const double* recp = a.REC + (size_t)(N - 1) * rec_stride(NTC);  // walks down; padded before step 0
#pragma unroll
  for (int q = 0; q < PF; ++q) {
    rq[q] = load_rec<NTC>(recp);
    recp -= rec_stride(NTC);
  }
  // node/midpoint states and control samples are prefetched one chunk (CH steps) ahead, ping-pong
  auto step = [&](const Rec& r, int i, const double* cxA, const double* cxM, const double* cuA,
                  const double* cuM) OCS_INLINE {
    // classical RK4 with step -h from node i+1 to node i
    double k1[NS], k2[NS], k3[NS], k4[NS], L[NS];
    rhs(r.tcB, xB, l, uB, k1);
#pragma unroll
    for (int k = 0; k < NS; ++k) L[k] = __builtin_fma(-r.hh, k1[k], l[k]);
    rhs(r.tcM, cxM, L, cuM, k2);
#pragma unroll
    for (int k = 0; k < NS; ++k) L[k] = __builtin_fma(-r.hh, k2[k], l[k]);
    rhs(r.tcM, cxM, L, cuM, k3);
#pragma unroll
    for (int k = 0; k < NS; ++k) L[k] = __builtin_fma(-r.h, k3[k], l[k]);
    rhs(r.tcA, cxA, L, cuA, k4);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      l[k] = __builtin_fma(-r.h6, __builtin_fma(2.0, k3[k], __builtin_fma(2.0, k2[k], k1[k])) + k4[k], l[k]);
      (lq - NS * lrow)[k * lrow] = l[k];
      xB[k] = cxA[k];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) uB[c] = cuA[c];
    lq -= NS * lrow;
    (void)i;
  };
  auto next_rec = [&]() OCS_INLINE {
    const Rec r = rq[0];
#pragma unroll
    for (int q = 0; q + 1 < PF; ++q) rq[q] = rq[q + 1];
    rq[PF - 1] = load_rec<NTC>(recp);
    recp -= rec_stride(NTC);
    return r;
  };
  // steps per prefetched chunk: two chunks of CH (2 NS + 2 NC) doubles are live at a time -- 8 steps for the small problems,
  // fewer as the state grows (8 steps of a six-state problem are 576 registers: the pass ran out of scratch memory at 3.4 ms
  // where it takes 1 ms)
#ifdef OCS_COSTATE_CH
  constexpr int CH = OCS_COSTATE_CH;
#else
  constexpr int CH = NS + NC <= 3 ? 8 : (NS + NC <= 6 ? 4 : (NS + NC <= 10 ? 2 : 1));
#endif
  const int nch = N / CH;
  for (int i = N - 1; i >= nch * CH; --i) {  // remainder steps at the top, direct loads
    double xA[NS], xM[NS], uA[NC], uM[NC];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      xA[k] = xp[((size_t)i * a.ldx + k) * B];
      xM[k] = mp[((size_t)i * NS + k) * B];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uA[c] = up[((size_t)(2 * i) * NC + c) * B];
      uM[c] = up[((size_t)(2 * i + 1) * NC + c) * B];
    }
    const Rec r = next_rec();
    step(r, i, xA, xM, uA, uM);
  }
  struct Chunk {
    double xA[CH][NS], xM[CH][NS], uA[CH][NC], uM[CH][NC];
  };
  Chunk c0, c1;
  auto load_chunk = [&](Chunk& d, int c) OCS_INLINE {
#pragma unroll
    for (int s = CH - 1; s >= 0; --s) {
      const int i = c * CH + s;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        d.xA[s][k] = xp[((size_t)i * a.ldx + k) * B];
        d.xM[s][k] = mp[((size_t)i * NS + k) * B];
      }
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        d.uA[s][cc] = up[((size_t)(2 * i) * NC + cc) * B];
        d.uM[s][cc] = up[((size_t)(2 * i + 1) * NC + cc) * B];
      }
    }
  };
  auto run_chunk = [&](const Chunk& d, int c) OCS_INLINE {
#pragma unroll
    for (int s = CH - 1; s >= 0; --s) {
      const Rec r = next_rec();
      step(r, c * CH + s, d.xA[s], d.xM[s], d.uA[s], d.uM[s]);
    }
  };
  int c = nch - 1;
  if (c >= 0) load_chunk(c0, c);
  for (; c >= 1; c -= 2) {
    load_chunk(c1, c - 1);
    run_chunk(c0, c);
    if (c >= 2) load_chunk(c0, c - 2);
    run_chunk(c1, c - 1);
  }
  if (c == 0) run_chunk(c0, 0);
  if (warm == 1.234567e300) lp[0] = warm;  // never true; keeps the table sweep alive
}

// ---------------------------------------------------------------------------------------
// new control on the grid: uNew(t_j) = ControlChar(t_j, x(t_j), lam(t_j))   fb_sweep.m:96
// nodes use the node samples, midpoints the pchip midpoints.  Runs AFTER the convergence decision of
// the sweep and overwrites u in place for the instances that continue.
// ---------------------------------------------------------------------------------------
struct ControlGridArgs {
  int N, batch;
  const double* TU;    // [2N+1][NTU]
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* lb;
  const double* ub;
  const double* x;     // [N+1][ldx][B]
  int ldx;
  const double* xmid;  // [N][nS][B]
  const double* lam;   // [N+1][nS][B]; its pchip midpoints are formed here (one run of intervals per thread)
  PchipTab T;
  const double* TM;
  double* u;           // the control grid, updated in place
  const int* status;   // only instances that are still active (status 0) take the new control (fb_sweep.m:85);
                       // a converged instance keeps its old one for the final sweep (:82)
  // Error points == grid nodes: the weighted change |uNew - u| / (relTol |u| + absTol) at the nodes (fb_sweep.m:107)
  // is folded in here, against the node samples of u that are being replaced, as one partial maximum per run of
  // intervals: metric [runs][B] (-1: no valid value), reduced by k_fbs_advance.  The update then happens BEFORE the
  // convergence decision: an instance that turns out converged has taken uNew, which is harmless because from
  // then on it is frozen (its x, lam, J are the ones already computed from the old control, fb_sweep.m:82).
  double* metric;
  double relTol, absTol;
  int ld;  // row distance when the launch covers a window of a larger batch; 0 = batch
  const int* gate;  // optional: the launch does nothing if *gate == 0
  double relax;     // the samples become u + relax (uNew - u) (1: the reference's u = uNew, fb_sweep.m:85)
};

// the control-grid run of larger state vectors (k_control_grid): OWNX = the midpoints of x are formed here too
template <class P, bool OWNX>
__device__ static inline void control_grid_slide(const ControlGridArgs& a, int b, int i0, size_t B, const typename P::Par& p,
                                                 const double* lb, const double* ub) {
  constexpr int NS = P::NS, NC = P::NC, NTU = P::NTU, R = kPchipRun;
  const int N = a.N;
  const PchipTabU TU_(a.T);
  const uniform_ptr TMu = as_uniform(a.TM), TUu = as_uniform(a.TU);
  // Larger state vectors: the run as a ROLLED loop over its intervals with a sliding window per row (PchipSlide) instead of
  // the unrolled register tables below -- those are 2 NS R doubles and 17 inlined ControlChar evaluations, 170 KB of code
  // for a six-state problem, and the kernel ran at the speed of the instruction cache (452 us where the data take 100).
  // Same formulas on the same operands: bit-equal values.
  double nmax = 0.0, dmax = 1.0;
  bool any = false;
  auto emit = [&](int j, const double* x, const double* lam, const double* uold) OCS_INLINE {
    double tu[NTU], u[NC];
#pragma unroll
    for (int k = 0; k < NTU; ++k) tu[k] = TUu[(size_t)j * NTU + k];
    P::control_char(tu, x, lam, p, lb, ub, u);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double* dst = a.u + ((size_t)j * NC + c) * B + b;
      const double o = uold[c];
      if (a.metric && !(j & 1)) {
        const double n = fabs(u[c] - o), d = a.relTol * fabs(o) + a.absTol;
        if (n == n && d == d && !(n == 0.0 && d == 0.0)) {
          if (!any || n * dmax > nmax * d) {
            nmax = n;
            dmax = d;
          }
          any = true;
        }
      }
      if (a.relax != 1.0) u[c] = __builtin_fma(a.relax, u[c] - o, o);
      OCS_FBS_USTORE(u[c], dst);
    }
  };
  PchipSlide sl[NS], sx[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    sl[k].init(TU_, a.lam + (size_t)k * B + b, (size_t)NS * B, i0);
    if constexpr (OWNX) sx[k].init(TU_, a.x + (size_t)k * B + b, (size_t)a.ldx * B, i0);
  }
  const int iend = i0 + R < N ? i0 + R : N;
  // every load of an interval is issued one interval ahead, at the top of the loop body and ahead of the branches of the
  // slope formulas (the waves of this kernel spent 80 % of their time waiting for one load after the other)
  double nl[NS], nx[NS], nm[NS], no[2][NC];
  auto fetch = [&](int i) OCS_INLINE {
    const int ic = i < N ? i : N - 1;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      nl[k] = PchipSlide::next(TU_, a.lam + (size_t)k * B + b, (size_t)NS * B, i);
      if constexpr (OWNX) {
        nx[k] = PchipSlide::next(TU_, a.x + (size_t)k * B + b, (size_t)a.ldx * B, i);
      } else {
        nx[k] = a.x[((size_t)ic * a.ldx + k) * B + b];
        nm[k] = a.xmid[((size_t)ic * NS + k) * B + b];
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {   // the samples that are replaced (the weighted change, the damped update)
      no[0][c] = a.u[((size_t)(2 * ic) * NC + c) * B + b];
      no[1][c] = a.u[((size_t)(2 * ic + 1) * NC + c) * B + b];
    }
  };
  fetch(i0);
#pragma unroll 1
  for (int i = i0; i < iend; ++i) {
    double cl_[NS], cx[NS], cm[NS], co[2][NC];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      cl_[k] = nl[k];
      cx[k] = nx[k];
      if constexpr (!OWNX) cm[k] = nm[k];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      co[0][c] = no[0][c];
      co[1][c] = no[1][c];
    }
    fetch(i + 1);
    const double sm = TMu[i] - TU_.TN[i];
    double x[NS], lam[NS], xm[NS], lm[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      lam[k] = sl[k].va;
      lm[k] = sl[k].step(TU_, a.lam + (size_t)k * B + b, (size_t)NS * B, i, sm, cl_[k]);
      if constexpr (OWNX) {
        x[k] = sx[k].va;
        xm[k] = sx[k].step(TU_, a.x + (size_t)k * B + b, (size_t)a.ldx * B, i, sm, cx[k]);
      } else {
        x[k] = cx[k];
        xm[k] = cm[k];
      }
    }
    emit(2 * i, x, lam, co[0]);
    emit(2 * i + 1, xm, lm, co[1]);
  }
  if (i0 + R >= N) {  // the run that ends the grid also owns the last node
    double x[NS], lam[NS], uo[NC];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      x[k] = a.x[((size_t)N * a.ldx + k) * B + b];
      lam[k] = a.lam[((size_t)N * NS + k) * B + b];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) uo[c] = a.u[((size_t)(2 * N) * NC + c) * B + b];
    emit(2 * N, x, lam, uo);
  }
  if (a.metric) a.metric[(size_t)blockIdx.y * B + b] = any ? nmax / dmax : -1.0;
}

// OWNX: larger state vectors (NS > 4) only -- an instance for launches without a midpoint array of x (a.xmid == nullptr) carries
// a second set of sliding windows (282 registers: one wave per SIMD); it is kept out of the common instance (216: two waves), and
// not instantiated at all while no costate kernel of these shapes forms the midpoints itself (the launcher refuses).  Up to four
// states one instance (OWNX = false) serves both cases and looks at a.xmid itself.
template <class P, bool OWNX = false>
__global__ __launch_bounds__(256) void k_control_grid(const ControlGridArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTU = P::NTU, R = kPchipRun;
  if (a.gate && *a.gate == 0) return;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int i0 = blockIdx.y * R;  // first interval of this thread's run
  if (b >= a.batch || a.status[b] != 0) return;
  const size_t B = (size_t)(a.ld ? a.ld : a.batch);
  const int N = a.N;
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  double lb[NC], ub[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    lb[c] = a.lb[c];
    ub[c] = a.ub[c];
  }
  const bool ownx = a.xmid == nullptr;  // no midpoint array of x: form those here as well
  if constexpr (NS > 4) {
    control_grid_slide<P, OWNX>(a, b, i0, B, p, lb, ub);   // (the launcher picks the instance by a.xmid)
    return;
  }
  double lmid[NS][R];
#pragma unroll
  for (int k = 0; k < NS; ++k) pchip_mid_run<R>(a.T, a.lam + (size_t)k * B + b, (size_t)NS * B, i0, a.TM, lmid[k]);
  double xmr[NS][R];
  if (ownx) {
#pragma unroll
    for (int k = 0; k < NS; ++k) pchip_mid_run<R>(a.T, a.x + (size_t)k * B + b, (size_t)a.ldx * B, i0, a.TM, xmr[k]);
  }
  // the largest weighted change as a fraction nmax / dmax: ratios are compared by cross-multiplication and divided
  // once per thread (fp64 divisions were a large share of this kernel's instructions)
  double nmax = 0.0, dmax = 1.0;
  bool any = false;
  auto emit = [&](int j, const double* x, const double* lam) OCS_INLINE {  // grid point j
    double tu[NTU], u[NC];
#pragma unroll
    for (int k = 0; k < NTU; ++k) tu[k] = as_uniform(a.TU)[(size_t)j * NTU + k];
    P::control_char(tu, x, lam, p, lb, ub, u);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double* dst = a.u + ((size_t)j * NC + c) * B + b;
      if (a.metric && !(j & 1)) {  // a node: max() skips NaN (:108)
        const double o = *dst;
        const double n = fabs(u[c] - o), d = a.relTol * fabs(o) + a.absTol;
        if (n == n && d == d && !(n == 0.0 && d == 0.0)) {  // n / d is not NaN
          if (!any || n * dmax > nmax * d) {
            nmax = n;
            dmax = d;
          }
          any = true;
        }
      }
      if (a.relax != 1.0) u[c] = __builtin_fma(a.relax, u[c] - *dst, *dst);   // damped update (extension)
      OCS_FBS_USTORE(u[c], dst);
    }
  };
#pragma unroll
  for (int c = 0; c < R; ++c) {
    const int i = i0 + c;
    if (i < N) {
      double x[NS], lam[NS];
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        x[k] = a.x[((size_t)i * a.ldx + k) * B + b];
        lam[k] = a.lam[((size_t)i * NS + k) * B + b];
      }
      emit(2 * i, x, lam);
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        x[k] = ownx ? xmr[k][c] : a.xmid[((size_t)i * NS + k) * B + b];
        lam[k] = lmid[k][c];
      }
      emit(2 * i + 1, x, lam);
    }
  }
  if (i0 + R >= N) {  // the run that ends the grid also owns the last node
    double x[NS], lam[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      x[k] = a.x[((size_t)N * a.ldx + k) * B + b];
      lam[k] = a.lam[((size_t)N * NS + k) * B + b];
    }
    emit(2 * N, x, lam);
  }
  if (a.metric) a.metric[(size_t)blockIdx.y * B + b] = any ? nmax / dmax : -1.0;
}

// ---------------------------------------------------------------------------------------
// vectorInterpolant(x, v, method)(tq) for a batch (functions/vectorInterpolant.m:1-12): out[q][c][b] from samples
// v[node][c][b]; KQ/SQ: interval index and local coordinate of each query point (host-made, like the error points
// of the sweep); method 0 'linear', 2 'previous' (KQ = -1: before the first node -> NaN; SQ unused), 3 'pchip'.
// One thread takes kInterpPts consecutive query points of one (component, instance) column.
// ---------------------------------------------------------------------------------------
// 'pchip' with the query points sorted by interval (host: QS[k] .. QS[k+1] are the positions, in the sorted order, of the points
// of interval k; QI their indices in the caller's order; SS their local coordinates): a thread takes kInterpRun consecutive
// intervals of one (component, instance) column, forms each node slope once from a register window (pchip_run_w's formulas:
// reciprocal spacings) and evaluates every point of its intervals -- 2 loads per output at two points per interval where the
// point-by-point kernel below makes 8 and divides 6 times.
constexpr int kInterpRun = 4;
__global__ __launch_bounds__(256) void k_interp_pchip_sorted(PchipTab T, int nComp, int batch, const int* __restrict__ QS,
                                                             const int* __restrict__ QI, const double* __restrict__ SS,
                                                             const double* __restrict__ V, double* __restrict__ out) {
  constexpr int R = kInterpRun;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.z, n = T.n;
  const int i0 = blockIdx.y * R;
  if (b >= batch || i0 > n - 2) return;
  const size_t B = (size_t)batch, ldB = (size_t)nComp * B;
  const double* v = V + (size_t)c * B + b;
  auto clampi = [&](int k, int hi) OCS_INLINE { return k < 0 ? 0 : (k > hi ? hi : k); };
  double w[R + 4];
#pragma unroll
  for (int j = 0; j < R + 4; ++j) w[j] = v[(size_t)clampi(i0 - 2 + j, n - 1) * ldB];
  double sec[R + 3], d[R + 1];
#pragma unroll
  for (int j = 0; j < R + 3; ++j) sec[j] = (w[j + 1] - w[j]) * T.IH[clampi(i0 - 2 + j, n - 2)];
#pragma unroll
  for (int cc = 0; cc <= R; ++cc) {
    const int k = i0 + cc;
    double dk;
    if (n == 2) dk = sec[2];
    else if (k == 0) dk = pchip_end(T.HN[0], T.HN[1], sec[cc + 2], sec[cc + 3]);
    else if (k == n - 1) dk = pchip_end(T.HN[n - 2], T.HN[n - 3], sec[cc + 1], sec[cc]);
    else if (k < n - 1) dk = pchip_interior1(sec[cc + 1], sec[cc + 2], T.W1[k], T.W2[k]);
    else dk = 0.0;
    d[cc] = dk;
  }
#pragma unroll
  for (int cc = 0; cc < R; ++cc) {
    const int k = i0 + cc;
    if (k > n - 2) break;
    const double ih = T.IH[k], del = sec[cc + 2];
    const double dzzdx = (del - d[cc]) * ih, dzdxdx = (d[cc + 1] - del) * ih;
    const double c3 = (dzdxdx - dzzdx) * ih, c2 = 2.0 * dzzdx - dzdxdx;
    const int q1 = QS[k + 1];
    for (int q = QS[k]; q < q1; ++q) {   // (wave-uniform trip count)
      const double sq = SS[q];
      out[((size_t)QI[q] * nComp + c) * B + b] = w[cc + 2] + sq * (d[cc] + sq * (c2 + sq * c3));
    }
  }
}

constexpr int kInterpPts = 4;
__global__ __launch_bounds__(256) void k_interp(int method, PchipTab T, int nComp, int nq, int batch, const int* KQ,
                                                const double* SQ, const double* V, double* out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.z;
  if (b >= batch) return;
  const size_t B = (size_t)batch, ldB = (size_t)nComp * B;
  const double* v = V + (size_t)c * B + b;
  const int q0 = blockIdx.y * kInterpPts;
#pragma unroll
  for (int e = 0; e < kInterpPts; ++e) {
    const int q = q0 + e;
    if (q >= nq) break;
    const int k = KQ[q];
    double val;
    if (method == 2) {
      val = k < 0 ? __builtin_nan("") : v[(size_t)k * ldB];
    } else if (method == 0) {
      const double v0 = v[(size_t)k * ldB], v1 = v[(size_t)(k + 1) * ldB];
      val = v0 + (v1 - v0) * (SQ[q] / T.HN[k]);
    } else {
      val = pchip_eval(T, v, ldB, k, SQ[q]);
    }
    out[((size_t)q * nComp + c) * B + b] = val;
  }
}

// ---------------------------------------------------------------------------------------
// control at arbitrary points: out[q][c][b] = ControlChar(tq, x(tq), lam(tq)) with pchip x, lam
// (errorPts fb_sweep.m:107, interpPts :123).  KQ/SQ: interval index and local coordinate of tq.
// With `metric` set this is the error-point mode of the sweep: out is read (old control) and replaced in place.
// ---------------------------------------------------------------------------------------
struct ControlPtsArgs {
  int nq, batch;
  PchipTab T;
  const int* KQ;
  const double* SQ;
  const double* TUQ;   // [nq][NTU] ControlChar time coefficients at the query points
  const double* ps;
  const double* pb;
  unsigned pmask;
  const double* lb;
  const double* ub;
  const double* x;
  int ldx;
  const double* lam;
  double* out;
  const int* usel;
  long long odelta;
  // error-point mode (usel != nullptr): the weighted change |uNew - u| / (relTol |u| + absTol) against the
  // instance's current buffer is folded into metric[b] (bit pattern of a non-negative double, atomicMax)
  double* metric;  // [blocks in y][B]: this block's maximum for instance b, or -1 if none of its values was valid
  int* anyvalid;   // unused (kept for the argument layout)
  double relTol, absTol;
  double relax;    // error-point mode: the samples become u + relax (uNew - u) (1: the reference's u = uNew, fb_sweep.m:85)
  const int* gate = nullptr;   // optional: the launch does nothing if *gate == 0 (a sweep enqueued ahead of the verdict before it)
};

constexpr int kPtsPerThread = 8;

template <class P>
__global__ __launch_bounds__(256) void k_control_pts(const ControlPtsArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTU = P::NTU;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (a.gate && *a.gate == 0) return;
  if (b >= a.batch) return;
  double wmax = 0.0;
  bool any = false;
  const int q0 = (int)blockIdx.y * kPtsPerThread;
  const size_t B = (size_t)a.batch;
  // Block-uniform fast path: the points of this block lie in consecutive intervals (error / interpolation points
  // as dense as the grid, the default of fb_sweep.m:21-22): one register window per row serves all of them.
  // (larger state vectors take the points one by one in a rolled loop: the windows of 2 NS rows and kPtsPerThread inlined
  // ControlChar evaluations were 87 000 instructions and 784 bytes of scratch per thread for a six-state problem)
  constexpr bool kWindows = NS <= 4;
  constexpr int kUnrollPts = kWindows ? kPtsPerThread : 1;
  bool aligned = kWindows && q0 + kPtsPerThread <= a.nq;
  const int kq0 = a.KQ[q0];
  double sq[kPtsPerThread];
  if (aligned) {
#pragma unroll
    for (int c = 0; c < kPtsPerThread; ++c) {
      aligned = aligned && a.KQ[q0 + c] == kq0 + c;
      sq[c] = a.SQ[q0 + c];
    }
  }
  const PchipTabU TU_(a.T);
  PchipSlide sl[NS], sx[NS];   // (larger state vectors: sliding windows in the place of the register tables)
  bool slide = !kWindows && q0 + kPtsPerThread <= a.nq;
  if constexpr (!kWindows) {
#pragma unroll 1
    for (int c = 1; c < kPtsPerThread && slide; ++c) slide = a.KQ[q0 + c] == kq0 + c;
    if (slide) {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        if (P::CC_READS_X) sx[k].init(TU_, a.x + (size_t)k * B + b, (size_t)a.ldx * B, kq0);
        sl[k].init(TU_, a.lam + (size_t)k * B + b, (size_t)NS * B, kq0);
      }
    }
  }
  double xw[NS][kPtsPerThread], lw[NS][kPtsPerThread];
  if (aligned) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      // (a ControlChar that does not read x -- P::CC_READS_X -- leaves the state out: half the traffic of this kernel)
      if (P::CC_READS_X) pchip_run<kPtsPerThread>(a.T, a.x + (size_t)k * B + b, (size_t)a.ldx * B, kq0, sq, xw[k]);
      pchip_run<kPtsPerThread>(a.T, a.lam + (size_t)k * B + b, (size_t)NS * B, kq0, sq, lw[k]);
    }
  }
#pragma unroll kUnrollPts
  for (int cq = 0; cq < kPtsPerThread; ++cq) {
  const int q = q0 + cq;
  if (q >= a.nq) break;
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  double x[NS], lam[NS], tu[NTU], lb[NC], ub[NC], u[NC];
  if (aligned) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      x[k] = P::CC_READS_X ? xw[k][cq] : 0.0;
      lam[k] = lw[k][cq];
    }
  } else if (slide) {
    const double s = a.SQ[q];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      x[k] = P::CC_READS_X ? sx[k].step(TU_, a.x + (size_t)k * B + b, (size_t)a.ldx * B, kq0 + cq, s, PchipSlide::next(TU_, a.x + (size_t)k * B + b, (size_t)a.ldx * B, kq0 + cq)) : 0.0;
      lam[k] = sl[k].step(TU_, a.lam + (size_t)k * B + b, (size_t)NS * B, kq0 + cq, s, PchipSlide::next(TU_, a.lam + (size_t)k * B + b, (size_t)NS * B, kq0 + cq));
    }
  } else {
    const int k0 = a.KQ[q];
    const double s = a.SQ[q];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      x[k] = P::CC_READS_X ? pchip_eval(a.T, a.x + (size_t)k * B + b, (size_t)a.ldx * B, k0, s) : 0.0;
      lam[k] = pchip_eval(a.T, a.lam + (size_t)k * B + b, (size_t)NS * B, k0, s);
    }
  }
#pragma unroll
  for (int k = 0; k < NTU; ++k) tu[k] = as_uniform(a.TUQ)[(size_t)q * NTU + k];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    lb[c] = a.lb[c];
    ub[c] = a.ub[c];
  }
  P::control_char(tu, x, lam, p, lb, ub, u);
  // Error-point mode: `out` holds the instance's current control at these points and is replaced in place.
  // An instance that turns out converged (k_fbs_advance) never reads these samples again -- its x, lam, J already
  // belong to the old control (fb_sweep.m:82) -- and one that continues takes uNew anyway (:85), so no second
  // buffer and no per-instance select is needed: loads and stores stay coalesced.
  double* dst = a.out;
  if (a.metric) {  // fb_sweep.m:107  abs(uNew - u) ./ (uRelTol*abs(u) + uAbsTol), max() skips NaN (:108)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double o = dst[((size_t)q * NC + c) * B + b];
      const double w = fabs(u[c] - o) / (a.relTol * fabs(o) + a.absTol);
      if (w == w) {
        wmax = any ? fmax(wmax, w) : w;
        any = true;
      }
      if (a.relax != 1.0) u[c] = __builtin_fma(a.relax, u[c] - o, o);   // damped update (extension; after the test)
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) dst[((size_t)q * NC + c) * B + b] = u[c];
  }  // q
  // partial maxima go to memory (one coalesced store per block row) and are reduced by k_fbs_advance: no atomics
  if (a.metric) a.metric[(size_t)blockIdx.y * B + b] = any ? wmax : -1.0;
}

// The error-point mode of k_control_pts for points SORTED by interval (fb_sweep's linspace points are): a thread takes
// kCtlRun consecutive intervals of one instance, walks them with one sliding window per row (the cubic of an interval once, every
// point of the interval from it) and accumulates the weighted change over its points.  QS[k] .. QS[k+1]: the points of interval k.
// Two points per interval (the reference's 1001 points on 500 steps) cost the point-by-point kernel 8 loads and 6 divisions per
// row and point: 200 us per sweep at batch 16384 against 30-80 for every other kernel of the sweep.
constexpr int kCtlRun = 4;
template <class P>
__global__ __launch_bounds__(256) void k_control_pts_sorted(const ControlPtsArgs a, const int* __restrict__ QS) {
  constexpr int NS = P::NS, NC = P::NC, NTU = P::NTU, R = kCtlRun;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (a.gate && *a.gate == 0) return;
  if (b >= a.batch) return;
  const int n = a.T.n, i0 = blockIdx.y * R;
  const size_t B = (size_t)a.batch;
  double wmax = 0.0;
  bool any = false;
  if (i0 <= n - 2) {
    const PchipTabU TU_(a.T);
    const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
    double lb[NC], ub[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      lb[c] = a.lb[c];
      ub[c] = a.ub[c];
    }
    PchipSlide sl[NS], sx[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      sl[k].init(TU_, a.lam + (size_t)k * B + b, (size_t)NS * B, i0);
      if (P::CC_READS_X) sx[k].init(TU_, a.x + (size_t)k * B + b, (size_t)a.ldx * B, i0);
    }
    const int iend = i0 + R < n - 1 ? i0 + R : n - 1;
#pragma unroll 1
    for (int i = i0; i < iend; ++i) {
      double lv[NS], ld[NS], l2[NS], l3[NS], xv[NS], xd[NS], x2[NS], x3[NS];
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const double* lp = a.lam + (size_t)k * B + b;
        sl[k].coef(TU_, lp, (size_t)NS * B, i, PchipSlide::next(TU_, lp, (size_t)NS * B, i), lv[k], ld[k], l2[k], l3[k]);
        if (P::CC_READS_X) {
          const double* xp = a.x + (size_t)k * B + b;
          sx[k].coef(TU_, xp, (size_t)a.ldx * B, i, PchipSlide::next(TU_, xp, (size_t)a.ldx * B, i), xv[k], xd[k], x2[k], x3[k]);
        }
      }
      const int q1 = QS[i + 1];
      for (int q = QS[i]; q < q1; ++q) {   // (wave-uniform)
        const double sq = a.SQ[q];
        double x[NS], lam[NS], tu[NTU], u[NC];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          lam[k] = lv[k] + sq * (ld[k] + sq * (l2[k] + sq * l3[k]));
          x[k] = P::CC_READS_X ? xv[k] + sq * (xd[k] + sq * (x2[k] + sq * x3[k])) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NTU; ++k) tu[k] = as_uniform(a.TUQ)[(size_t)q * NTU + k];
        P::control_char(tu, x, lam, p, lb, ub, u);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          double* dst = a.out + ((size_t)q * NC + c) * B + b;
          if (a.metric) {   // fb_sweep.m:107, as k_control_pts
            const double o = *dst;
            const double w = fabs(u[c] - o) / (a.relTol * fabs(o) + a.absTol);
            if (w == w) {
              wmax = any ? fmax(wmax, w) : w;
              any = true;
            }
            if (a.relax != 1.0) u[c] = __builtin_fma(a.relax, u[c] - o, o);
          }
          *dst = u[c];
        }
      }
    }
  }
  if (a.metric) a.metric[(size_t)blockIdx.y * B + b] = any ? wmax : -1.0;
}

// ControlChar-side time coefficients at arbitrary times
template <class P>
__global__ void k_tu_at(int nq, const double* __restrict__ tq, const double* __restrict__ ps, double* __restrict__ TUQ) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  double tc[P::NTC], tu[P::NTU];
  P::tcoef(tq[q], ps, tc, tu);
#pragma unroll
  for (int k = 0; k < P::NTU; ++k) TUQ[(size_t)q * P::NTU + k] = tu[k];
}

// ---------------------------------------------------------------------------------------
// check_convergence (fb_sweep.m:99-115) + loop bookkeeping (:79-87), one thread per instance.
// status: 0 active, k > 0 converged at sweep k.  An active instance whose change is <= 1 keeps
// its OLD control (final_sweep(u), :82) and freezes; otherwise it switches to the new buffer.
// ---------------------------------------------------------------------------------------
__global__ void k_fbs_advance(int batch, int sweep, int nparts, const double* __restrict__ metric,
                              int* __restrict__ anyvalid, int* __restrict__ usel, int* __restrict__ status,
                              double* __restrict__ maxChange, int* __restrict__ nactive, int ldb,
                              const int* __restrict__ gate) {
  if (gate && *gate == 0) return;  // a sweep enqueued ahead of the verdict of the one before, which was the last
  const size_t ld = (size_t)(ldb ? ldb : batch);  // row distance of metric / maxChange
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  bool still = false;
  if (b < batch) {
    (void)anyvalid;
    double mx = -1.0;  // max over the valid (non-NaN) weighted changes, NaN if there is none (:108)
    int q = 0;
    for (; q + 32 <= nparts; q += 32) {  // 32 independent loads in flight: the kernel is a few waves of pure latency
      double m[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) m[j] = metric[(size_t)(q + j) * ld + b];
#pragma unroll
      for (int j = 0; j < 32; ++j) mx = fmax(mx, m[j]);
    }
    for (; q + 8 <= nparts; q += 8) {
      double m[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = metric[(size_t)(q + j) * ld + b];
#pragma unroll
      for (int j = 0; j < 8; ++j) mx = fmax(mx, m[j]);
    }
    for (; q < nparts; ++q) mx = fmax(mx, metric[(size_t)q * ld + b]);
    if (mx < 0.0) mx = __builtin_nan("");
    if (status[b] == 0) {
      maxChange[(size_t)(sweep - 1) * ld + b] = mx;  // the value :109 prints
      if (mx <= 1.0) {                                   // :110
        status[b] = sweep;
      } else {
        usel[b] = 1 - usel[b];                           // u = uNew  :85
        still = true;
      }
    }
  }
  const unsigned long long m = __ballot(still);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(nactive, __popcll(m));
}


}  // namespace ocs
