// ocs_fbs_device.hpp -- kernel templates of the grid forward-backward sweep (pchip coupling, costate
// RK4, control update, convergence bookkeeping).  Shared by ocs_fbs_kernels.hip (built-in problems) and the
// hipRTC path for user problems; device code only.
#pragma once
#include "ocs_device_common.hpp"

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

// Node tables (uniform): TN[0..n-1] node times, HN[0..n-2] spacings, W1/W2[1..n-2] slope weights.
struct PchipTab {
  int n;
  const double* TN;
  const double* HN;
  const double* W1;
  const double* W2;
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
// interpolant in interval k at local coordinate s = q - TN[k]
__device__ static inline double pchip_eval(const PchipTab& T, const double* V, size_t ldB, int k, double s) {
  const double v0 = V[(size_t)k * ldB], v1 = V[(size_t)(k + 1) * ldB];
  const double h = T.HN[k];
  const double dk = pchip_slope_at(T, V, ldB, k), dk1 = pchip_slope_at(T, V, ldB, k + 1);
  const double del = (v1 - v0) / h;
  const double dzzdx = (del - dk) / h, dzdxdx = (dk1 - del) / h;
  const double c3 = (dzdxdx - dzzdx) / h, c2 = 2.0 * dzzdx - dzdxdx;
  return v0 + s * (dk + s * (c2 + s * c3));
}

// midpoint samples: out[i][r][b] = pchip(V(r,:))(t_mid_i)   for rows r < nrows of V [n][ld][B]
__global__ __launch_bounds__(256) void k_pchip_mid(PchipTab T, int nrows, int ld, int batch,
                                                   const double* __restrict__ TM, const double* __restrict__ V,
                                                   double* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (b >= batch || i >= T.n - 1) return;
  const size_t B = (size_t)batch;
  const double s = TM[i] - T.TN[i];
  for (int r = 0; r < nrows; ++r)
    out[((size_t)i * nrows + r) * B + b] = pchip_eval(T, V + (size_t)r * B + b, (size_t)ld * B, i, s);
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
  const int* usel;
  long long udelta;
  double* lam;         // [N+1][nS][B]
};

template <class P, int PF>
__global__ __launch_bounds__(64) void k_costate(const CostateArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTC = P::NTC;
  using Rec = StepRec<NTC>;
  const int b0 = blockIdx.x * 64 + threadIdx.x;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int N = a.N;
  const uniform_ptr PS = as_uniform(a.ps);
  const typename P::Par p = P::load(ParamSrc{PS, a.pb, a.pmask, B, b});
  const double warm = warm_table(a.REC, (size_t)N * rec_stride(NTC));

  const double* xp = a.x + b;
  const double* mp = a.xmid + b;
  const double* up = a.u + b;
  if (a.usel) up += (long long)a.usel[b] * a.udelta;
  double* lp = a.lam + b;

  double l[NS], xB[NS], uB[NC];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    l[k] = 0.0;  // lam0 = 0*x0   compute_x_lam.m:4
    xB[k] = xp[((size_t)N * a.ldx + k) * B];
    lp[((size_t)N * NS + k) * B] = 0.0;
  }
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
      lp[((size_t)i * NS + k) * B] = l[k];
      xB[k] = cxA[k];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) uB[c] = cuA[c];
  };
  auto next_rec = [&]() OCS_INLINE {
    const Rec r = rq[0];
#pragma unroll
    for (int q = 0; q + 1 < PF; ++q) rq[q] = rq[q + 1];
    rq[PF - 1] = load_rec<NTC>(recp);
    recp -= rec_stride(NTC);
    return r;
  };
  constexpr int CH = 8;
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
  const double* lam;   // [N+1][nS][B]
  const double* lmid;  // [N][nS][B]
  double* u;           // the control grid, updated in place
  const int* status;   // only instances that are still active (status 0) take the new control (fb_sweep.m:85);
                       // a converged instance keeps its old one for the final sweep (:82)
};

template <class P>
__global__ __launch_bounds__(256) void k_control_grid(const ControlGridArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTU = P::NTU;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;  // grid point
  if (b >= a.batch || a.status[b] != 0) return;
  const size_t B = (size_t)a.batch;
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  double x[NS], lam[NS], tu[NTU], lb[NC], ub[NC], u[NC];
  const int i = j >> 1;
  if (j & 1) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      x[k] = a.xmid[((size_t)i * NS + k) * B + b];
      lam[k] = a.lmid[((size_t)i * NS + k) * B + b];
    }
  } else {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      x[k] = a.x[((size_t)i * a.ldx + k) * B + b];
      lam[k] = a.lam[((size_t)i * NS + k) * B + b];
    }
  }
#pragma unroll
  for (int k = 0; k < NTU; ++k) tu[k] = a.TU[(size_t)j * NTU + k];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    lb[c] = a.lb[c];
    ub[c] = a.ub[c];
  }
  P::control_char(tu, x, lam, p, lb, ub, u);
#pragma unroll
  for (int c = 0; c < NC; ++c) a.u[((size_t)j * NC + c) * B + b] = u[c];
}

// ---------------------------------------------------------------------------------------
// control at arbitrary points: out[q][c][b] = ControlChar(tq, x(tq), lam(tq)) with pchip x, lam
// (errorPts fb_sweep.m:107, interpPts :123).  KQ/SQ: interval index and local coordinate of tq.
// If usel is given the result goes to buffer 1 - usel[b] (out + that * odelta).
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
  unsigned long long* metric;
  int* anyvalid;
  double relTol, absTol;
};

constexpr int kPtsPerThread = 8;

template <class P>
__global__ __launch_bounds__(256) void k_control_pts(const ControlPtsArgs a) {
  constexpr int NS = P::NS, NC = P::NC, NTU = P::NTU;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  double wmax = 0.0;
  bool any = false;
  const int q0 = (int)blockIdx.y * kPtsPerThread;
  for (int q = q0; q < q0 + kPtsPerThread && q < a.nq; ++q) {
  const size_t B = (size_t)a.batch;
  const typename P::Par p = P::load(ParamSrc{as_uniform(a.ps), a.pb, a.pmask, B, b});
  const int k0 = a.KQ[q];
  const double s = a.SQ[q];
  double x[NS], lam[NS], tu[NTU], lb[NC], ub[NC], u[NC];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    x[k] = pchip_eval(a.T, a.x + (size_t)k * B + b, (size_t)a.ldx * B, k0, s);
    lam[k] = pchip_eval(a.T, a.lam + (size_t)k * B + b, (size_t)NS * B, k0, s);
  }
#pragma unroll
  for (int k = 0; k < NTU; ++k) tu[k] = a.TUQ[(size_t)q * NTU + k];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    lb[c] = a.lb[c];
    ub[c] = a.ub[c];
  }
  P::control_char(tu, x, lam, p, lb, ub, u);
  double* dst = a.out + (a.usel ? (long long)(1 - a.usel[b]) * a.odelta : 0);
#pragma unroll
  for (int c = 0; c < NC; ++c) dst[((size_t)q * NC + c) * B + b] = u[c];
  if (a.metric) {  // fb_sweep.m:107  abs(uNew - u) ./ (uRelTol*abs(u) + uAbsTol), max() skips NaN (:108)
    const double* old = a.out + (long long)a.usel[b] * a.odelta;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double o = old[((size_t)q * NC + c) * B + b];
      const double w = fabs(u[c] - o) / (a.relTol * fabs(o) + a.absTol);
      if (w == w) {
        wmax = any ? fmax(wmax, w) : w;
        any = true;
      }
    }
  }
  }  // q
  if (a.metric && any) {
    atomicMax(&a.metric[b], (unsigned long long)__double_as_longlong(wmax));
    a.anyvalid[b] = 1;
  }
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
__global__ void k_fbs_advance(int batch, int sweep, unsigned long long* __restrict__ metric,
                              int* __restrict__ anyvalid, int* __restrict__ usel, int* __restrict__ status,
                              double* __restrict__ maxChange, int* __restrict__ nactive) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  bool still = false;
  if (b < batch) {
    const double mx = anyvalid[b] ? __longlong_as_double((long long)metric[b]) : __builtin_nan("");
    metric[b] = 0ull;  // reset for the next sweep
    anyvalid[b] = 0;
    if (status[b] == 0) {
      maxChange[(size_t)(sweep - 1) * batch + b] = mx;  // the value :109 prints
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
