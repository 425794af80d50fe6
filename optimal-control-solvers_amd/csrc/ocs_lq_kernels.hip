// ocs_lq_kernels.hip -- gfx950 matrix-core kernels for problems whose dynamics are linear with a
// Jacobian shared by the whole batch (the build-defined LQ problem of BASELINE config 5):
//
//   F(t,[x;.],u) = [ A x + Bu u ;  e^{-rt} ( sum_k q_k x_k^2 + sum_c R_c u_c^2 ) ]
//   dFdx_times_vec = [ A' v_x + 2 e^{-rt} (q .* x) v_last ; 0 ]
//   dFdu_times_vec =   Bu' v_x + 2 e^{-rt} (R .* u) v_last
//
// (the OCProblem plugin contract of OCProblem/OCProblem.m:8-21; same parameter block
// [r | A | Bu | q | rdiag] as oracle/ocs_oracle.c).  The RK4 recursion and its discrete adjoint are
// those of Integrator/RK4Integrator.m:28-56, :59-94, :97-121.
//
// Mapping ("M"): one wave integrates 16 trajectories.  Every stage evaluation A*Y is a
// (16 RT x 16 RT) x (16 RT x 16) product done with v_mfma_f64_16x16x4_f64: A sits in registers as
// MFMA A-operand fragments for the whole kernel (RT * 4RT fragments, one double per lane each), the
// stage state Y is the B operand.  The C/D layout of that instruction (lane (g, n) = (lane>>4, lane&15)
// holds rows g + 4j of column n) is exactly its B-operand layout for k-step j, so the result of one
// stage feeds the next stage's product without leaving the registers: lane (g, n) owns rows
// {4m + g} of trajectory n for the whole pass.  Bu*u is one more k-step (nC <= 4 = K).
// The adjoint pass recomputes Y2..Y4 from the checkpoint y_i with the same instruction sequence
// (bit-identical to the forward pass) and runs the four A'k products on the transposed fragments.
#include "ocs_device_common.hpp"
#include "ocs_internal.hpp"
#include "ocs_rk4_kernels.hpp"

namespace ocs {

static inline int hip_rc_lq(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

typedef double d4 __attribute__((ext_vector_type(4)));

// time coefficients of the LQ problem for the shared table builders (k_tcoef / k_build_rec)
struct LQTime {
  static constexpr int NTC = 1, NTU = 1, NSC = 0;
  __device__ static inline void tcoef(double t, const double* ps, double* tc, double* tu) {
    tc[0] = exp(-ps[0] * t);
    tu[0] = exp(ps[0] * t);
  }
  __device__ static inline void step_consts(double, double, const double*, const double*, const double*, double*) {}
};

int launch_tcoef_lq(const ProblemDesc& p, const GridDesc& g, hipStream_t s) {
  const int nT = 2 * g.N + 1;
  k_tcoef<LQTime><<<dim3((nT + 255) / 256), dim3(256), 0, s>>>(nT, g.T, p.ps, g.TC, g.TU);
  k_build_rec<LQTime><<<dim3((g.N + 2 * kRecPad + 255) / 256), dim3(256), 0, s>>>(g.N, g.HT, g.TC, g.REC);
  return hip_rc_lq(hipGetLastError());
}

struct LQArgs {
  int N, batch, nS, nC;
  const double* REC;
  const double* ps;     // [r | A nS x nS col-major | Bu nS x nC | q nS | rdiag nC]
  const double* x0;     // forward: [nS][B]
  const double* xck;    // backward: checkpoints [N+1][nAug][B]
  const double* u;      // [2N+1][nC][B]; UCONST: [nC]
  double* x;            // forward out [N+1][nAug][B] or null
  double* J;            // forward out [B]
  const double* Jadd;   // optional [B]
  const double* lamT;   // backward: [nAug][B] or null (default e_last, RK4Integrator.m:63-66)
  double* lam;          // [N+1][nAug][B] or null
  double* dJdu;         // [2N+1][nC][B] or null
  double* lam0;         // [nAug][B] or null
};

// D = A(16x4) * B(4x16) + C on one wave; a: lane (g,i) holds A[i][g]; b: lane (g,n) holds B[g][n];
// c/d: lane (g,n) holds rows g + 4j of column n.
__device__ static inline d4 mma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

template <int RT>
struct LQMat {
  static constexpr int KS = 4 * RT;
  double f[RT][KS];
};

// fragments of M (rows x cols, column-major with leading dimension ld, zero outside) as the A operand:
// tile rt, k-step kk: lane (g,i) <- M[16 rt + i][4 kk + g];  TRANS reads M' instead.
template <int RT, bool TRANS>
__device__ static inline void load_frags(LQMat<RT>& o, const double* M, int ld, int rows, int cols, int g, int i) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int kk = 0; kk < 4 * RT; ++kk) {
      const int r = 16 * rt + i, c = 4 * kk + g;
      const int rr = TRANS ? c : r, cc = TRANS ? r : c;
      o.f[rt][kk] = (rr < rows && cc < cols) ? M[rr + (size_t)ld * cc] : 0.0;
    }
}

// acc (tile rt, reg j) <-> row 4 (4 rt + j) + g, i.e. per-lane value index m = 4 rt + j
template <int RT>
__device__ static inline void matvec(const LQMat<RT>& A, const double (&v)[4 * RT], d4 (&acc)[RT]) {
  if constexpr (RT == 1) {  // a single tile: two independent accumulation chains instead of one dependent one
    d4 alt = {0.0, 0.0, 0.0, 0.0};
    acc[0] = mma(A.f[0][0], v[0], acc[0]);
    alt = mma(A.f[0][1], v[1], alt);
    acc[0] = mma(A.f[0][2], v[2], acc[0]);
    alt = mma(A.f[0][3], v[3], alt);
    acc[0] += alt;
  } else {
#pragma unroll
    for (int kk = 0; kk < 4 * RT; ++kk)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = mma(A.f[rt][kk], v[kk], acc[rt]);
  }
}

template <int RT>
__device__ static inline void unpack(const d4 (&acc)[RT], double (&f)[4 * RT]) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    f[4 * rt + 0] = acc[rt].x;
    f[4 * rt + 1] = acc[rt].y;
    f[4 * rt + 2] = acc[rt].z;
    f[4 * rt + 3] = acc[rt].w;
  }
}

// sum over the four lanes (g = 0..3) that share a trajectory
__device__ static inline double sum_over_g(double v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

template <int RT>
struct LQCore {
  static constexpr int KS = 4 * RT;
  LQMat<RT> A;
  double Bu[RT];   // A-operand fragments of Bu (nS x nC, K = 4 >= nC): lane (g,i) <- Bu[16 rt + i][g]
  double q[KS];    // q[4m + g]
  double R;        // rdiag[g] (0 for g >= nC)

  __device__ inline void load(const double* ps, int nS, int nC, int g, int i) {
    load_frags<RT, false>(A, ps + 1, nS, nS, nS, g, i);
    const double* bu = ps + 1 + (size_t)nS * nS;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int r = 16 * rt + i;
      Bu[rt] = (r < nS && g < nC) ? bu[r + (size_t)nS * g] : 0.0;
    }
    const double* qq = bu + (size_t)nS * nC;
#pragma unroll
    for (int m = 0; m < KS; ++m) q[m] = (4 * m + g < nS) ? qq[4 * m + g] : 0.0;
    R = (g < nC) ? qq[nS + g] : 0.0;
  }
  // Bu * u for 16 trajectories (lane (g,n) holds u_g of trajectory n)
  __device__ inline void bu_times(double u, d4 (&o)[RT]) const {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const d4 z = {0.0, 0.0, 0.0, 0.0};
      o[rt] = mma(Bu[rt], u, z);
    }
  }
  // state rows of F: A Y + Bu u (bu = Bu u precomputed)
  __device__ inline void Fx(const double (&Y)[KS], const d4 (&bu)[RT], double (&f)[KS]) const {
    d4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = bu[rt];
    matvec<RT>(A, Y, acc);
    unpack<RT>(acc, f);
  }
  // this lane's share of the objective integrand e^{-rt}(sum q x^2 + sum R u^2)
  __device__ inline double cost_part(const double (&Y)[KS], double u, double e) const {
    double s = R * (u * u);
#pragma unroll
    for (int m = 0; m < KS; ++m) s = __builtin_fma(q[m], Y[m] * Y[m], s);
    return e * s;
  }
};

// ---------------------------------------------------------------------------------------
// forward pass   RK4Integrator.m:28-56
// ---------------------------------------------------------------------------------------
template <int RT, bool OUT_X, bool UCONST>
__global__ __launch_bounds__(64) void k_lq_forward(const LQArgs a) {
  constexpr int KS = 4 * RT;
  using Rec = StepRec<1>;
  const int lane = threadIdx.x, g = lane >> 4, n = lane & 15;
  const int b0 = blockIdx.x * 16 + n;
  const int b = b0 < a.batch ? b0 : a.batch - 1;  // lanes past the batch recompute the last trajectory
  const size_t B = (size_t)a.batch;
  const int nS = a.nS, nC = a.nC, N = a.N;
  const size_t nAugB = (size_t)(nS + 1) * B;

  LQCore<RT> P;
  P.load(a.ps, nS, nC, g, n);

  double y[KS], yc = 0.0;  // xK(:,1,1) = [x0; 0]   :33
#pragma unroll
  for (int m = 0; m < KS; ++m) y[m] = (4 * m + g < nS) ? a.x0[(size_t)(4 * m + g) * B + b] : 0.0;

  double* xo = a.x + (size_t)g * B + b;
  auto store_x = [&]() OCS_INLINE {
    if (!OUT_X) return;
#pragma unroll
    for (int m = 0; m < KS; ++m)
      if (4 * m + g < nS) xo[(size_t)(4 * m) * B] = y[m];
    if (g == 0) xo[(size_t)nS * B] = yc;
    xo += nAugB;
  };
  store_x();

  const bool uact = g < nC;
  const size_t ustride = (size_t)nC * B;
  const double* up = a.u + (size_t)(uact ? g : 0) * B + b;  // u(:,1) of this lane's control row
  double uA = UCONST ? (uact ? a.u[g] : 0.0) : (uact ? *up : 0.0);
  d4 buA[RT], buM[RT], buB[RT];
  P.bu_times(uA, buA);
  if (UCONST) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) buM[rt] = buB[rt] = buA[rt];
  }

  const double* recp = a.REC;
  Rec cur = load_rec<1>(recp);
  double uM = uA, uB = uA;
  if (!UCONST) {
    uM = uact ? up[ustride] : 0.0;
    uB = uact ? up[2 * ustride] : 0.0;
  }
  for (int i = 0; i < N; ++i) {
    // next step's uniform record and control samples are requested now and consumed a step later
    recp += rec_stride(1);
    const Rec nxt = load_rec<1>(recp);  // the table is padded past step N-1
    double uMn = uM, uBn = uB;
    if (!UCONST) {
      const int in = i + 1 < N ? i + 1 : i;
      const double* q = up + (size_t)(2 * in) * ustride;
      uMn = uact ? q[ustride] : 0.0;
      uBn = uact ? q[2 * ustride] : 0.0;
      P.bu_times(uM, buM);
      P.bu_times(uB, buB);
    }
    double F1[KS], F2[KS], F3[KS], F4[KS], Y[KS];
    P.Fx(y, buA, F1);                                                         // :37
    double cs = P.cost_part(y, uA, cur.tcA[0]);
#pragma unroll
    for (int m = 0; m < KS; ++m) Y[m] = __builtin_fma(cur.hh, F1[m], y[m]);   // :40
    P.Fx(Y, buM, F2);                                                         // :41
    cs += 2.0 * P.cost_part(Y, uM, cur.tcM[0]);
#pragma unroll
    for (int m = 0; m < KS; ++m) Y[m] = __builtin_fma(cur.hh, F2[m], y[m]);   // :44
    P.Fx(Y, buM, F3);                                                         // :45
    cs += 2.0 * P.cost_part(Y, uM, cur.tcM[0]);
#pragma unroll
    for (int m = 0; m < KS; ++m) Y[m] = __builtin_fma(cur.h, F3[m], y[m]);    // :48
    P.Fx(Y, buB, F4);                                                         // :49
    cs += P.cost_part(Y, uB, cur.tcB[0]);
#pragma unroll
    for (int m = 0; m < KS; ++m)                                              // :50
      y[m] = __builtin_fma(cur.h6, (F1[m] + 2.0 * F2[m]) + (2.0 * F3[m] + F4[m]), y[m]);
    yc = __builtin_fma(cur.h6, sum_over_g(cs), yc);
    store_x();
    cur = nxt;
    uA = uB;
    uM = uMn;
    uB = uBn;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) buA[rt] = buB[rt];
  }
  if (g == 0) a.J[b] = a.Jadd ? a.Jadd[b] + yc : yc;  // J = x(end,end)   :55
}

// ---------------------------------------------------------------------------------------
// adjoint pass   RK4Integrator.m:59-121
// ---------------------------------------------------------------------------------------
template <int RT, bool OUT_LAM, bool OUT_DJDU, bool UCONST>
__global__ __launch_bounds__(64) void k_lq_backward(const LQArgs a) {
  constexpr int KS = 4 * RT;
  using Rec = StepRec<1>;
  const int lane = threadIdx.x, g = lane >> 4, n = lane & 15;
  const int b0 = blockIdx.x * 16 + n;
  const int b = b0 < a.batch ? b0 : a.batch - 1;
  const size_t B = (size_t)a.batch;
  const int nS = a.nS, nC = a.nC, N = a.N;
  const size_t nAugB = (size_t)(nS + 1) * B;

  LQCore<RT> P;
  P.load(a.ps, nS, nC, g, n);
  LQMat<RT> AT;  // fragments of A'
  load_frags<RT, true>(AT, a.ps + 1, nS, nS, nS, g, n);
  double BuT[KS];  // A-operand fragments of Bu' (nC x nS, rows padded to 16): lane (g,i) <- Bu[4 kk + g][i]
  {
    const double* bu = a.ps + 1 + (size_t)nS * nS;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) BuT[kk] = (4 * kk + g < nS && n < nC) ? bu[(4 * kk + g) + (size_t)nS * n] : 0.0;
  }
  // (Bu' v)_g of trajectory n lands in register 0 of lane (g, n)
  auto but_times = [&](const double (&v)[KS]) OCS_INLINE {
    d4 acc = {0.0, 0.0, 0.0, 0.0}, alt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < KS; kk += 2) {
      acc = mma(BuT[kk], v[kk], acc);
      alt = mma(BuT[kk + 1], v[kk + 1], alt);
    }
    return acc.x + alt.x;
  };
  // A' k + 2 e (q .* Y) k_last
  auto ATx = [&](const double (&k)[KS], const double (&Y)[KS], double e2kl, double (&gout)[KS]) OCS_INLINE {
    d4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      acc[rt].x = P.q[4 * rt + 0] * Y[4 * rt + 0] * e2kl;
      acc[rt].y = P.q[4 * rt + 1] * Y[4 * rt + 1] * e2kl;
      acc[rt].z = P.q[4 * rt + 2] * Y[4 * rt + 2] * e2kl;
      acc[rt].w = P.q[4 * rt + 3] * Y[4 * rt + 3] * e2kl;
    }
    matvec<RT>(AT, k, acc);
    unpack<RT>(acc, gout);
  };

  double lam[KS], lamc;  // lam(:,end) = lamT   :69; the last row of dFdx_times_vec is 0, so lam(end,:) is constant
#pragma unroll
  for (int m = 0; m < KS; ++m) lam[m] = (a.lamT && 4 * m + g < nS) ? a.lamT[(size_t)(4 * m + g) * B + b] : 0.0;
  lamc = a.lamT ? a.lamT[(size_t)nS * B + b] : 1.0;

  double* lo = a.lam + (size_t)N * nAugB + (size_t)g * B + b;
  auto store_lam = [&]() OCS_INLINE {
    if (!OUT_LAM) return;
#pragma unroll
    for (int m = 0; m < KS; ++m)
      if (4 * m + g < nS) lo[(size_t)(4 * m) * B] = lam[m];
    if (g == 0) lo[(size_t)nS * B] = lamc;
    lo -= nAugB;
  };
  store_lam();

  const bool uact = g < nC;
  const size_t ustride = (size_t)nC * B;
  const size_t uoff = (size_t)(uact ? g : 0) * B + b;
  const double* up = a.u + uoff;
  double* dq = a.dJdu + uoff;
  double uB = UCONST ? (uact ? a.u[g] : 0.0) : (uact ? up[(size_t)(2 * N) * ustride] : 0.0);
  double uA = uB, uM = uB;
  d4 buA[RT], buM[RT];
  P.bu_times(uB, buA);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) buM[rt] = buA[rt];

  double k1c[KS], k1lc = 0.0;  // k1 of step i+1 (the other half of the node column of dJdu, :108-112)
#pragma unroll
  for (int m = 0; m < KS; ++m) k1c[m] = 0.0;

  const double* recp = a.REC + (size_t)(N - 1) * rec_stride(1);
  const double* ckp = a.xck + (size_t)(N - 1) * nAugB + (size_t)g * B + b;
  Rec cur = load_rec<1>(recp);
  double y[KS];
#pragma unroll
  for (int m = 0; m < KS; ++m) y[m] = (4 * m + g < nS) ? ckp[(size_t)(4 * m) * B] : 0.0;
  if (!UCONST) {
    uA = uact ? up[(size_t)(2 * N - 2) * ustride] : 0.0;
    uM = uact ? up[(size_t)(2 * N - 1) * ustride] : 0.0;
  }
  double eA0 = cur.tcA[0];

  for (int i = N - 1; i >= 0; --i) {
    // requests for step i-1 (consumed at the end of this iteration)
    recp -= rec_stride(1);
    const Rec nxt = load_rec<1>(recp);  // the table is padded before step 0
    const int ip = i > 0 ? i - 1 : 0;
    const double* cq = a.xck + (size_t)ip * nAugB + (size_t)g * B + b;
    double yn[KS];
#pragma unroll
    for (int m = 0; m < KS; ++m) yn[m] = (4 * m + g < nS) ? cq[(size_t)(4 * m) * B] : 0.0;
    double uAn = uA, uMn = uM;
    if (!UCONST) {
      uAn = uact ? up[(size_t)(2 * ip) * ustride] : 0.0;
      uMn = uact ? up[(size_t)(2 * ip + 1) * ustride] : 0.0;
      P.bu_times(uA, buA);
      P.bu_times(uM, buM);
    }
    // recompute the stage states (same instruction sequence as the forward pass)
    double F[KS], Y2[KS], Y3[KS], Y4[KS];
    P.Fx(y, buA, F);
#pragma unroll
    for (int m = 0; m < KS; ++m) Y2[m] = __builtin_fma(cur.hh, F[m], y[m]);
    P.Fx(Y2, buM, F);
#pragma unroll
    for (int m = 0; m < KS; ++m) Y3[m] = __builtin_fma(cur.hh, F[m], y[m]);
    P.Fx(Y3, buM, F);
#pragma unroll
    for (int m = 0; m < KS; ++m) Y4[m] = __builtin_fma(cur.h, F[m], y[m]);

    double k4[KS], k3[KS], k2[KS], k1[KS], g3[KS], g2[KS], g1[KS], g0[KS];
    const double k4l = cur.h6 * lamc, k3l = cur.h3 * lamc, k2l = k3l, k1l = k4l;
#pragma unroll
    for (int m = 0; m < KS; ++m) k4[m] = cur.h6 * lam[m];                              // :73
    ATx(k4, Y4, 2.0 * cur.tcB[0] * k4l, g3);                                            // :74
#pragma unroll
    for (int m = 0; m < KS; ++m) k3[m] = __builtin_fma(cur.h, g3[m], cur.h3 * lam[m]);  // :77
    ATx(k3, Y3, 2.0 * cur.tcM[0] * k3l, g2);                                            // :78
#pragma unroll
    for (int m = 0; m < KS; ++m) k2[m] = __builtin_fma(cur.hh, g2[m], cur.h3 * lam[m]); // :81
    ATx(k2, Y2, 2.0 * cur.tcM[0] * k2l, g1);                                            // :82
#pragma unroll
    for (int m = 0; m < KS; ++m) k1[m] = __builtin_fma(cur.hh, g1[m], cur.h6 * lam[m]); // :85
    ATx(k1, y, 2.0 * cur.tcA[0] * k1l, g0);                                             // :86-88

    if (OUT_DJDU) {
      // dFdu_times_vec is linear in v and does not read y for this problem, so the two terms of a
      // column (:104-106 midpoint, :108-112 node) are one product of the summed multipliers
      double v[KS];
#pragma unroll
      for (int m = 0; m < KS; ++m) v[m] = k4[m] + k1c[m];
      const double dn = but_times(v) + 2.0 * cur.tcB[0] * P.R * uB * (k4l + k1lc);
#pragma unroll
      for (int m = 0; m < KS; ++m) v[m] = k2[m] + k3[m];
      const double dm = but_times(v) + 2.0 * cur.tcM[0] * P.R * uM * (k2l + k3l);
      if (uact) {
        dq[(size_t)(2 * i + 2) * ustride] = dn;
        dq[(size_t)(2 * i + 1) * ustride] = dm;
      }
#pragma unroll
      for (int m = 0; m < KS; ++m) k1c[m] = k1[m];
      k1lc = k1l;
    }
#pragma unroll
    for (int m = 0; m < KS; ++m) lam[m] = (((lam[m] + g1[m]) + g2[m]) + g3[m]) + g0[m];  // :86-88
    store_lam();

    eA0 = cur.tcA[0];
    cur = nxt;
    uB = uA;
    uA = uAn;
    uM = uMn;
#pragma unroll
    for (int m = 0; m < KS; ++m) y[m] = yn[m];
  }
  if (OUT_DJDU) {  // first column: B(t_1, y_1, u_1)' k1_1   :100-101   (uB now holds u(:,1))
    const double d0 = but_times(k1c) + 2.0 * eA0 * P.R * uB * k1lc;
    if (uact) dq[0] = d0;
  }
  if (a.lam0) {
#pragma unroll
    for (int m = 0; m < KS; ++m)
      if (4 * m + g < nS) a.lam0[(size_t)(4 * m + g) * B + b] = lam[m];
    if (g == 0) a.lam0[(size_t)nS * B + b] = lamc;
  }
}

// ---------------------------------------------------------------------------------------
// plugin evaluation (ocs_problem_F / dFdx_times_vec / dFdu_times_vec): one thread per column
// ---------------------------------------------------------------------------------------
__global__ void k_lq_eval(int which, int k, int nS, int nC, const double* __restrict__ t,
                          const double* __restrict__ y, const double* __restrict__ u, const double* __restrict__ v,
                          const double* __restrict__ ps, double* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  const int nAug = nS + 1;
  const double *A = ps + 1, *Bu = A + (size_t)nS * nS, *q = Bu + (size_t)nS * nC, *R = q + nS;
  const double* yj = y + (size_t)j * nAug;
  const double* uj = u + (size_t)j * nC;
  const double e = exp(-ps[0] * t[j]);
  if (which == 0) {
    double* o = out + (size_t)j * nAug;
    double s = 0.0;
    for (int i = 0; i < nS; ++i) {
      double acc = 0.0;
      for (int l = 0; l < nS; ++l) acc += A[i + (size_t)l * nS] * yj[l];
      for (int l = 0; l < nC; ++l) acc += Bu[i + (size_t)l * nS] * uj[l];
      o[i] = acc;
      s += q[i] * (yj[i] * yj[i]);
    }
    for (int l = 0; l < nC; ++l) s += R[l] * (uj[l] * uj[l]);
    o[nS] = e * s;
  } else if (which == 1) {
    const double* vj = v + (size_t)j * nAug;
    double* o = out + (size_t)j * nAug;
    for (int i = 0; i < nS; ++i) {
      double acc = 0.0;
      for (int l = 0; l < nS; ++l) acc += A[l + (size_t)i * nS] * vj[l];
      o[i] = acc + 2 * e * q[i] * yj[i] * vj[nS];
    }
    o[nS] = 0.0;
  } else {
    const double* vj = v + (size_t)j * nAug;
    double* o = out + (size_t)j * nC;
    for (int l = 0; l < nC; ++l) {
      double acc = 0.0;
      for (int i = 0; i < nS; ++i) acc += Bu[i + (size_t)l * nS] * vj[i];
      o[l] = acc + 2 * e * R[l] * uj[l] * vj[nS];
    }
  }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
bool lq_supported(int nS, int nC) { return nS >= 1 && nS <= 32 && nC >= 1 && nC <= 4; }

template <int RT>
static void run_lq_forward(const LQArgs& a, bool uconst, hipStream_t s) {
  const dim3 grid((a.batch + 15) / 16), block(64);
  if (uconst)
    k_lq_forward<RT, true, true><<<grid, block, 0, s>>>(a);
  else if (a.x)
    k_lq_forward<RT, true, false><<<grid, block, 0, s>>>(a);
  else
    k_lq_forward<RT, false, false><<<grid, block, 0, s>>>(a);
}
int launch_forward_lq(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                      double* x, double* J, const FwdOpts& o, hipStream_t s) {
  if (!lq_supported(p.nS, p.nC) || p.pmask || o.usel || (o.uconst && !x)) return -1;
  LQArgs a{};
  a.N = g.N; a.batch = batch; a.nS = p.nS; a.nC = p.nC; a.REC = g.REC; a.ps = p.ps;
  a.x0 = x0; a.u = u; a.x = x; a.J = J; a.Jadd = o.Jadd;
  if (p.nS <= 16) run_lq_forward<1>(a, o.uconst, s); else run_lq_forward<2>(a, o.uconst, s);
  return hip_rc_lq(hipGetLastError());
}

template <int RT>
static void run_lq_backward(const LQArgs& a, bool uconst, hipStream_t s) {
  const dim3 grid((a.batch + 15) / 16), block(64);
  if (uconst)
    k_lq_backward<RT, false, false, true><<<grid, block, 0, s>>>(a);
  else if (a.lam && a.dJdu)
    k_lq_backward<RT, true, true, false><<<grid, block, 0, s>>>(a);
  else if (a.lam)
    k_lq_backward<RT, true, false, false><<<grid, block, 0, s>>>(a);
  else
    k_lq_backward<RT, false, true, false><<<grid, block, 0, s>>>(a);
}
int launch_backward_lq(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                       const double* lamT, double* lam, double* dJdu, const BwdOpts& o, hipStream_t s) {
  if (!lq_supported(p.nS, p.nC) || p.pmask || o.usel) return -1;
  if (o.uconst ? (lam || dJdu || !o.lam0) : (!lam && !dJdu)) return -1;
  LQArgs a{};
  a.N = g.N; a.batch = batch; a.nS = p.nS; a.nC = p.nC; a.REC = g.REC; a.ps = p.ps;
  a.xck = xck; a.u = u; a.lamT = lamT; a.lam = lam; a.dJdu = dJdu; a.lam0 = o.lam0;
  if (p.nS <= 16) run_lq_backward<1>(a, o.uconst, s); else run_lq_backward<2>(a, o.uconst, s);
  return hip_rc_lq(hipGetLastError());
}

int launch_eval_lq(const ProblemDesc& p, int which, int k, const double* t, const double* y, const double* u,
                   const double* v, double* out, hipStream_t s) {
  k_lq_eval<<<dim3((k + 127) / 128), dim3(128), 0, s>>>(which, k, p.nS, p.nC, t, y, u, v, p.ps, out);
  return hip_rc_lq(hipGetLastError());
}

}  // namespace ocs
