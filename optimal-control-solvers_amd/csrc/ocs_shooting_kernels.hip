// ocs_shooting_kernels.hip -- per-instance bookkeeping of the batched single-shooting driver (ocs_shooting.cpp):
// spectral projected gradient on  min_v J_b(v), Lb <= v <= Ub  for B independent instances.  The objective and its
// gradient come from the hot path (nlpObjective, single_shooting.m:137-150); these kernels are the outer iteration
// that fmincon('sqp') performs in the reference (single_shooting.m:114), one thread per instance, every array
// [rows][B] so that a wave reads consecutive instances.
#include "ocs_device_common.hpp"
#include "ocs_internal.hpp"

namespace ocs {

static inline int hip_rc6(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

__device__ static inline double spg_proj(double w, double lb, double ub) { return fmin(fmax(w, lb), ub); }

// first evaluation done: step length 1 / ||P(v - g) - v||_inf, objective history filled with J
__global__ __launch_bounds__(256) void k_spg_init(SpgArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  const size_t B = (size_t)a.batch;
  double pgn = 0.0;
  for (int i = 0; i < a.nV; ++i) {
    const double v = a.v[i * B + b], g = a.g[i * B + b];
    pgn = fmax(pgn, fabs(spg_proj(v - g, a.lb[i], a.ub[i]) - v));
  }
  a.alpha[b] = 1.0 / fmax(pgn, 1e-12);
  for (int m = 0; m < a.memory; ++m) a.hist[m * B + b] = a.J[b];
  a.active[b] = 1;
  a.iters[b] = 0;
}

// stopping test, search direction d = P(v - alpha g) - v, g'd, the non-monotone reference value, first trial point
__global__ __launch_bounds__(256) void k_spg_direction(SpgArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  const size_t B = (size_t)a.batch;
  int act = a.active[b];
  if (act) {
    double pgn = 0.0;
    for (int i = 0; i < a.nV; ++i) {
      const double v = a.v[i * B + b], g = a.g[i * B + b];
      pgn = fmax(pgn, fabs(spg_proj(v - g, a.lb[i], a.ub[i]) - v));
    }
    if (!(pgn > a.TolFun)) act = 0;
    a.active[b] = act;
  }
  if (!act) {  // finished: its trial point stays its solution, so that the batch evaluation leaves it alone
    a.accepted[b] = 1;
    for (int i = 0; i < a.nV; ++i) a.vt[i * B + b] = a.v[i * B + b];
    return;
  }
  const double alpha = a.alpha[b];
  double gtd = 0.0;
  for (int i = 0; i < a.nV; ++i) {
    const double v = a.v[i * B + b], g = a.g[i * B + b];
    const double d = spg_proj(v - alpha * g, a.lb[i], a.ub[i]) - v;
    a.d[i * B + b] = d;
    a.vt[i * B + b] = v + d;
    gtd += g * d;
  }
  double fmx = a.hist[b];
  for (int m = 1; m < a.memory; ++m) fmx = fmax(fmx, a.hist[m * B + b]);
  a.gtd[b] = gtd;
  a.fmax[b] = fmx;
  a.lam[b] = 1.0;
  a.accepted[b] = 0;
  atomicAdd(a.counter, 1);
}

// Armijo test of the trial point against the non-monotone reference; a rejected instance halves its step and gets
// its next trial point
__global__ __launch_bounds__(256) void k_spg_accept(SpgArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch || a.accepted[b]) return;
  const size_t B = (size_t)a.batch;
  const double lam = a.lam[b];
  if (a.Jt[b] <= a.fmax[b] + 1e-4 * lam * a.gtd[b]) {
    a.accepted[b] = 2;  // accepted in this iteration
    return;
  }
  const double half = 0.5 * lam;
  a.lam[b] = half;
  for (int i = 0; i < a.nV; ++i) a.vt[i * B + b] = a.v[i * B + b] + half * a.d[i * B + b];
  atomicAdd(a.counter, 1);
}

// take the accepted point, Barzilai-Borwein step length for the next iteration, history, stopping on a small step
__global__ __launch_bounds__(256) void k_spg_update(SpgArgs a, int it) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  const size_t B = (size_t)a.batch;
  if (a.active[b]) {
    a.iters[b] += 1;
    if (a.accepted[b] == 2) {
      double sty = 0.0, sts = 0.0, smax = 0.0;
      for (int i = 0; i < a.nV; ++i) {
        const double vn = a.vt[i * B + b], gn = a.gt[i * B + b];
        const double s = vn - a.v[i * B + b], y = gn - a.g[i * B + b];
        sty += s * y;
        sts += s * s;
        smax = fmax(smax, fabs(s));
        a.v[i * B + b] = vn;
        a.g[i * B + b] = gn;
      }
      a.J[b] = a.Jt[b];
      const double an = sty > 0.0 ? sts / fmax(sty, 1e-300) : 1e3;
      a.alpha[b] = fmin(fmax(an, 1e-10), 1e10);
      if (smax <= a.TolX) a.active[b] = 0;
    } else {
      a.active[b] = 0;  // the line search failed: the instance stops where it is
    }
  }
  a.hist[(size_t)(it % a.memory) * B + b] = a.J[b];
}

// ||P(v - g) - v||_inf of the result and the verdict
__global__ __launch_bounds__(256) void k_spg_finish(SpgArgs a, double* pgnorm, int* converged) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  const size_t B = (size_t)a.batch;
  double pgn = 0.0;
  for (int i = 0; i < a.nV; ++i) {
    const double v = a.v[i * B + b], g = a.g[i * B + b];
    pgn = fmax(pgn, fabs(spg_proj(v - g, a.lb[i], a.ub[i]) - v));
  }
  if (pgnorm) pgnorm[b] = pgn;
  if (converged) converged[b] = pgn <= 10.0 * a.TolFun;
}

int launch_spg(int which, const SpgArgs& a, int it, double* pgnorm, int* converged, hipStream_t s) {
  const dim3 grid((a.batch + 255) / 256), block(256);
  switch (which) {
    case 0: k_spg_init<<<grid, block, 0, s>>>(a); break;
    case 1: k_spg_direction<<<grid, block, 0, s>>>(a); break;
    case 2: k_spg_accept<<<grid, block, 0, s>>>(a); break;
    case 3: k_spg_update<<<grid, block, 0, s>>>(a, it); break;
    case 4: k_spg_finish<<<grid, block, 0, s>>>(a, pgnorm, converged); break;
    default: return -1;
  }
  return hip_rc6(hipGetLastError());
}

}  // namespace ocs
