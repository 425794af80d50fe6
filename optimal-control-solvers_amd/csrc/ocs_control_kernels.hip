// ocs_control_kernels.hip -- control parametrisations on the device.
//
// Reference: every Control class is a fixed basis matrix B (nBasis x nT) with
//   u = reshape(v, nC, []) * B          (PWLinearControl.m:59-62 and twins)
//   dJdv = reshape(dJdu * B', [], 1)    (PWLinearControl.m:53-56 and twins)
// computed as dense products although B is ~98 % zeros for the piecewise bases.  B is shared
// by the whole batch, so it is kept as a wave-uniform sparse matrix (CSC for u, CSR for dJdv);
// each (time point, trajectory) / (basis function, trajectory) pair is one thread, terms are
// accumulated in ascending index order = the order of the reference's dense dot product
// (the skipped terms are exact zeros).  Batch-minor layouts: v [nBasis][nC][B], u [nT][nC][B].
#include "ocs_device_common.hpp"
#include "ocs_internal.hpp"

namespace ocs {

static inline int hip_rc2(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// u[j][c][b] = sum_{k in col j} v[row[k]][c][b] * val[k]
__global__ __launch_bounds__(256) void k_basis_expand(int nT, int nC, int batch, const int* __restrict__ colptr,
                                                      const int* __restrict__ row, const double* __restrict__ val,
                                                      const double* __restrict__ v, double* __restrict__ u) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (b >= batch || j >= nT) return;
  const size_t B = (size_t)batch;
  const int k0 = colptr[j], k1 = colptr[j + 1];
  for (int c = 0; c < nC; ++c) {
    double acc = 0.0;
    for (int k = k0; k < k1; ++k) acc += v[((size_t)row[k] * nC + c) * B + b] * val[k];
    u[((size_t)j * nC + c) * B + b] = acc;
  }
}

// dJdv[i][c][b] = sum_{k in row i} dJdu[col[k]][c][b] * val[k]
__global__ __launch_bounds__(256) void k_basis_contract(int nBasis, int nC, int batch, const int* __restrict__ rowptr,
                                                        const int* __restrict__ col, const double* __restrict__ val,
                                                        const double* __restrict__ dJdu, double* __restrict__ dJdv) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (b >= batch || i >= nBasis) return;
  const size_t B = (size_t)batch;
  const int k0 = rowptr[i], k1 = rowptr[i + 1];
  for (int c = 0; c < nC; ++c) {
    double acc = 0.0;
    for (int k = k0; k < k1; ++k) acc += dJdu[((size_t)col[k] * nC + c) * B + b] * val[k];
    dJdv[((size_t)i * nC + c) * B + b] = acc;
  }
}

// Dense bases (Chebyshev), time-parallel.  B is shared by the batch: the basis column of a grid point is a wave-uniform row of
// the transposed table BT [nT][NB] (scalar loads).
//   expand    u(:, j) = sum_i v_i B(i, j): a thread takes kExpandTJ consecutive grid points of one trajectory with the NB
//             coefficients in registers; the sum over i in ascending order = the reference's dense product, bit for bit.
//   contract  dJdv_i = sum_j dJdu(:, j) B(i, j): the grid points are cut into S segments, one wave per segment and 64 trajectories
//             accumulates its NB partial sums over its segment (ascending j), the segments are added in ascending order through
//             LDS: the reference's sum over j with S - 1 of its additions re-associated (round-off level; S = 8 up to 16 basis
//             functions, else 4).
// Until round 4 both were ONE thread per trajectory walking all grid points serially: 95 / 565 us at batch 4096 x 2001 points
// with 16 functions where the data take 15 / 15 us (the objective of a four-state problem with a Chebyshev control spent 0.9 of
// its 1.08 ms here).
constexpr int kExpandTJ = 16;
template <int NB>
__global__ __launch_bounds__(64) void k_basis_expand_dense(int nT, int nC, int batch, const double* __restrict__ BT,
                                                           const double* __restrict__ v, double* __restrict__ u) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.z;
  if (b >= batch) return;
  const size_t B = (size_t)batch;
  const uniform_ptr BTu = as_uniform(BT);
  double vv[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) vv[i] = v[((size_t)i * nC + c) * B + b];
  const int j0 = blockIdx.y * kExpandTJ, j1 = j0 + kExpandTJ < nT ? j0 + kExpandTJ : nT;
  for (int j = j0; j < j1; ++j) {
    const uniform_ptr bt = BTu + (size_t)j * NB;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < NB; ++i) acc += vv[i] * bt[i];
    u[((size_t)j * nC + c) * B + b] = acc;
  }
}
template <int NB, int S>
__global__ __launch_bounds__(64 * S) void k_basis_contract_dense(int nT, int nC, int batch, const double* __restrict__ BT,
                                                                 const double* __restrict__ dJdu, double* __restrict__ dJdv) {
  // a wave walks its segment in tiles of TJ grid points: the TJ rows of BT go through the wave's own corner of LDS (one
  // coalesced load; scalar loads of a row per grid point cost their latency every iteration), the TJ samples of dJdu are loaded
  // together, then TJ x NB multiply-adds with the rows read back as LDS broadcasts.  The partial sums take the same LDS afterwards.
  constexpr int TJ = 16, PER = (TJ * NB + 63) / 64;
  static_assert(TJ * NB <= NB * 64, "staging area inside the wave's partial-sum area");
  __shared__ double part[S][NB][64];
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int b0 = blockIdx.x * 64 + lane;
  const int b = b0 < batch ? b0 : batch - 1;
  const int c = blockIdx.y;
  const size_t B = (size_t)batch;
  const int len = (nT + S - 1) / S, j0 = seg * len, j1 = j0 + len < nT ? j0 + len : nT;
  double* stage = &part[seg][0][0];
  double acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = 0.0;
  const double* dp = dJdu + (size_t)c * B + b;
  for (int t0 = j0; t0 < j1; t0 += TJ) {
    double dd[TJ];
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
      const int j = t0 + jj;
      dd[jj] = j < j1 ? dp[(size_t)j * nC * B] : 0.0;   // (past the segment: a zero sample on a valid row of BT)
    }
    __builtin_amdgcn_wave_barrier();   // (the reads of the tile before are done: wave-local LDS, no workgroup barrier)
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = q * 64 + lane;
      if (e < TJ * NB) {
        const int j = t0 + e / NB;
        stage[e] = BT[(size_t)(j < nT ? j : nT - 1) * NB + e % NB];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
#pragma unroll
      for (int i = 0; i < NB; ++i) acc[i] += dd[jj] * stage[jj * NB + i];
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < NB; ++i) part[seg][i][lane] = acc[i];
  __syncthreads();
  // wave `seg` adds the segments of the functions seg, seg + S, ... in ascending order of the segments
  for (int i = seg; i < NB; i += S) {
    double sum = part[0][i][lane];
#pragma unroll
    for (int q = 1; q < S; ++q) sum += part[q][i][lane];
    if (b0 < batch) dJdv[((size_t)i * nC + c) * B + b0] = sum;
  }
}

template <int NB>
static void run_dense(bool expand, int nT, int nC, int batch, const double* BT, const double* in, double* out,
                      hipStream_t s) {
  constexpr int S = NB <= 16 ? 8 : 4;
  if (expand)
    k_basis_expand_dense<NB><<<dim3((batch + 63) / 64, (nT + kExpandTJ - 1) / kExpandTJ, nC), dim3(64), 0, s>>>(nT, nC, batch, BT, in, out);
  else
    k_basis_contract_dense<NB, S><<<dim3((batch + 63) / 64, nC), dim3(64 * S), 0, s>>>(nT, nC, batch, BT, in, out);
}
bool basis_dense_supported(int nBasis) { return nBasis >= 1 && nBasis <= 32; }
// BT: transposed basis [nT][nBasis].  expand: in = v, out = u; else in = dJdu, out = dJdv.
int launch_basis_dense(bool expand, int nBasis, int nT, int nC, int batch, const double* BT, const double* in,
                       double* out, hipStream_t s) {
  switch (nBasis) {
#define OCS_NB(n) case n: run_dense<n>(expand, nT, nC, batch, BT, in, out, s); break;
    OCS_NB(1) OCS_NB(2) OCS_NB(3) OCS_NB(4) OCS_NB(5) OCS_NB(6) OCS_NB(7) OCS_NB(8)
    OCS_NB(9) OCS_NB(10) OCS_NB(11) OCS_NB(12) OCS_NB(13) OCS_NB(14) OCS_NB(15) OCS_NB(16)
    OCS_NB(17) OCS_NB(18) OCS_NB(19) OCS_NB(20) OCS_NB(21) OCS_NB(22) OCS_NB(23) OCS_NB(24)
    OCS_NB(25) OCS_NB(26) OCS_NB(27) OCS_NB(28) OCS_NB(29) OCS_NB(30) OCS_NB(31) OCS_NB(32)
#undef OCS_NB
    default: return -1;
  }
  return hip_rc2(hipGetLastError());
}

// dst[r][b] = src[idx[r]][b]   (gathers rows of a batch-minor array; used for lam(FreeInitStates,1))
__global__ void k_gather_rows(int nrows, int batch, const int* __restrict__ idx, const double* __restrict__ src,
                              double* __restrict__ dst) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (b >= batch || r >= nrows) return;
  dst[(size_t)r * batch + b] = src[(size_t)idx[r] * batch + b];
}
// dst[idx[r]][b] = src[r][b]
__global__ void k_scatter_rows(int nrows, int batch, const int* __restrict__ idx, const double* __restrict__ src,
                               double* __restrict__ dst) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (b >= batch || r >= nrows) return;
  dst[(size_t)idx[r] * batch + b] = src[(size_t)r * batch + b];
}

// out[j][c][b] = val[c]  (u0 = ControlBounds(:,1)*ones(1,length(t)), fb_sweep.m:23)
__global__ void k_add_vec(int n, const double* __restrict__ a, const double* __restrict__ b, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}
__global__ void k_fill_rows(int ncols, int nC, int batch, const double* __restrict__ val, double* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (b >= batch || j >= ncols) return;
  for (int c = 0; c < nC; ++c) out[((size_t)j * nC + c) * batch + b] = val[c];
}
int launch_add_vec(int n, const double* a, const double* b, double* out, hipStream_t s) {
  k_add_vec<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(n, a, b, out);
  return hip_rc2(hipGetLastError());
}
int launch_fill_rows(int ncols, int nC, int batch, const double* val, double* out, hipStream_t s) {
  k_fill_rows<<<dim3((batch + 255) / 256, ncols), dim3(256), 0, s>>>(ncols, nC, batch, val, out);
  return hip_rc2(hipGetLastError());
}

int launch_basis_expand(int nT, int nC, int batch, const int* colptr, const int* row, const double* val,
                        const double* v, double* u, hipStream_t s) {
  k_basis_expand<<<dim3((batch + 255) / 256, nT), dim3(256), 0, s>>>(nT, nC, batch, colptr, row, val, v, u);
  return hip_rc2(hipGetLastError());
}
int launch_basis_contract(int nBasis, int nC, int batch, const int* rowptr, const int* col, const double* val,
                          const double* dJdu, double* dJdv, hipStream_t s) {
  k_basis_contract<<<dim3((batch + 255) / 256, nBasis), dim3(256), 0, s>>>(nBasis, nC, batch, rowptr, col, val,
                                                                             dJdu, dJdv);
  return hip_rc2(hipGetLastError());
}
int launch_gather_rows(int nrows, int batch, const int* idx, const double* src, double* dst, hipStream_t s) {
  k_gather_rows<<<dim3((batch + 255) / 256, nrows), dim3(256), 0, s>>>(nrows, batch, idx, src, dst);
  return hip_rc2(hipGetLastError());
}
int launch_scatter_rows(int nrows, int batch, const int* idx, const double* src, double* dst, hipStream_t s) {
  k_scatter_rows<<<dim3((batch + 255) / 256, nrows), dim3(256), 0, s>>>(nrows, batch, idx, src, dst);
  return hip_rc2(hipGetLastError());
}

}  // namespace ocs
