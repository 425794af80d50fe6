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

// Dense bases (Chebyshev): one thread per trajectory keeps all NB coefficients / accumulators in registers
// and streams over time once, reading the basis column of each grid point as a wave-uniform row of the
// transposed table BT [nT][NB] -- dJdu is then read once instead of once per basis function.
// Accumulation order over j (ascending) and over i (ascending) is the reference's dense product.
template <int NB>
__global__ __launch_bounds__(64) void k_basis_expand_dense(int nT, int nC, int batch, const double* __restrict__ BT,
                                                           const double* __restrict__ v, double* __restrict__ u) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (b >= batch) return;
  const size_t B = (size_t)batch;
  double vv[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) vv[i] = v[((size_t)i * nC + c) * B + b];
  for (int j = 0; j < nT; ++j) {
    const double* bt = BT + (size_t)j * NB;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < NB; ++i) acc += vv[i] * bt[i];
    u[((size_t)j * nC + c) * B + b] = acc;
  }
}
template <int NB>
__global__ __launch_bounds__(64) void k_basis_contract_dense(int nT, int nC, int batch, const double* __restrict__ BT,
                                                             const double* __restrict__ dJdu, double* __restrict__ dJdv) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (b >= batch) return;
  const size_t B = (size_t)batch;
  double acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = 0.0;
  for (int j = 0; j < nT; ++j) {
    const double* bt = BT + (size_t)j * NB;
    const double d = dJdu[((size_t)j * nC + c) * B + b];
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[i] += d * bt[i];
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) dJdv[((size_t)i * nC + c) * B + b] = acc[i];
}

template <int NB>
static void run_dense(bool expand, int nT, int nC, int batch, const double* BT, const double* in, double* out,
                      hipStream_t s) {
  const dim3 grid((batch + 63) / 64, nC), block(64);
  if (expand)
    k_basis_expand_dense<NB><<<grid, block, 0, s>>>(nT, nC, batch, BT, in, out);
  else
    k_basis_contract_dense<NB><<<grid, block, 0, s>>>(nT, nC, batch, BT, in, out);
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
__global__ void k_fill_rows(int ncols, int nC, int batch, const double* __restrict__ val, double* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (b >= batch || j >= ncols) return;
  for (int c = 0; c < nC; ++c) out[((size_t)j * nC + c) * batch + b] = val[c];
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
