// Probe: fp64 VALU issue rate of one SIMD of MI355X against the number of resident waves.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_rate fp64_rate.hip && ./fp64_rate
// Grid = one workgroup of 256 NW threads per CU (NW waves per SIMD); every wave runs `iters` iterations of 64 fp64 FMAs in
// C independent chains (C = 1: one dependent chain).  Reported: shader cycles per iteration per wave (s_memtime), the
// cycles one SIMD spends per wave-instruction, and the chip rate from the wall clock of the whole launch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int C, bool F32>
__global__ void k_rate(long long* out, double* sink, int iters) {
  double b = 1.0 + threadIdx.x * 1e-9, a = threadIdx.x * 1e-6;
  double f[8];
  float g[8], bf = (float)b, af = (float)a;
  for (int k = 0; k < 8; ++k) { f[k] = a + k; g[k] = (float)f[k]; }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      if (F32) g[k % C] = __builtin_fmaf(g[k % C], bf, af);
      else f[k % C] = __builtin_fma(f[k % C], b, a);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int k = 0; k < 8; ++k) s += f[k] + g[k];
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int C, bool F32>
static int run(int nw, long long* d_out, double* d_sink) {
  const int iters = 4000, nblk = 256;
  for (int rep = 0; rep < 2; ++rep) {
    k_rate<C, F32><<<nblk, 256 * nw>>>(d_out, d_sink, iters);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
  }
  const auto w0 = std::chrono::steady_clock::now();
  k_rate<C, F32><<<nblk, 256 * nw>>>(d_out, d_sink, iters);
  CHECK(hipDeviceSynchronize());
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
  std::vector<long long> h(nblk);
  CHECK(hipMemcpy(h.data(), d_out, sizeof(long long) * nblk, hipMemcpyDeviceToHost));
  double cyc = 0;
  for (auto v : h) cyc += (double)v;
  cyc /= nblk * (double)iters;
  const double flops = 2.0 * 64 * 64 * (double)iters * nblk * 4 * nw;
  printf("%s chains %d, waves/SIMD %d: %8.1f cycles per 64 FMAs per wave = %5.2f SIMD cycles per wave-instruction; launch %.1f TFLOP/s\n",
         F32 ? "f32" : "f64", C, nw, cyc, cyc / 64 / nw, flops / wall / 1e12);
  return 0;
}
int main() {
  long long* d_out; double* d_sink;
  CHECK(hipMalloc(&d_out, sizeof(long long) * 256));
  CHECK(hipMalloc(&d_sink, sizeof(double) * 256 * 1024));
  for (int nw = 1; nw <= 4; ++nw) if (run<1, false>(nw, d_out, d_sink)) return 1;
  for (int nw = 1; nw <= 4; ++nw) if (run<8, false>(nw, d_out, d_sink)) return 1;
  for (int nw = 1; nw <= 4; nw *= 2) if (run<8, true>(nw, d_out, d_sink)) return 1;
  return 0;
}
