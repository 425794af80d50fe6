// Probe: does v_fmac_f64_dpp (row_newbcast) issue at the rate of a plain v_fma_f64 on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o dpp_rate dpp_rate.hip && ./dpp_rate
// One workgroup of 256 NW threads per CU (NW waves per SIMD); every wave runs `iters` iterations of 64 fp64
// multiply-adds into 16 independent accumulators: MODE 0 plain v_fma_f64 (vector operands), MODE 1 v_fmac_f64_dpp
// row_newbcast (the BL-4 lane kernels' basis products), MODE 2 v_fma_f64 with the multiplier in an SGPR pair.
// Reported: chip rate from the wall clock of the whole launch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k_rate(double* sink, int iters, double sc) {
  double b = 1.0 + threadIdx.x * 1e-9, x = 1.0 + threadIdx.x * 1e-7;
  double f[16];
  for (int k = 0; k < 16; ++k) f[k] = k * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) f[k] = __builtin_fma(b, x, f[k]);
      } else if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[k]) : "s"(sc), "v"(x));
      } else {
        asm volatile("s_nop 1\n\t"
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
          : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]),
            "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15])
          : "v"(b), "v"(x));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  double s = 0;
  for (int k = 0; k < 16; ++k) s += f[k];
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static int run(int nw, double* d_sink) {
  const int iters = 4000, nblk = 256;
  for (int rep = 0; rep < 2; ++rep) {
    k_rate<MODE><<<nblk, 256 * nw>>>(d_sink, iters, 1.000001);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
  }
  const auto w0 = std::chrono::steady_clock::now();
  k_rate<MODE><<<nblk, 256 * nw>>>(d_sink, iters, 1.000001);
  CHECK(hipDeviceSynchronize());
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
  const double flops = 2.0 * 64 * 64 * (double)iters * nblk * 4 * nw;
  const char* names[] = {"v_fma_f64 (vector operands)", "v_fmac_f64_dpp row_newbcast", "v_fma_f64 (SGPR multiplier)"};
  printf("%-30s waves/SIMD %d: %.1f TFLOP/s  (%.2f ns per wave-instruction per SIMD)\n", names[MODE], nw, flops / wall / 1e12,
         wall * 1e9 / ((double)iters * 64 * nw));
  return 0;
}
int main() {
  double* d_sink;
  CHECK(hipMalloc(&d_sink, sizeof(double) * 256 * 1024));
  for (int nw = 1; nw <= 4; nw *= 2) {
    if (run<0>(nw, d_sink)) return 1;
    if (run<1>(nw, d_sink)) return 1;
    if (run<2>(nw, d_sink)) return 1;
  }
  return 0;
}
