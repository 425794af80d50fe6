// micro-probe: cycles per v_mfma_f64_16x16x4_f64 on gfx950 (one wave per SIMD), independent vs dependent chains
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ __launch_bounds__(64) void probe(double* out, long long* cyc, int iters) {
  d4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = d4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS>
void run(int blocks, const char* tag) {
  double* out; long long* cyc;
  hipMalloc(&out, blocks * 64 * 8); hipMalloc(&cyc, blocks * 8);
  const int iters = 2000;
  probe<CHAINS><<<blocks, 64>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<CHAINS><<<blocks, 64>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 8 * CHAINS;
  printf("%s chains=%d blocks=%d: %.1f counter-ticks/MFMA, %.2f ns/MFMA/wave, %.2f TFLOP/s total\n", tag, CHAINS, blocks,
         c0 / n, ms * 1e6 / n, blocks * n * 2048 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1>(1, "dependent");
  run<2>(1, "2-chain");
  run<4>(1, "4-chain");
  run<2>(1024, "2-chain");
  run<4>(1024, "4-chain");
  run<4>(2048, "4-chain");
  run<4>(4096, "4-chain");
  return 0;
}
