// Probe: does v_mfma_f64_16x16x4_f64 run beside fp64 VALU work on one SIMD of MI355X (gfx950)?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
// One wave per SIMD (256-thread workgroup, 1 per CU, a few CUs).  Each loop iteration issues M matrix instructions
// (independent accumulators) and K fp64 FMAs; cycles per iteration by s_memtime.
//   same wave, K independent FMAs    : overlap -> max(64 M, 4 K), shared datapath -> 64 M + 4 K
//   same wave, K FMAs in ONE dependent chain
//   two waves on one SIMD (512 threads): wave A only MFMAs, wave B only a dependent FMA chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int M, int K, bool DEP>
__global__ __launch_bounds__(256) void k_same(long long* out, double* sink, int iters) {
  d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  double f[8];
  for (int k = 0; k < 8; ++k) f[k] = a + k;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (DEP)
        f[0] = __builtin_fma(f[0], b, a);
      else
        f[k & 7] = __builtin_fma(f[k & 7], b, a);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int k = 0; k < 8; ++k) s += f[k];
  for (int m = 0; m < 4; ++m) s += acc[m].x + acc[m].y + acc[m].z + acc[m].w;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// 512 threads: waves 0-3 (one per SIMD) issue MFMAs only, waves 4-7 a dependent fp64 chain only
template <int M, int K>
__global__ __launch_bounds__(512) void k_pair(long long* out, double* sink, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6, f = a;
  const bool do_mfma = wave < 4 && (mode & 1), do_valu = wave >= 4 && (mode & 2);
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < K; ++k) f = __builtin_fma(f, b, a);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = f;
  for (int m = 0; m < 4; ++m) s += acc[m].x + acc[m].y + acc[m].z + acc[m].w;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

// NW waves per SIMD, each a dependent chain of K fp64 FMAs per iteration (DPP: v_fmac_f64_dpp row_newbcast)
template <int NW, int K, bool DPP>
__global__ __launch_bounds__(256 * NW) void k_multi(long long* out, double* sink, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6, f = a;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (DPP) {
#pragma unroll
      for (int k = 0; k < K; ++k)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(f) : "v"(b), "v"(a));
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) f = __builtin_fma(f, b, a);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = f;
  if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 8) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

static long long* d_out;
static double* d_sink;
template <class F>
static void run(const char* name, F launch, int iters, int nwaves) {
  launch();
  hipDeviceSynchronize();
  launch();
  hipDeviceSynchronize();
  std::vector<long long> h(8 * 8);
  hipMemcpy(h.data(), d_out, sizeof(long long) * 64, hipMemcpyDeviceToHost);
  printf("%-58s", name);
  for (int w = 0; w < nwaves; ++w) printf(" %7.1f", (double)h[w] / iters);
  printf("   cycles/iter per wave\n");
}
int main() {
  hipMalloc(&d_out, sizeof(long long) * 8 * 64);
  hipMalloc(&d_sink, sizeof(double) * 1024 * 64);
  const int it = 20000;
#define SAME(M, K, DEP) run("same wave: " #M " mfma + " #K " fma " #DEP, [&] { k_same<M, K, DEP><<<8, 256>>>(d_out, d_sink, it); }, it, 4)
  SAME(0, 16, false); SAME(0, 16, true); SAME(1, 0, false); SAME(2, 0, false); SAME(4, 0, false);
  SAME(1, 8, false); SAME(1, 16, false); SAME(1, 32, false); SAME(1, 8, true); SAME(1, 16, true);
  SAME(2, 16, false); SAME(2, 32, false); SAME(2, 16, true);
#define PAIR(M, K, MODE) run("two waves/SIMD: " #M " mfma | " #K " dep fma, mode " #MODE, [&] { k_pair<M, K><<<8, 512>>>(d_out, d_sink, it, MODE); }, it, 8)
  PAIR(4, 16, 1); PAIR(4, 16, 2); PAIR(4, 16, 3); PAIR(1, 16, 3);
#define MULTI(NW, K, DPP) run("waves/SIMD " #NW ": dependent chain of " #K " fma, dpp " #DPP, [&] { k_multi<NW, K, DPP><<<8, 256 * NW>>>(d_out, d_sink, it); }, it, 4)
  MULTI(1, 16, false); MULTI(2, 16, false); MULTI(3, 16, false); MULTI(4, 16, false);
  MULTI(1, 16, true); MULTI(2, 16, true); MULTI(4, 16, true);
  return 0;
}
