// copy_rates.hip -- what a read + write stream reaches on this chip (1 GiB in, 1 GiB out), by access width, store policy,
// grid shape and the number of loads a lane issues before its first store (reads and writes in longer bursts).
//   hipcc --offload-arch=gfx950 -O3 -o copy_rates copy_rates.hip && ./copy_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double dbl2 __attribute__((ext_vector_type(2)));

template <class T, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy(const T* __restrict__ a, T* __restrict__ b, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    T v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = NTL ? __builtin_nontemporal_load(a + i + k * stride) : a[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (NTS) __builtin_nontemporal_store(v[k], b + i + k * stride); else b[i + k * stride] = v[k];
    }
  }
  for (; i < n; i += stride) b[i] = a[i];
}
template <class T, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ a, double* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    T v = NTL ? __builtin_nontemporal_load(a + i) : a[i];
    if constexpr (sizeof(T) == 16) s += v.x + v.y; else s += v;
  }
  if (s == 12345.678) out[0] = s;
}
template <class T, bool NTS>
__global__ __launch_bounds__(256) void k_fill(T* __restrict__ b, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  T v; if constexpr (sizeof(T) == 16) { v.x = 1; v.y = 2; } else v = 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (NTS) __builtin_nontemporal_store(v, b + i); else b[i] = v;
  }
}

template <class F>
double timeit(F f, int K = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int k = 0; k < 3; ++k) f();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int k = 0; k < K; ++k) f();
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e-3 / K;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  double *a, *b, *o; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  const size_t n8 = bytes / 8, n16 = bytes / 16;
  for (int grid : {1024, 2048, 4096, 8192, 65536}) {
#define RUN(T, U, NTL, NTS, n) { double t = timeit([&] { k_copy<T, U, NTL, NTS><<<grid, 256>>>((const T*)a, (T*)b, n); }); \
    printf("copy %2zu B/lane, %d loads ahead, nt load %d, nt store %d, grid %5d: %5.2f TB/s (read + write)\n", sizeof(T), U, NTL, NTS, grid, 2.0 * bytes / t / 1e12); fflush(stdout); }
    RUN(double, 1, false, false, n8) RUN(double, 4, false, true, n8)
    RUN(dbl2, 1, false, false, n16) RUN(dbl2, 1, false, true, n16) RUN(dbl2, 1, true, true, n16)
    RUN(dbl2, 4, false, true, n16) RUN(dbl2, 8, false, true, n16) RUN(dbl2, 8, true, true, n16) RUN(dbl2, 16, false, true, n16)
  }
  for (int grid : {2048, 8192}) {
    double t = timeit([&] { k_read<dbl2, 1, false, false><<<grid, 256>>>((const dbl2*)a, o, n16); });
    printf("read  16 B/lane grid %5d: %5.2f TB/s\n", grid, bytes / t / 1e12);
    t = timeit([&] { k_read<dbl2, 1, true, false><<<grid, 256>>>((const dbl2*)a, o, n16); });
    printf("read  16 B/lane nt grid %5d: %5.2f TB/s\n", grid, bytes / t / 1e12);
    t = timeit([&] { k_fill<dbl2, false><<<grid, 256>>>((dbl2*)b, n16); });
    printf("fill  16 B/lane grid %5d: %5.2f TB/s\n", grid, bytes / t / 1e12);
    t = timeit([&] { k_fill<dbl2, true><<<grid, 256>>>((dbl2*)b, n16); });
    printf("fill  16 B/lane nt grid %5d: %5.2f TB/s\n", grid, bytes / t / 1e12);
  }
  return 0;
}
