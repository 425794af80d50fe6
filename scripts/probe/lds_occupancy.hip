// lds_occupancy.hip -- how many workgroups of a given LDS size and wave count share a CU on this chip: every workgroup spins for a
// fixed number of clock ticks; a launch of 256 k workgroups takes k rounds if one fits per CU, k / 2 if two do, ...
// Also reports hipOccupancyMaxActiveBlocksPerMultiprocessor.
//   hipcc --offload-arch=gfx950 -O3 -o lds_occupancy lds_occupancy.hip && ./lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LDS_DBL, int THREADS>
__global__ __launch_bounds__(THREADS) void k_spin(long long ticks, double* out) {
  __shared__ double buf[LDS_DBL];
  buf[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) s += buf[(threadIdx.x * 7 + (int)s) & 255];
  if (s == 1.2345e300) out[0] = s;
}
template <int LDS_DBL, int THREADS>
static void run(const char* name, double* out) {
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_spin<LDS_DBL, THREADS>, THREADS, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long long ticks = 50 * 100;   // s_memrealtime ticks at 100 MHz: 50 us
  for (int k = 1; k <= 4; ++k) {
    k_spin<LDS_DBL, THREADS><<<256 * k, THREADS>>>(ticks, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_spin<LDS_DBL, THREADS><<<256 * k, THREADS>>>(ticks, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s: LDS %6d B, %4d threads, occupancy API %d blocks/CU; %4d workgroups: %.0f us\n", name, LDS_DBL * 8, THREADS, nb, 256 * k, ms * 1e3);
  }
}
int main() {
  double* out; CK(hipMalloc(&out, 8));
  run<7616, 256>("61 KB, 4 waves", out);
  run<7616, 320>("61 KB, 5 waves", out);
  run<7616, 384>("61 KB, 6 waves (k_forward_p2, four states)", out);
  run<7616, 448>("61 KB, 7 waves", out);
  run<7616, 512>("61 KB, 8 waves", out);
  run<7616, 1024>("61 KB, 16 waves", out);
  run<6144, 384>("48 KB, 6 waves", out);
  run<5120, 384>("40 KB, 6 waves", out);
  run<4096, 384>("32 KB, 6 waves", out);
  run<9600, 256>("75 KB, 4 waves", out);
  run<10240, 512>("80 KB, 8 waves", out);
  return 0;
}
