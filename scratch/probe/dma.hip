#include <hip/hip_runtime.h>
// probe: global_load_lds_dwordx4 semantics: each lane gives a global address, LDS dst = uniform base + lane*16
__global__ void k_dma(const double* src, double* out, int n) {
  __shared__ double sh[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) sh[i] = -1.0;
  __syncthreads();
  // lane l loads the pair at src[2*(63-l)], src[2*(63-l)+1]  (reversed order) into LDS[2l], LDS[2l+1]
  const double* g = src + 2 * (63 - lane);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)&sh[0], 16, 0, 0);
  // second DMA with an LDS offset of 1 KiB and only 32 active lanes
  if (lane < 32) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 128 + 2 * lane),
                                     (__attribute__((address_space(3))) void*)&sh[128], 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = sh[i];
}
extern "C" int probe_dma(const double* src, double* out, void* stream) {
  hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 0, (hipStream_t)stream, src, out, 256);
  return (int)hipGetLastError();
}
