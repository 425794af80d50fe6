#include <hip/hip_runtime.h>
typedef const double __attribute__((address_space(4))) * cdp;
__device__ inline cdp as_uniform(const double* p) { return (cdp)p; }
__global__ void k(const double* H, const double* x, double* y, int n, int B) {
  int b = blockIdx.x*64+threadIdx.x;
  cdp h = as_uniform(H);
  double acc = x[b];
  for (int i=0;i<n;++i){ acc = __builtin_fma(acc, h[2*i], h[2*i+1]); y[(size_t)i*B+b]=acc; }
}
