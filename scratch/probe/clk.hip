#include <hip/hip_runtime.h>
__global__ void k_clk(double* out, long long* res, double a, double b, int iters){
  double v0 = threadIdx.x*1e-3, v1 = v0+1, v2=v0+2, v3=v0+3;
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = __builtin_amdgcn_s_memtime();
  for(int it=0; it<iters; ++it){
    #pragma unroll
    for(int r=0;r<16;r++){ v0=__builtin_fma(v0,a,b); v1=__builtin_fma(v1,a,b); v2=__builtin_fma(v2,a,b); v3=__builtin_fma(v3,a,b);}
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x*blockDim.x+threadIdx.x]=v0+v1+v2+v3;
  if(threadIdx.x==0){ res[2*blockIdx.x]=t1-t0; res[2*blockIdx.x+1]=r1-r0; }
}
extern "C" int probe_clk(double* out, long long* res, int iters, int blocks, int threads, void* stream){
  hipLaunchKernelGGL(k_clk, dim3(blocks), dim3(threads),0,(hipStream_t)stream,out,res,0.999,1e-3,iters);
  return (int)hipGetLastError();
}
