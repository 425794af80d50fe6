#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k_axpy(const double* x, double* y, double a, int n){
  int i = blockIdx.x*blockDim.x+threadIdx.x; if(i<n) y[i] = a*x[i]+y[i];
}
// dependent chain of NOPS fma
template<int ILP>
__global__ void k_chain(double* out, long long* cyc, double a, double b, int iters){
  double v[ILP];
  for(int k=0;k<ILP;k++) v[k] = threadIdx.x*1e-3 + k;
  long long t0 = clock64();
  for(int it=0; it<iters; ++it){
    #pragma unroll
    for(int r=0;r<16;r++){
      #pragma unroll
      for(int k=0;k<ILP;k++) v[k] = __builtin_fma(v[k], a, b);
    }
  }
  long long t1 = clock64();
  double s=0; for(int k=0;k<ILP;k++) s+=v[k];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  if(threadIdx.x==0) cyc[blockIdx.x]=t1-t0;
}
extern "C" {
int probe_axpy(const double* x, double* y, double a, int n, void* stream){
  hipLaunchKernelGGL(k_axpy, dim3((n+255)/256), dim3(256), 0, (hipStream_t)stream, x, y, a, n);
  return (int)hipGetLastError();
}
int probe_chain(double* out, long long* cyc, int ilp, int iters, int blocks, int threads, void* stream){
  hipStream_t s=(hipStream_t)stream;
  switch(ilp){
    case 1: hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(threads),0,s,out,cyc,0.999,1e-3,iters); break;
    case 2: hipLaunchKernelGGL(k_chain<2>, dim3(blocks), dim3(threads),0,s,out,cyc,0.999,1e-3,iters); break;
    case 4: hipLaunchKernelGGL(k_chain<4>, dim3(blocks), dim3(threads),0,s,out,cyc,0.999,1e-3,iters); break;
    case 8: hipLaunchKernelGGL(k_chain<8>, dim3(blocks), dim3(threads),0,s,out,cyc,0.999,1e-3,iters); break;
  }
  return (int)hipGetLastError();
}
}
