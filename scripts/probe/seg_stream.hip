// seg_stream.hip -- what HBM delivers for the adjoint scan's access pattern, without its arithmetic.
// Arrays as in the BL-2 adjoint pass: X, LAM [N+1][5][B], U, D [2N+1][B] (B = 4096, N = 1000), 256 workgroups of 16 waves,
// every wave a chunk of 4 consecutive steps of a 64-step superblock, superblocks from the end of the horizon.
// Template G = state rows per wave (the scan kernel: 4, i.e. four 128-byte segments per memory instruction; 1 = one
// 512-byte segment), V = trajectories per lane (2 = 16-byte accesses).  ROT buffer sets used round-robin (3: nothing
// survives in the 256 MiB memory-side cache from one launch to the next).
//   hipcc --offload-arch=gfx950 -O3 -o seg_stream seg_stream.hip && ./seg_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int V> struct Vec;
template <> struct Vec<1> { typedef double T; };
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <> struct Vec<2> { typedef dbl2 T; };
__device__ inline double sum(double a) { return a; }
__device__ inline double sum(dbl2 a) { return a.x + a.y; }
__device__ inline double mk(double s, double) { return s; }
__device__ inline dbl2 mk(double s, dbl2) { dbl2 o; o.x = s; o.y = s + 1.0; return o; }

template <int G, int V, bool NT, bool NTL = false, bool PF = false>
__global__ __launch_bounds__(1024) void k_stream(int N, int B, const double* __restrict__ X, const double* __restrict__ U,
                                                 double* __restrict__ LAM, double* __restrict__ D) {
  typedef typename Vec<V>::T T;
  constexpr int LPR = 64 / G, TPW = LPR * V, RG = 4 / G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x / RG, rg = blockIdx.x % RG;
  const int r = lane / LPR, tl = lane % LPR;
  const int row = rg * G + r;
  const size_t traj = (size_t)tile * TPW + (size_t)tl * V;
  const int nsb = (N + 63) / 64;
  auto ldv = [&](const double* p) -> T { return NTL ? __builtin_nontemporal_load((const T*)p) : *(const T*)p; };
  T xv[4], u0[4], u1[4], xn[4], un0[4], un1[4];
  auto loads = [&](int sb, T* a, T* b, T* c) {
    const int i0 = sb * 64 + wave * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j < N ? i0 + j : N - 1;
      a[j] = ldv(X + ((size_t)i * 5 + row) * B + traj);
      b[j] = ldv(U + (size_t)(2 * i) * B + traj);
      c[j] = ldv(U + (size_t)(2 * i + 1) * B + traj);
    }
  };
  if (PF) loads(nsb - 1, xn, un0, un1);
  for (int sb = nsb - 1; sb >= 0; --sb) {
    const int i0 = sb * 64 + wave * 4;
    if (PF) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { xv[j] = xn[j]; u0[j] = un0[j]; u1[j] = un1[j]; }
      if (sb > 0) loads(sb - 1, xn, un0, un1);
    } else {
      loads(sb, xv, u0, u1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j;
      if (i >= N) continue;
      const double s = sum(xv[j]) * 1.0001 + sum(u0[j]) + sum(u1[j]);
      T o = mk(s, xv[j]);
      T* pl = (T*)(LAM + ((size_t)i * 5 + row) * B + traj);
      if (NT) __builtin_nontemporal_store(o, pl); else *pl = o;
      if (row == 0) {
        T* pc = (T*)(LAM + ((size_t)i * 5 + 4) * B + traj);
        T* d0 = (T*)(D + (size_t)(2 * i) * B + traj);
        T* d1 = (T*)(D + (size_t)(2 * i + 1) * B + traj);
        if (NT) { __builtin_nontemporal_store(o, pc); __builtin_nontemporal_store(o, d0); __builtin_nontemporal_store(o, d1); }
        else { *pc = o; *d0 = o; *d1 = o; }
      }
    }
  }
}

template <int G, int V, bool NT, bool NTL = false, bool PF = false>
void run(const char* name, int N, int B, int ROT, std::vector<double*>& X, std::vector<double*>& U, std::vector<double*>& L,
         std::vector<double*>& D) {
  constexpr int TPW = 64 / G * V, RG = 4 / G;
  const int grid = B / TPW * RG;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int k = 0; k < 30; ++k) k_stream<G, V, NT, NTL, PF><<<grid, 1024>>>(N, B, X[k % ROT], U[k % ROT], L[k % ROT], D[k % ROT]);
  CK(hipDeviceSynchronize());
  const int K = 90;
  CK(hipEventRecord(e0));
  for (int k = 0; k < K; ++k) k_stream<G, V, NT, NTL, PF><<<grid, 1024>>>(N, B, X[k % ROT], U[k % ROT], L[k % ROT], D[k % ROT]);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / K;
  const double bytes = 8.0 * B * ((double)N * (4 + 2 + 5 + 2));   // x rows + u in, lam rows + dJdu out
  printf("%-44s ROT=%d grid %4d: %7.1f us per launch, %5.2f TB/s\n", name, ROT, grid, us, bytes / us / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int N = 1000, B = argc > 1 ? atoi(argv[1]) : 4096;
  for (int ROT : {1, 3}) {
    std::vector<double*> X(ROT), U(ROT), L(ROT), D(ROT);
    const size_t nx = (size_t)(N + 1) * 5 * B, nu = (size_t)(2 * N + 1) * B;
    for (int k = 0; k < ROT; ++k) {
      CK(hipMalloc(&X[k], nx * 8)); CK(hipMalloc(&U[k], nu * 8)); CK(hipMalloc(&L[k], nx * 8)); CK(hipMalloc(&D[k], nu * 8));
      CK(hipMemset(X[k], 0, nx * 8)); CK(hipMemset(U[k], 0, nu * 8));
    }
    run<4, 1, true>("G=4 (4 x 128 B per instruction), nt stores", N, B, ROT, X, U, L, D);
    run<4, 1, false>("G=4, default stores", N, B, ROT, X, U, L, D);
    run<4, 1, true, true>("G=4, nt stores, nt loads", N, B, ROT, X, U, L, D);
    run<4, 1, true, false, true>("G=4, nt stores, loads one superblock ahead", N, B, ROT, X, U, L, D);
    run<4, 1, true, true, true>("G=4, nt stores, nt loads one superblock ahead", N, B, ROT, X, U, L, D);
    run<2, 1, true>("G=2 (2 x 256 B), nt stores", N, B, ROT, X, U, L, D);
    run<2, 1, true, true, true>("G=2, nt stores, nt loads one superblock ahead", N, B, ROT, X, U, L, D);
    run<1, 1, true>("G=1 (512 B), nt stores", N, B, ROT, X, U, L, D);
    run<1, 2, true>("G=1, 16 B per lane (1 KiB), nt stores", N, B, ROT, X, U, L, D);
    run<4, 2, true>("G=4, 16 B per lane (4 x 256 B), nt stores", N, B, ROT, X, U, L, D);
    for (int k = 0; k < ROT; ++k) { CK(hipFree(X[k])); CK(hipFree(U[k])); CK(hipFree(L[k])); CK(hipFree(D[k])); }
  }
  return 0;
}
