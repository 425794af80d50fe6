// ocs_fold_kernels.hip -- launcher of the state pass that forms its control from the costate of the sweep before
// (ocs_fold_kernel.hpp); registry problems whose ControlChar does not read x.
#include "ocs_fold_kernel.hpp"
#include "ocs_internal.hpp"
#include "ocs_jit.hpp"
#include "ocs_problems.hpp"
#include <cstdio>

namespace ocs {

static inline int hip_rc7(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

bool fold_supported(const ProblemDesc& p, const GridDesc& g, int batch) {
  if (p.nC != 1 || !(p.nS == 1 || p.nS == 2 || p.nS == 4)) return false;
  if (p.functor == Functor::User)   // hipRTC instances; the costate pass is the scan kernel (costate_scan_ok)
    return user_fold(p.user) && g.N >= 8 && g.N % 8 == 0 && tile_ok(batch, 64 / p.nS) && costate_scan_ok(p, g, batch) && g.TU && g.REC;
  if (p.functor != Functor::Logistic) return false;
  return g.N >= 8 && g.N % 8 == 0 && tile_ok(batch, 64 / p.nS) && costate_forms_midpoints(p, g.N, batch) && g.TU && g.REC;
}

template <class P>
static void run_forward_cc(const FwdArgsCC& a, bool uniform, hipStream_t s) {
  using C_ = FoldCfg<P::NS>;
  const dim3 grid(tile_count(a.batch, C_::TPW)), block(C_::NWAVE * 64);
  if (uniform)
    k_forward_cc<P, true><<<grid, block, 0, s>>>(a);
  else
    k_forward_cc<P, false><<<grid, block, 0, s>>>(a);
}

int launch_forward_cc(const ProblemDesc& p, const GridDesc& g, int batch, const double* PR, const double* lb,
                      const double* ub, const double* x0, const double* lam, double* x, double* J, const int* frozen,
                      bool no_cost_row, const int* gate, bool first_sweep, hipStream_t s) {
  if (!fold_supported(p, g, batch) || !PR || !lam || !x || !J) return -1;
  const FwdArgsCC a{g.N, batch, g.REC, PR, g.TU, p.ps, p.pb, p.pmask, lb, ub, x0, lam, x, J, frozen, no_cost_row ? 1 : 0, gate, first_sweep ? 1 : 0};
  if (p.functor == Functor::User) {
    const int nwave = p.nS == 1 ? FoldCfg<1>::NWAVE : (p.nS == 2 ? FoldCfg<2>::NWAVE : FoldCfg<4>::NWAVE);
    void* args[] = {(void*)&a};
    return jit_launch(p.user, g.uniform ? UK_FWD_CC_UNI : UK_FWD_CC, dim3(tile_count(batch, 64 / p.nS)), dim3(nwave * 64), args, s);
  }
  if (p.nS == 1)
    run_forward_cc<LogisticK<1>>(a, g.uniform, s);
  else if (p.nS == 2)
    run_forward_cc<LogisticK<2>>(a, g.uniform, s);
  else
    run_forward_cc<LogisticK<4>>(a, g.uniform, s);
#ifdef OCS_P2_STAMPS
  {
    static int calls = 0;
    if (++calls % 8 == 0) {
      (void)hipStreamSynchronize(s);
      long long h[64];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_p2_stamp), sizeof(h));
      fprintf(stderr, "[fold fwd nS=%d] cycles (barrier wait / total):", p.nS);
      for (int w = 0; w < 16; ++w) fprintf(stderr, " w%d %lld/%lld", w, h[4 * w], h[4 * w + 1]);
      static long long hw[4096];
      (void)hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_p2_wg), sizeof(hw));
      fprintf(stderr, "\n   workgroup 0: %lld cycles in %lld ticks of 10 ns = %.3f GHz\n", hw[1] - hw[0], hw[3],
              (double)(hw[1] - hw[0]) / (10.0 * hw[3]));
    }
  }
#endif
  return hip_rc7(hipGetLastError());
}

}  // namespace ocs
