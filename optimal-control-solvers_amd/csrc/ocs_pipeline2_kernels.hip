// ocs_pipeline2_kernels.hip -- launchers of the state-pass kernel k_forward_p2 (ocs_pipeline2_kernel.hpp) for the
// registry problems and the dispatch to the hipRTC instances of user problems given as row functions.
#include "ocs_pipeline2_kernel.hpp"
#include "ocs_pipelinev_kernel.hpp"
#include "ocs_internal.hpp"
#include "ocs_jit.hpp"
#include "ocs_problems.hpp"
#include <cstdio>
#include <cstdlib>

namespace ocs {

static inline int hip_rc7(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// ---------------------------------------------------------------------------------------
bool pipeline_supported(Functor f, int nS, int nC);
bool pipeline_shape_ok(int nS, int N, int batch, bool backward);
bool pipeline_problem_ok(const ProblemDesc& p) {
  if (p.functor == Functor::User) return user_rowsep(p.user) && (p.nS == 1 || p.nS == 2 || p.nS == 4) && p.nC == 1;
  return pipeline_supported(p.functor, p.nS, p.nC);
}

bool vector_problem_ok(const ProblemDesc& p) {
  if (!vector_shape_ok(p.nS, p.nC)) return false;
  if (p.functor == Functor::User) return user_vector(p.user);
  return p.functor == Functor::Logistic;
}
template <class P>
static void run_forward_pv(const FwdArgsP2& a, hipStream_t s) {
  static_assert(PVCfg<P::NS, P::NC>::NWAVE == kPvWaves, "launch shape of the hipRTC instances");
  const dim3 grid(tile_count(a.batch, 64)), block(kPvWaves * 64);
  if (a.x)
    k_forward_pv<P, true><<<grid, block, 0, s>>>(a);
  else
    k_forward_pv<P, false><<<grid, block, 0, s>>>(a);
}
int launch_forward_pv(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u, double* x,
                      double* J, hipStream_t s, bool no_cost_row, const int* gate, const int* frozen) {
  if (!vector_problem_ok(p) || g.N < 8 || g.N % 8 != 0 || batch < 64 || !tile_ok(batch, 64)) return -1;
  const FwdArgsP2 a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, x0, u, x, J, frozen, 0, no_cost_row ? 1 : 0, gate};
  if (p.functor == Functor::User) {
    void* args[] = {(void*)&a};
    return jit_launch(p.user, x ? UK_FWD_PV_X : UK_FWD_PV_J, dim3(tile_count(batch, 64)), dim3(kPvWaves * 64), args, s);
  }
  switch (p.nS) {
    case 1: run_forward_pv<LogisticK<1>>(a, s); break;
    case 2: run_forward_pv<LogisticK<2>>(a, s); break;
    case 3: run_forward_pv<LogisticK<3>>(a, s); break;
    default: run_forward_pv<LogisticK<4>>(a, s); break;
  }
  return hip_rc7(hipGetLastError());
}

template <class P, bool UNI>
static void run_forward_p2u(const FwdArgsP2& a, hipStream_t s) {
  using C_ = P2Cfg<P::NS>;
  const dim3 grid((a.batch + C_::TPW - 1) / C_::TPW), block(C_::NWAVE * 64);   // (a ragged last tile overlaps its neighbour)
  // One workgroup per CU: the recursion wave must have its SIMD to itself.  With less than half of the CU's LDS per
  // workgroup the dispatcher packs two workgroups on one CU while other CUs stay empty (measured: 102 us against
  // 60 us); unused dynamic LDS keeps a second workgroup out.
  const size_t lds_pad = (P::NS == 1) ? 0 : 48 * 1024;
  if (a.frozen) {
    if (a.x)
      k_forward_p2<P, true, true, UNI><<<grid, block, lds_pad, s>>>(a);
    else
      k_forward_p2<P, false, true, UNI><<<grid, block, lds_pad, s>>>(a);
  } else if (a.x) {
    k_forward_p2<P, true, false, UNI><<<grid, block, lds_pad, s>>>(a);
  } else {
    k_forward_p2<P, false, false, UNI><<<grid, block, lds_pad, s>>>(a);
  }
}
template <class P>
static void run_forward_p2(const FwdArgsP2& a, bool uniform, hipStream_t s) {
  if (uniform)
    run_forward_p2u<P, true>(a, s);
  else
    run_forward_p2u<P, false>(a, s);
}

int launch_forward_p2(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                      double* x, double* J, const int* frozen, int ld, hipStream_t s, bool no_cost_row,
                      const int* gate) {
  if (!pipeline_problem_ok(p) || !pipeline_shape_ok(p.nS, g.N, batch, false)) return -1;
  if (batch % (64 / p.nS) != 0 && (ld ? ld : batch) % 2 != 0) return -1;   // (ragged last tile: even row distance only)
  if (p.functor == Functor::User) {   // the hipRTC instances of the same kernel template (generic row functions)
    const FwdArgsP2 au{g.N, batch, g.REC, p.ps, p.pb, p.pmask, x0, u, x, J, frozen, ld, no_cost_row ? 1 : 0, gate};
    static_assert(p2_waves(1) == P2Cfg<1>::NWAVE && p2_waves(2) == P2Cfg<2>::NWAVE && p2_waves(4) == P2Cfg<4>::NWAVE,
                  "launch shape of the hipRTC instances");
    const int TPW = 64 / p.nS, nwave = p2_waves(p.nS);
    void* args[] = {(void*)&au};
    return jit_launch(p.user, x ? UK_FWD_P2_X : UK_FWD_P2_J, dim3((batch + TPW - 1) / TPW), dim3(nwave * 64), args, s,
                      p.nS == 1 ? 0u : 48u * 1024u);
  }
  const FwdArgsP2 a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, x0, u, x, J, frozen, ld, no_cost_row ? 1 : 0, gate};
  if (p.nS == 1)
    run_forward_p2<LogisticK<1>>(a, g.uniform, s);
  else if (p.nS == 2)
    run_forward_p2<LogisticK<2>>(a, g.uniform, s);
  else if (p.nS == 4)
    run_forward_p2<LogisticK<4>>(a, g.uniform, s);
  else
    return -1;
#ifdef OCS_P2_STAMPS
  {
    static int calls = 0;
    if (++calls % 8 == 0) {   // (only these launches are followed by a synchronisation: the others run back to back)
      (void)hipStreamSynchronize(s);
      long long h[32];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_p2_stamp), sizeof(h));
      fprintf(stderr, "[p2 fwd nS=%d] cycles (barrier wait / total):", p.nS);
      for (int w = 0; w < 7; ++w) fprintf(stderr, " w%d %lld/%lld", w, h[4 * w], h[4 * w + 1]);
      fprintf(stderr, "\n");
      static long long hw[4096];
      (void)hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_p2_wg), sizeof(hw));
      const int nwg = batch / (64 / p.nS) < 1024 ? batch / (64 / p.nS) : 1024;
      long long t0 = hw[0], t1 = hw[1], dmin = 1LL << 60, dmax = 0, bsum = 0;
      for (int w = 0; w < nwg; ++w) {
        t0 = hw[4 * w] < t0 ? hw[4 * w] : t0;
        t1 = hw[4 * w + 1] > t1 ? hw[4 * w + 1] : t1;
        const long long d = hw[4 * w + 1] - hw[4 * w];
        dmin = d < dmin ? d : dmin; dmax = d > dmax ? d : dmax; bsum += hw[4 * w + 2];
      }
      {
        long long r0 = 1LL << 62, r1 = 0, e1 = 0; int late = 0;
        for (int w = 0; w < nwg; ++w) { r0 = hw[4 * w + 2] < r0 ? hw[4 * w + 2] : r0; r1 = hw[4 * w + 2] > r1 ? hw[4 * w + 2] : r1; e1 = hw[4 * w + 2] + hw[4 * w + 3] > e1 ? hw[4 * w + 2] + hw[4 * w + 3] : e1; }
        for (int w = 0; w < nwg; ++w) late += (hw[4 * w + 2] - r0) > 1000;
        fprintf(stderr, "   real time (10 ns ticks): last start - first start %lld, last end - first start %lld, workgroups starting > 10 us late: %d\n", r1 - r0, e1 - r0, late);
      }
      fprintf(stderr, "   workgroup 0: %lld cycles in %lld ticks of 10 ns = %.3f GHz; workgroup %d: %.3f GHz\n", hw[1] - hw[0], hw[3],
              (double)(hw[1] - hw[0]) / (10.0 * hw[3]), nwg - 1, (double)(hw[4 * (nwg - 1) + 1] - hw[4 * (nwg - 1)]) / (10.0 * hw[4 * (nwg - 1) + 3]));
      fprintf(stderr, "   S wave over %d workgroups: duration min %lld max %lld, first start -> last end %lld, mean barrier wait %lld\n",
              nwg, dmin, dmax, t1 - t0, bsum / nwg);
    }
  }
#endif
  return hip_rc7(hipGetLastError());
}

}  // namespace ocs
