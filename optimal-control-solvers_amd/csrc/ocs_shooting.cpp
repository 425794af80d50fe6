// ocs_shooting.cpp -- batched direct single shooting on the device: B independent NLPs
//   min_v J_b(v),  Lb <= v <= Ub        (single_shooting.m:69-115; one per column of x0 / per parameter set)
// solved together.  The reference hands one problem to fmincon('sqp') (:114), a MATLAB toolbox; the outer iteration
// here is a spectral projected gradient (Birgin-Martinez-Raydan: Barzilai-Borwein step length, projection on the
// bounds of compute_nlp_bounds, non-monotone Armijo back-tracking) in which every instance keeps its own step
// length, line search and stopping test, and every objective / gradient evaluation is one call of the hot path
// (nlpObjective, :137-150) over the whole batch.  Iterates differ from fmincon's; the KKT point is the same.
#include "ocs_trace.hpp"
#include "ocs_handles.hpp"

#include <limits>
#include <vector>

using namespace ocs;

extern "C" {

int ocs_ss_default_options(ocs_ss_options* o) {
  if (!o) return fail(OCS_ERR_INVALID, "null argument");
  o->TolX = 1e-5;     // single_shooting.m:20
  o->TolFun = 3e-4;   // :21
  o->MaxIter = 500;
  o->memory = 10;
  o->maxBacktracks = 25;
  return OCS_OK;
}

// device: x0 [nS][B] (overwritten at FreeInitStates, :146), v [nV+nFree][B] start -> solution; host: Lb, Ub
// [nV+nFree] or NULL (unbounded); outputs J [B], iterations int[B], converged int[B], pgnorm [B] (any may be NULL
// except J).  Returns OCS_NUM_NOT_CONVERGED if an instance did not reach 10 TolFun.
int ocs_single_shooting_batch_dev(ocs_integrator g, ocs_problem p, ocs_control c, int batch, double* x0, double* v,
                                  int nFree, const int* FreeInitStates, const double* Lb, const double* Ub,
                                  const ocs_ss_options* opt, double* J, int* iterations, int* converged,
                                  double* pgnorm, void* stream) {
  OCS_TRACE("ocs_single_shooting_batch_dev");
  if (!g || !p || !c || !x0 || !v || !opt || !J || batch < 1 || nFree < 0) return fail(OCS_ERR_INVALID, "bad argument");
  if (opt->MaxIter < 0 || opt->memory < 1 || opt->maxBacktracks < 1 || !(opt->TolFun > 0))
    return fail(OCS_ERR_INVALID, "bad options");
  hipStream_t s = (hipStream_t)stream;
  int nB = 0, nC = 0, nT = 0;
  OCS_TRY(ocs_control_dims(c, &nB, &nC, &nT));
  const int nV = nB * nC + nFree;
  const size_t B = (size_t)batch, vb = sizeof(double) * (size_t)nV * B;
  DevBuf gbuf, dbuf, vtbuf, gtbuf, sc, hist, flags, lbub, counter, conv;
  struct Rel {
    DevBuf* b[10];
    ~Rel() {
      for (DevBuf* q : b) q->release();
    }
  } rel{{&gbuf, &dbuf, &vtbuf, &gtbuf, &sc, &hist, &flags, &lbub, &counter, &conv}};
  OCS_TRY(gbuf.ensure(vb));
  OCS_TRY(dbuf.ensure(vb));
  OCS_TRY(vtbuf.ensure(vb));
  OCS_TRY(gtbuf.ensure(vb));
  OCS_TRY(sc.ensure(sizeof(double) * 5 * B));  // Jt, alpha, lam, gtd, fmax
  OCS_TRY(hist.ensure(sizeof(double) * (size_t)opt->memory * B));
  OCS_TRY(flags.ensure(sizeof(int) * 3 * B));  // active, accepted, iters
  OCS_TRY(lbub.ensure(sizeof(double) * 2 * (size_t)nV));
  OCS_TRY(counter.ensure(sizeof(int)));
  {
    std::vector<double> h(2 * (size_t)nV);
    const double inf = std::numeric_limits<double>::infinity();
    for (int i = 0; i < nV; ++i) {
      h[i] = Lb ? Lb[i] : -inf;
      h[nV + i] = Ub ? Ub[i] : inf;
      if (!(h[i] <= h[nV + i])) return fail(OCS_ERR_INVALID, "Lb > Ub at coefficient %d", i);
    }
    HIP_TRY(hipMemcpyAsync(lbub.p, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  int* fl = (int*)flags.p;
  SpgArgs a;
  a.batch = batch;
  a.nV = nV;
  a.memory = opt->memory;
  a.TolFun = opt->TolFun;
  a.TolX = opt->TolX;
  a.lb = lbub.d();
  a.ub = lbub.d() + nV;
  a.v = v;
  a.g = gbuf.d();
  a.d = dbuf.d();
  a.vt = vtbuf.d();
  a.gt = gtbuf.d();
  a.J = J;
  a.Jt = sc.d();
  a.alpha = sc.d() + B;
  a.lam = sc.d() + 2 * B;
  a.gtd = sc.d() + 3 * B;
  a.fmax = sc.d() + 4 * B;
  a.hist = hist.d();
  a.active = fl;
  a.accepted = fl + B;
  a.iters = fl + 2 * B;
  a.counter = (int*)counter.p;
  auto count_after = [&](int which, int it, int* n) -> int {
    HIP_TRY(hipMemsetAsync(counter.p, 0, sizeof(int), s));
    LAUNCH_TRY(launch_spg(which, a, it, nullptr, nullptr, s));
    HIP_TRY(hipMemcpyAsync(n, counter.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return OCS_OK;
  };
  OCS_TRY(ocs_nlp_objective_dev(g, p, c, batch, x0, v, nFree, FreeInitStates, J, a.g, stream));
  LAUNCH_TRY(launch_spg(0, a, 0, nullptr, nullptr, s));
  for (int it = 0; it < opt->MaxIter; ++it) {
    int nactive = 0;
    OCS_TRY(count_after(1, it, &nactive));  // stopping test + direction + first trial point
    if (nactive == 0) break;
    for (int ls = 0; ls < opt->maxBacktracks; ++ls) {
      OCS_TRY(ocs_nlp_objective_dev(g, p, c, batch, x0, a.vt, nFree, FreeInitStates, a.Jt, a.gt, stream));
      int nrejected = 0;
      OCS_TRY(count_after(2, it, &nrejected));
      if (nrejected == 0) break;
    }
    LAUNCH_TRY(launch_spg(3, a, it, nullptr, nullptr, s));
  }
  // x0 at FreeInitStates follows the last evaluated trial point: make it (and J, g) those of the returned v
  if (nFree > 0) OCS_TRY(ocs_nlp_objective_dev(g, p, c, batch, x0, v, nFree, FreeInitStates, J, a.g, stream));
  OCS_TRY(conv.ensure(sizeof(int) * B));
  LAUNCH_TRY(launch_spg(4, a, 0, pgnorm, (int*)conv.p, s));
  std::vector<int> hc(B);
  HIP_TRY(hipMemcpyAsync(hc.data(), conv.p, sizeof(int) * B, hipMemcpyDeviceToHost, s));
  if (converged) HIP_TRY(hipMemcpyAsync(converged, conv.p, sizeof(int) * B, hipMemcpyDeviceToDevice, s));
  if (iterations) HIP_TRY(hipMemcpyAsync(iterations, a.iters, sizeof(int) * B, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (size_t b = 0; b < B; ++b)
    if (!hc[b]) return OCS_NUM_NOT_CONVERGED;
  return OCS_OK;
}

}  // extern "C"
