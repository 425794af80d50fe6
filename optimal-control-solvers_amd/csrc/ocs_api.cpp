// ocs_api.cpp -- the C-ABI of libocs.so (include/ocs.h): handle bookkeeping, argument
// validation, host<->device staging and layout conversion around the gfx950 kernels.
// No compute happens on the host here; if no MI355X is usable every compute entry point
// fails with OCS_ERR_NO_DEVICE.
#include "ocs_trace.hpp"
#include "ocs_handles.hpp"

using namespace ocs;

// ------------------------------------------------------------------------------------
// library
// ------------------------------------------------------------------------------------
extern "C" {

const char* ocs_version(void) { return "ocs-mi355x 0.1 (gfx950, fp64)"; }
const char* ocs_last_error(void) { return err_string().c_str(); }

int ocs_device_count(int* count) {
  if (!count) return fail(OCS_ERR_INVALID, "count is NULL");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return OCS_OK;
}
int ocs_set_device(int device) {
  OCS_TRY(require_device());
  HIP_TRY(hipSetDevice(device));
  return OCS_OK;
}
int ocs_synchronize(void) {
  OCS_TRY(require_device());
  HIP_TRY(hipDeviceSynchronize());
  return OCS_OK;
}

// ------------------------------------------------------------------------------------
// OCProblem
// ------------------------------------------------------------------------------------
int ocs_problem_create(ocs_problem* out, int problem_id, int nS, int nC, const double* params, int nparams,
                       const double* control_bounds) {
  if (!out || !params || !control_bounds) return fail(OCS_ERR_INVALID, "null argument");
  *out = nullptr;
  ocs_problem_s* p = new ocs_problem_s();
  p->id = problem_id;
  p->nS = nS;
  p->nC = nC;
  if (problem_id == OCS_PROBLEM_TEST) {
    // tests/TestOCProblem.m:16-20, params [c m r] -> LogisticK<1> block [c r m]
    if (nS != 1 || nC != 1 || nparams != 3) {
      delete p;
      return fail(OCS_ERR_SHAPE, "TestOCProblem needs nS=1, nC=1, params [c m r]");
    }
    p->functor = Functor::Logistic;
    p->par = {params[0], params[2], params[1]};
    p->user2func = {0, 2, 1};
  } else if (problem_id == OCS_PROBLEM_LOGISTIC) {
    if (nS < 1 || nC != 1 || nparams != 2 + nS) {
      delete p;
      return fail(OCS_ERR_SHAPE, "LogisticK needs nC=1 and params [c r m_1..m_nS]");
    }
    p->functor = Functor::Logistic;
    p->par.assign(params, params + nparams);
    p->user2func.resize(nparams);
    for (int k = 0; k < nparams; ++k) p->user2func[k] = k;
  } else if (problem_id == OCS_PROBLEM_LQ) {
    // build-defined linear-quadratic problem: params [r | A | Bu | q | rdiag]
    if (nS < 1 || nC < 1 || nparams != 1 + nS * nS + nS * nC + nS + nC) {
      delete p;
      return fail(OCS_ERR_SHAPE, "LQ needs params [r | A (nS x nS) | Bu (nS x nC) | q (nS) | rdiag (nC)]");
    }
    p->functor = Functor::LQ;
    p->par.assign(params, params + nparams);
    p->user2func.resize(nparams);
    for (int k = 0; k < nparams; ++k) p->user2func[k] = k;
  } else {
    delete p;
    return fail(OCS_ERR_UNSUPPORTED, "unknown problem id %d", problem_id);
  }
  if (!functor_supported(p->functor, nS, nC)) {
    delete p;
    return fail(OCS_ERR_UNSUPPORTED, "no kernel instantiated for nS=%d nC=%d", nS, nC);
  }
  p->bounds.assign(control_bounds, control_bounds + 2 * nC);
  p->version = next_version();
  *out = p;
  return OCS_OK;
}

int ocs_problem_destroy(ocs_problem p) {
  if (!p) return OCS_OK;
  if (p->shadow) ocs_problem_destroy(p->shadow);
  if (p->user) jit_free(p->user);
  p->d_ps.release();
  p->d_pb.release();
  p->d_lb.release();
  p->d_ub.release();
  delete p;
  return OCS_OK;
}

int ocs_problem_dims(ocs_problem p, int* nS, int* nC) {
  if (!p) return fail(OCS_ERR_INVALID, "null problem");
  if (nS) *nS = p->nS;
  if (nC) *nC = p->nC;
  return OCS_OK;
}

int ocs_problem_set_batch_params(ocs_problem p, int batch, const int* param_index, int nidx,
                                 const double* values) {
  if (!p) return fail(OCS_ERR_INVALID, "null problem");
  if (nidx == 0) {  // clear
    p->pmask = 0;
    p->pb_batch = 0;
    p->version = next_version();
    return OCS_OK;
  }
  if (!param_index || !values || batch < 1 || nidx < 0) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(upload_problem(p));
  const int npar = (int)p->par.size();
  if (npar > 32) return fail(OCS_ERR_UNSUPPORTED, "per-trajectory parameters need <= 32 parameters");
  if (p->user && (npar > 16 || user_rowsep(p->user)))
    return fail(OCS_ERR_UNSUPPORTED, "user problems with more than 16 parameters or given as row functions read the "
                                     "shared parameter block only: no per-trajectory parameters");
  if (p->user && p->user->tcoef_hooks)
    return fail(OCS_ERR_UNSUPPORTED, "the problem source defines ocs_tcoef / ocs_cc_tcoef, whose values are tabulated once per "
                                     "grid point from the SHARED parameter block: no per-trajectory parameters");
  const unsigned tcmask = functor_tc_param_mask(p->functor, p->nS);
  unsigned mask = 0;
  std::vector<double> pb((size_t)npar * batch, 0.0);
  for (int q = 0; q < nidx; ++q) {
    const int ui = param_index[q];
    if (ui < 0 || ui >= npar) return fail(OCS_ERR_INVALID, "parameter index %d out of range", ui);
    const int fi = p->user2func[ui];
    if ((tcmask >> fi) & 1u)
      return fail(OCS_ERR_UNSUPPORTED,
                  "parameter %d feeds the time-coefficient table and must be batch-uniform", ui);
    mask |= 1u << fi;
    for (int b = 0; b < batch; ++b) pb[(size_t)fi * batch + b] = values[(size_t)b * nidx + q];
  }
  OCS_TRY(p->d_pb.ensure(sizeof(double) * pb.size()));
  HIP_TRY(hipMemcpy(p->d_pb.p, pb.data(), sizeof(double) * pb.size(), hipMemcpyHostToDevice));
  p->pmask = mask;
  p->pb_batch = batch;
  p->version = next_version();
  return OCS_OK;
}

static int eval_common(ocs_problem p, int which, int k, const double* t, const double* y, const double* u,
                       const double* v, double* out) {
  if (!p || !t || !y || !u || !out || (which != 0 && !v) || k < 1) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(upload_problem(p));
  const int nAug = p->nS + 1, nC = p->nC;
  const int nout = (which == 2) ? nC : nAug;
  DevBuf dt, dy, du, dv, dout;
  int rc = OCS_OK;
  auto body = [&]() -> int {
    OCS_TRY(dt.ensure(sizeof(double) * k));
    OCS_TRY(dy.ensure(sizeof(double) * (size_t)nAug * k));
    OCS_TRY(du.ensure(sizeof(double) * (size_t)nC * k));
    OCS_TRY(dv.ensure(sizeof(double) * (size_t)nAug * k));
    OCS_TRY(dout.ensure(sizeof(double) * (size_t)nout * k));
    HIP_TRY(hipMemcpy(dt.p, t, sizeof(double) * k, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dy.p, y, sizeof(double) * (size_t)nAug * k, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(du.p, u, sizeof(double) * (size_t)nC * k, hipMemcpyHostToDevice));
    if (which != 0) HIP_TRY(hipMemcpy(dv.p, v, sizeof(double) * (size_t)nAug * k, hipMemcpyHostToDevice));
    LAUNCH_TRY(launch_eval(describe(p), which, k, dt.d(), dy.d(), du.d(), dv.d(), dout.d(), nullptr));
    HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * (size_t)nout * k, hipMemcpyDeviceToHost));
    return OCS_OK;
  };
  rc = body();
  dt.release();
  dy.release();
  du.release();
  dv.release();
  dout.release();
  return rc;
}
int ocs_problem_F(ocs_problem p, int k, const double* t, const double* y, const double* u, double* out) {
  return eval_common(p, 0, k, t, y, u, nullptr, out);
}
int ocs_problem_dFdx_times_vec(ocs_problem p, int k, const double* t, const double* y, const double* u,
                               const double* v, double* out) {
  return eval_common(p, 1, k, t, y, u, v, out);
}
int ocs_problem_dFdu_times_vec(ocs_problem p, int k, const double* t, const double* y, const double* u,
                               const double* v, double* out) {
  return eval_common(p, 2, k, t, y, u, v, out);
}

// compute_equilibrium.m:1-34, batched (k_equilibrium)
static int equilibrium_check(ocs_problem p, int batch) {
  if (!p || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  if (p->functor == Functor::LQ) return fail(OCS_ERR_UNSUPPORTED, "compute_equilibrium: not instantiated for the LQ problem");
  if (p->functor == Functor::User && 2 * p->nS + p->nC > 24)
    return fail(OCS_ERR_UNSUPPORTED, "compute_equilibrium: at most 24 unknowns per instance");
  if (p->pmask && p->pb_batch != batch)
    return fail(OCS_ERR_SHAPE, "problem has per-trajectory parameters for batch %d, call has batch %d", p->pb_batch, batch);
  return OCS_OK;
}
int ocs_compute_equilibrium_dev(ocs_problem p, int batch, double r, const double* yGuess, const double* lb,
                                const double* ub, double* y, double* resnorm, double* residual, int* exitflag,
                                void* stream) {
  OCS_TRACE("ocs_compute_equilibrium_dev");
  if (!yGuess || !lb || !ub || !y || !resnorm || !exitflag) return fail(OCS_ERR_INVALID, "null argument");
  OCS_TRY(equilibrium_check(p, batch));
  OCS_TRY(upload_problem(p));
  // 4000 iterations is lsqnonlin's MaxIter in the reference (:24); the residual tolerance is round-off level
  LAUNCH_TRY(launch_equilibrium(describe(p), batch, r, yGuess, lb, ub, y, resnorm, residual, exitflag, 4000, 1e-15,
                                (hipStream_t)stream));
  return OCS_OK;
}
int ocs_compute_equilibrium(ocs_problem p, int batch, double r, const double* yGuess, const double* lb,
                            const double* ub, double* y, double* resnorm, double* residual, int* exitflag) {
  OCS_TRACE("ocs_compute_equilibrium");
  if (!yGuess || !lb || !ub || !y || !resnorm || !exitflag) return fail(OCS_ERR_INVALID, "null argument");
  OCS_TRY(equilibrium_check(p, batch));
  const int n = 2 * p->nS + p->nC;
  const size_t nb = (size_t)n * batch;
  DevBuf dg, dl, du, dy, dr, dres, df, dst;
  auto body = [&]() -> int {
    OCS_TRY(dst.ensure(sizeof(double) * nb));
    OCS_TRY(dg.ensure(sizeof(double) * nb));
    OCS_TRY(dy.ensure(sizeof(double) * nb));
    OCS_TRY(dres.ensure(sizeof(double) * nb));
    OCS_TRY(dl.ensure(sizeof(double) * n));
    OCS_TRY(du.ensure(sizeof(double) * n));
    OCS_TRY(dr.ensure(sizeof(double) * batch));
    OCS_TRY(df.ensure(sizeof(int) * batch));
    HIP_TRY(hipMemcpy(dst.p, yGuess, sizeof(double) * nb, hipMemcpyHostToDevice));
    LAUNCH_TRY(launch_to_batch_minor(dst.d(), dg.d(), n, batch, nullptr));
    HIP_TRY(hipMemcpy(dl.p, lb, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(du.p, ub, sizeof(double) * n, hipMemcpyHostToDevice));
    OCS_TRY(ocs_compute_equilibrium_dev(p, batch, r, dg.d(), dl.d(), du.d(), dy.d(), dr.d(), dres.d(), (int*)df.p,
                                        nullptr));
    LAUNCH_TRY(launch_to_traj_major(dy.d(), dst.d(), n, batch, nullptr));
    HIP_TRY(hipMemcpy(y, dst.p, sizeof(double) * nb, hipMemcpyDeviceToHost));
    if (residual) {
      LAUNCH_TRY(launch_to_traj_major(dres.d(), dst.d(), n, batch, nullptr));
      HIP_TRY(hipMemcpy(residual, dst.p, sizeof(double) * nb, hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipMemcpy(resnorm, dr.p, sizeof(double) * batch, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(exitflag, df.p, sizeof(int) * batch, hipMemcpyDeviceToHost));
    for (int b = 0; b < batch; ++b)
      if (exitflag[b] < 0) return OCS_NUM_NONFINITE;   // (a numerical condition: the outputs of the other instances stand)
    return OCS_OK;
  };
  const int rc = body();
  DevBuf* all[] = {&dg, &dl, &du, &dy, &dr, &dres, &df, &dst};
  for (DevBuf* q : all) q->release();
  return rc;
}

// ------------------------------------------------------------------------------------
// Integrator
// ------------------------------------------------------------------------------------
int ocs_rk4_create(ocs_integrator* out, const double* tspan, int npts) {
  if (!out || !tspan) return fail(OCS_ERR_INVALID, "null argument");
  *out = nullptr;
  if (npts < 2) return fail(OCS_ERR_SHAPE, "tspan needs at least 2 points");
  ocs_integrator_s* g = new ocs_integrator_s();
  const int N = npts - 1;
  g->N = N;
  g->tspan.assign(tspan, tspan + npts);
  g->h.resize(N);
  g->t.resize(2 * (size_t)N + 1);
  for (int i = 0; i < N; ++i) g->h[i] = tspan[i + 1] - tspan[i];                  // RK4Integrator.m:17
  for (int i = 0; i <= N; ++i) g->t[2 * (size_t)i] = tspan[i];                    // :22
  for (int i = 0; i < N; ++i) g->t[2 * (size_t)i + 1] = (tspan[i] + tspan[i + 1]) / 2;  // :23
  for (int i = 0; i < N; ++i)
    if (!(g->h[i] > 0) || !std::isfinite(g->h[i])) {
      delete g;
      return fail(OCS_ERR_INVALID, "tspan must be finite and strictly increasing");
    }
  *out = g;
  return OCS_OK;
}

/* obj = RK4InfiniteIntegrator(tspan, tspanExtra, uStar)   Integrator/RK4InfiniteIntegrator.m:12-17 */
int ocs_rk4inf_create(ocs_integrator* out, const double* tspan, int npts, const double* tspanExtra,
                      int nptsExtra, const double* uStar, int nC) {
  if (!out || !tspan || !tspanExtra || !uStar) return fail(OCS_ERR_INVALID, "null argument");
  *out = nullptr;
  if (nC < 1) return fail(OCS_ERR_SHAPE, "uStar needs nC >= 1 entries");
  ocs_integrator g1 = nullptr, g2 = nullptr;
  OCS_TRY(ocs_rk4_create(&g1, tspan, npts));
  int rc = ocs_rk4_create(&g2, tspanExtra, nptsExtra);
  if (rc < 0) {
    ocs_integrator_destroy(g1);
    return rc;
  }
  g1->kind = 1;
  g1->leg2 = g2;
  g1->ustar.assign(uStar, uStar + nC);
  *out = g1;
  return OCS_OK;
}

int ocs_integrator_destroy(ocs_integrator g) {
  if (g && g->tc_event) (void)hipEventDestroy(g->tc_event);
  if (!g) return OCS_OK;
  if (g->leg2) ocs_integrator_destroy(g->leg2);
  if (g->fbs) ocs_fbs_state_free(g->fbs);
  if (g->lqws) lq_workspace_free(g->lqws);
  g->d_ustar.release();
  g->d_lam2.release();
  g->d_utail.release();
  g->d_lamtail.release();
  g->d_J2.release();
  if (g->stream) (void)hipStreamDestroy(g->stream);
  DevBuf* bufs[] = {&g->d_HT, &g->d_T, &g->d_TC, &g->d_TU, &g->d_REC, &g->d_RECS, &g->d_split, &g->d_x0, &g->d_u, &g->d_x, &g->d_J,
                    &g->d_lam, &g->d_dJdu, &g->d_lamT, &g->d_stage, &g->d_ck};
  for (DevBuf* b : bufs) b->release();
  delete g;
  return OCS_OK;
}
int ocs_integrator_set_mapping(ocs_integrator g, int mapping) {
  if (!g) return fail(OCS_ERR_INVALID, "null integrator");
  if (mapping < MAP_AUTO || mapping > MAP_SCAN) return fail(OCS_ERR_INVALID, "mapping must be 0..4");
  g->mapping = mapping;
  if (g->leg2) g->leg2->mapping = mapping;
  return OCS_OK;
}
int ocs_integrator_nsteps(ocs_integrator g, int* nsteps) {
  if (!g || !nsteps) return fail(OCS_ERR_INVALID, "null argument");
  *nsteps = g->N;
  return OCS_OK;
}
int ocs_integrator_t(ocs_integrator g, double* t) {
  if (!g || !t) return fail(OCS_ERR_INVALID, "null argument");
  memcpy(t, g->t.data(), sizeof(double) * g->t.size());
  return OCS_OK;
}
int ocs_integrator_h(ocs_integrator g, double* h) {
  if (!g || !h) return fail(OCS_ERR_INVALID, "null argument");
  memcpy(h, g->h.data(), sizeof(double) * g->h.size());
  return OCS_OK;
}

// one RK4 leg, forward.  ck receives the checkpoints (x itself when the caller wants x).
static int leg_forward(ocs_integrator_s* g, ocs_problem_s* p, int batch, const double* x0, const double* u,
                       double* x, double* J, const FwdOpts& o, hipStream_t s) {
  OCS_TRY(bind_problem(g, p, batch, s));
  double* ck = x;
  if (!ck && (g->kind == 1 || o.uconst || g->want_ck)) {
    // checkpoints go to handle-owned scratch so the adjoint pass can still run
    OCS_TRY(g->d_ck.ensure(sizeof(double) * (size_t)(p->nS + 1) * (g->N + 1) * batch));
    ck = g->d_ck.d();
  }
  g->ck = nullptr;
  FwdOpts om = o;
  om.mapping = g->mapping;
  LAUNCH_TRY(launch_forward(describe(p), describe(g), batch, x0, u, ck, J, om, s));
  g->ck = ck;
  g->ck_batch = batch;
  g->ck_prob = p;
  return OCS_OK;
}

int ocs_compute_states_dev(ocs_integrator g, ocs_problem p, int batch, const double* x0, const double* u,
                           double* x, double* J, void* stream) {
  OCS_TRACE("ocs_compute_states_dev");
  if (!g || !p || !x0 || !u || !J) return fail(OCS_ERR_INVALID, "null argument");
  if (batch < 1) return fail(OCS_ERR_SHAPE, "batch must be >= 1");
  hipStream_t s = (hipStream_t)stream;
  g->want_ck = true;
  if (g->kind == 0) return leg_forward(g, p, batch, x0, u, x, J, FwdOpts(), s);
  // RK4InfiniteIntegrator.m:20-24: [x,J1] = leg1(x0,u); [~,J2] = leg2(x(1:end-1,end), uStar); J = J1 + J2
  if ((int)g->ustar.size() != p->nC) return fail(OCS_ERR_SHAPE, "uStar has %d entries, problem has nC=%d",
                                                 (int)g->ustar.size(), p->nC);
  OCS_TRY(leg_forward(g, p, batch, x0, u, x, J, FwdOpts(), s));
  const double* xT = g->ck + (size_t)g->N * (p->nS + 1) * batch;  // x(1:nS, end): rows are contiguous [nS][B]
  ocs_integrator_s* g2 = g->leg2;
  OCS_TRY(g2->d_ck.ensure(sizeof(double) * (size_t)(p->nS + 1) * (g2->N + 1) * batch));
  // The tail leg has a constant control.  The lane kernels take it as a parameter and read nothing (the mapping for a full
  // chip); where the wave-specialised kernels would be chosen (small batches: the lane kernels are chain-bound, 650 against
  // 125 us for the tail at batch 4096) the tail runs on them, on SAMPLES of the constant control kept with the handle.
  OCS_TRY(bind_problem(g2, p, batch, s));
  g2->tail_wave = g2->mapping == MAP_AUTO && tail_leg_wave_ok(describe(p), describe(g2), batch);
  if (g2->tail_wave) {
    const size_t nU = (size_t)(2 * g2->N + 1) * p->nC * batch;
    OCS_TRY(g2->d_utail.ensure(sizeof(double) * nU));
    if (g2->utail_batch != batch) {
      LAUNCH_TRY(launch_fill_rows(2 * g2->N + 1, p->nC, batch, g->d_ustar.d(), g2->d_utail.d(), s));
      g2->utail_batch = batch;
    }
    OCS_TRY(g2->d_J2.ensure(sizeof(double) * (size_t)batch));
    OCS_TRY(leg_forward(g2, p, batch, xT, g2->d_utail.d(), g2->d_ck.d(), g2->d_J2.d(), FwdOpts(), s));
    LAUNCH_TRY(launch_add_vec(batch, J, g2->d_J2.d(), J, s));   // J = J1 + J2   :24
    return OCS_OK;
  }
  FwdOpts o2;
  o2.uconst = true;
  o2.Jadd = J;
  OCS_TRY(leg_forward(g2, p, batch, xT, g->d_ustar.d(), g2->d_ck.d(), J, o2, s));
  return OCS_OK;
}

int ocs_compute_adjoints_dev(ocs_integrator g, ocs_problem p, int batch, const double* u, const double* lamT,
                             double* lam, double* dJdu, void* stream) {
  OCS_TRACE("ocs_compute_adjoints_dev");
  if (!g || !p || !u) return fail(OCS_ERR_INVALID, "null argument");
  if (!lam && !dJdu) return fail(OCS_ERR_INVALID, "at least one of lam, dJdu must be requested");
  if (!g->ck || g->ck_prob != p || g->ck_batch != batch)
    return fail(OCS_ERR_ORDER, "compute_adjoints needs compute_states first on the same handle/problem/batch");
  hipStream_t s = (hipStream_t)stream;
  OCS_TRY(bind_problem(g, p, batch, s));
  BwdOpts o;
  o.mapping = g->mapping;
  o.lam0 = g->want_lam0;
  if (!lam) {
    OCS_TRY(g->d_split.ensure(sizeof(double) * (size_t)(p->nS + 1) * batch));
    o.split_scratch = g->d_split.d();
  }
  if (g->kind == 1) {
    // RK4InfiniteIntegrator.m:27-30: lam2 = leg2.adjoints(uStar); [lam,dJdu] = leg1.adjoints(u, lam2(:,1))
    if (lamT) return fail(OCS_ERR_INVALID, "RK4InfiniteIntegrator.compute_adjoints takes no lamT");
    ocs_integrator_s* g2 = g->leg2;
    if (!g2->ck || g2->ck_prob != p || g2->ck_batch != batch)
      return fail(OCS_ERR_ORDER, "tail leg has no forward pass");
    OCS_TRY(bind_problem(g2, p, batch, s));
    OCS_TRY(g->d_lam2.ensure(sizeof(double) * (size_t)(p->nS + 1) * batch));
    if (g2->tail_wave && g2->utail_batch == batch) {   // (the state pass of the tail ran on the sampled control: so does this)
      OCS_TRY(g2->d_lamtail.ensure(sizeof(double) * (size_t)(g2->N + 1) * (p->nS + 1) * batch));
      BwdOpts o2;
      o2.mapping = g2->mapping;
      LAUNCH_TRY(launch_backward(describe(p), describe(g2), batch, g2->ck, g2->d_utail.d(), nullptr, g2->d_lamtail.d(),
                                 nullptr, o2, s));
      lamT = g2->d_lamtail.d();   // lam2(:, 1): the first column
    } else {
      BwdOpts o2;
      o2.mapping = g2->mapping;
      o2.uconst = true;
      o2.lam0 = g->d_lam2.d();
      LAUNCH_TRY(launch_backward(describe(p), describe(g2), batch, g2->ck, g->d_ustar.d(), nullptr, nullptr, nullptr,
                                 o2, s));
      lamT = g->d_lam2.d();
    }
  }
  LAUNCH_TRY(launch_backward(describe(p), describe(g), batch, g->ck, u, lamT, lam, dJdu, o, s));
  return OCS_OK;
}

// host staging: MATLAB-shaped host array (per doubles per trajectory, trajectory-major) -> batch-minor
static int stage_in(ocs_integrator_s* g, const double* host, DevBuf& dst, int per, int batch) {
  const size_t bytes = sizeof(double) * (size_t)per * batch;
  OCS_TRY(g->d_stage.ensure(bytes));
  OCS_TRY(dst.ensure(bytes));
  HIP_TRY(hipMemcpyAsync(g->d_stage.p, host, bytes, hipMemcpyHostToDevice, g->stream));
  LAUNCH_TRY(launch_to_batch_minor(g->d_stage.d(), dst.d(), per, batch, g->stream));
  return OCS_OK;
}
static int stage_out(ocs_integrator_s* g, const DevBuf& src, double* host, int per, int batch) {
  const size_t bytes = sizeof(double) * (size_t)per * batch;
  OCS_TRY(g->d_stage.ensure(bytes));
  LAUNCH_TRY(launch_to_traj_major(src.d(), g->d_stage.d(), per, batch, g->stream));
  HIP_TRY(hipMemcpyAsync(host, g->d_stage.p, bytes, hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(hipStreamSynchronize(g->stream));  // d_stage is reused by the next stage_* call
  return OCS_OK;
}

int ocs_compute_states(ocs_integrator g, ocs_problem p, int batch, const double* x0, const double* u,
                       double* x, double* J) {
  OCS_TRACE("ocs_compute_states");
  if (!g || !p || !x0 || !u || !J) return fail(OCS_ERR_INVALID, "null argument");
  if (batch < 1) return fail(OCS_ERR_SHAPE, "batch must be >= 1");
  OCS_TRY(upload_grid(g));
  const int nAug = p->nS + 1, nC = p->nC, N = g->N;
  HIP_TRY(hipStreamSynchronize(g->stream));
  OCS_TRY(stage_in(g, x0, g->d_x0, p->nS, batch));
  HIP_TRY(hipStreamSynchronize(g->stream));
  OCS_TRY(stage_in(g, u, g->d_u, nC * (2 * N + 1), batch));
  OCS_TRY(g->d_x.ensure(sizeof(double) * (size_t)nAug * (N + 1) * batch));
  OCS_TRY(g->d_J.ensure(sizeof(double) * batch));
  OCS_TRY(ocs_compute_states_dev(g, p, batch, g->d_x0.d(), g->d_u.d(), g->d_x.d(), g->d_J.d(), g->stream));
  HIP_TRY(hipMemcpyAsync(J, g->d_J.p, sizeof(double) * batch, hipMemcpyDeviceToHost, g->stream));
  if (x) {
    OCS_TRY(stage_out(g, g->d_x, x, nAug * (N + 1), batch));
  } else {
    HIP_TRY(hipStreamSynchronize(g->stream));
  }
  g->traj_status.assign(batch, 0);
  int rc = OCS_OK;
  for (int b = 0; b < batch; ++b)
    if (!std::isfinite(J[b])) {
      g->traj_status[b] = OCS_NUM_NONFINITE;
      rc = OCS_NUM_NONFINITE;
    }
  return rc;
}

int ocs_integrator_trajectory_status(ocs_integrator g, int batch, int* status) {
  if (!g || !status) return fail(OCS_ERR_INVALID, "null argument");
  if ((int)g->traj_status.size() != batch)
    return fail(OCS_ERR_ORDER, "no host compute call with batch %d has run on this handle", batch);
  for (int b = 0; b < batch; ++b) status[b] = g->traj_status[b];
  return OCS_OK;
}
int ocs_trajectory_status_dev(int batch, const double* J, int* status, void* stream) {
  if (!J || !status || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(require_device());
  LAUNCH_TRY(launch_traj_status(J, batch, status, (hipStream_t)stream));
  return OCS_OK;
}
int ocs_tracing_enabled(void) { return roctx_available() ? 1 : 0; }

int ocs_compute_adjoints(ocs_integrator g, ocs_problem p, int batch, const double* u, const double* lamT,
                         double* lam, double* dJdu) {
  OCS_TRACE("ocs_compute_adjoints");
  if (!g || !p || !u || !lam) return fail(OCS_ERR_INVALID, "null argument");
  if (!g->ck || g->ck != g->d_x.d() || g->ck_prob != p || g->ck_batch != batch)
    return fail(OCS_ERR_ORDER, "compute_adjoints needs compute_states first on the same handle/problem/batch");
  const int nAug = p->nS + 1, nC = p->nC, N = g->N;
  HIP_TRY(hipStreamSynchronize(g->stream));
  OCS_TRY(stage_in(g, u, g->d_u, nC * (2 * N + 1), batch));
  if (lamT) {
    HIP_TRY(hipStreamSynchronize(g->stream));
    OCS_TRY(stage_in(g, lamT, g->d_lamT, nAug, batch));
  }
  OCS_TRY(g->d_lam.ensure(sizeof(double) * (size_t)nAug * (N + 1) * batch));
  if (dJdu) OCS_TRY(g->d_dJdu.ensure(sizeof(double) * (size_t)nC * (2 * N + 1) * batch));
  OCS_TRY(ocs_compute_adjoints_dev(g, p, batch, g->d_u.d(), lamT ? g->d_lamT.d() : nullptr, g->d_lam.d(),
                                   dJdu ? g->d_dJdu.d() : nullptr, g->stream));
  OCS_TRY(stage_out(g, g->d_lam, lam, nAug * (N + 1), batch));
  if (dJdu) OCS_TRY(stage_out(g, g->d_dJdu, dJdu, nC * (2 * N + 1), batch));
  return OCS_OK;
}

int ocs_device_malloc(void** ptr, unsigned long bytes) {
  if (!ptr || bytes == 0) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(require_device());
  *ptr = nullptr;
  HIP_TRY(hipMalloc(ptr, (size_t)bytes));
  return OCS_OK;
}
int ocs_device_free(void* ptr) {
  if (ptr) HIP_TRY(hipFree(ptr));
  return OCS_OK;
}
int ocs_device_upload(void* dst, const void* src, unsigned long bytes, void* stream) {
  if (!dst || !src) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(require_device());
  if (stream) HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  else HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
  return OCS_OK;
}
int ocs_device_download(void* dst, const void* src, unsigned long bytes, void* stream) {
  if (!dst || !src) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(require_device());
  if (stream) HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  else HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
  return OCS_OK;
}

int ocs_copy_dev(const double* src, double* dst, long n, void* stream) {
  if (!src || !dst || n < 1) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(require_device());
  LAUNCH_TRY(launch_copy8(src, dst, (size_t)n, (hipStream_t)stream));
  return OCS_OK;
}
int ocs_to_batch_minor_dev(const double* src, double* dst, int per_traj, int batch, void* stream) {
  if (!src || !dst || per_traj < 1 || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(require_device());
  LAUNCH_TRY(launch_to_batch_minor(src, dst, per_traj, batch, (hipStream_t)stream));
  return OCS_OK;
}
int ocs_to_traj_major_dev(const double* src, double* dst, int per_traj, int batch, void* stream) {
  if (!src || !dst || per_traj < 1 || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(require_device());
  LAUNCH_TRY(launch_to_traj_major(src, dst, per_traj, batch, (hipStream_t)stream));
  return OCS_OK;
}

}  // extern "C"
