// ocs_api.cpp -- the C-ABI of libocs.so (include/ocs.h): handle bookkeeping, argument
// validation, host<->device staging and layout conversion around the gfx950 kernels.
// No compute happens on the host here; if no MI355X is usable every compute entry point
// fails with OCS_ERR_NO_DEVICE.
#include "../../include/ocs.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ocs_internal.hpp"

using namespace ocs;

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(OCS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define LAUNCH_TRY(expr)                                                                    \
  do {                                                                                      \
    int rc_ = (expr);                                                                       \
    if (rc_ < 0) return fail(OCS_ERR_UNSUPPORTED, "%s: no kernel for this problem", #expr); \
    if (rc_ > 0) return fail(OCS_ERR_HIP, "%s: %s", #expr, hipGetErrorString((hipError_t)rc_)); \
  } while (0)
#define OCS_TRY(expr)         \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ < 0) return rc_;  \
  } while (0)

static int require_device() {
  static int state = 0;  // 0 unknown, 1 ok, -1 none
  if (state == 0) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    state = (e == hipSuccess && n > 0) ? 1 : -1;
  }
  if (state < 0)
    return fail(OCS_ERR_NO_DEVICE, "no HIP device: libocs has no CPU fallback, an MI355X is required");
  return OCS_OK;
}

// ------------------------------------------------------------------------------------
// device buffers
// ------------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return OCS_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(&p, bytes));
    cap = bytes;
    return OCS_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  double* d() const { return static_cast<double*>(p); }
};

// ------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------
static unsigned long long g_version_counter = 1;

struct ocs_problem_s {
  int id = 0, nS = 0, nC = 0;
  Functor functor = Functor::Logistic;
  std::vector<double> par;      // functor order
  std::vector<int> user2func;   // user parameter index -> functor parameter index
  std::vector<double> bounds;   // nC x 2
  DevBuf d_ps, d_pb, d_lb, d_ub;
  unsigned pmask = 0;
  int pb_batch = 0;
  unsigned long long version = 0;  // bumps whenever device-visible parameters change
  bool uploaded = false;
};

struct ocs_integrator_s {
  int N = 0;
  std::vector<double> tspan, t, h;
  DevBuf d_HT, d_T, d_TC, d_TU, d_REC;
  bool grid_uploaded = false;
  unsigned long long tc_version = 0;  // version of the problem TC was built for
  const ocs_problem_s* tc_prob = nullptr;
  // state of the last forward pass (the xK contract of RK4Integrator.m:10,32)
  const double* ck = nullptr;
  int ck_batch = 0;
  const ocs_problem_s* ck_prob = nullptr;
  // staging for the host entry points
  hipStream_t stream = nullptr;
  DevBuf d_x0, d_u, d_x, d_J, d_lam, d_dJdu, d_lamT, d_stage, d_ck;
};

static int upload_problem(ocs_problem_s* p) {
  if (p->uploaded) return OCS_OK;
  OCS_TRY(require_device());
  OCS_TRY(p->d_ps.ensure(sizeof(double) * p->par.size()));
  HIP_TRY(hipMemcpy(p->d_ps.p, p->par.data(), sizeof(double) * p->par.size(), hipMemcpyHostToDevice));
  OCS_TRY(p->d_lb.ensure(sizeof(double) * p->nC));
  OCS_TRY(p->d_ub.ensure(sizeof(double) * p->nC));
  HIP_TRY(hipMemcpy(p->d_lb.p, p->bounds.data(), sizeof(double) * p->nC, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_ub.p, p->bounds.data() + p->nC, sizeof(double) * p->nC, hipMemcpyHostToDevice));
  p->uploaded = true;
  return OCS_OK;
}

static ProblemDesc describe(const ocs_problem_s* p) {
  ProblemDesc d;
  d.functor = p->functor;
  d.nS = p->nS;
  d.nC = p->nC;
  d.npar = (int)p->par.size();
  d.ps = p->d_ps.d();
  d.pb = p->pmask ? p->d_pb.d() : nullptr;
  d.pmask = p->pmask;
  d.lb = p->d_lb.d();
  d.ub = p->d_ub.d();
  return d;
}

static int upload_grid(ocs_integrator_s* g) {
  if (g->grid_uploaded) return OCS_OK;
  OCS_TRY(require_device());
  const int N = g->N;
  std::vector<double> HT((size_t)4 * N);
  for (int i = 0; i < N; ++i) {  // the divisions the reference performs per step, done once in IEEE fp64
    HT[4 * i + 0] = g->h[i];
    HT[4 * i + 1] = g->h[i] / 2;  // RK4Integrator.m:40
    HT[4 * i + 2] = g->h[i] / 6;  // :50, :73
    HT[4 * i + 3] = g->h[i] / 3;  // :77
  }
  OCS_TRY(g->d_HT.ensure(sizeof(double) * HT.size()));
  HIP_TRY(hipMemcpy(g->d_HT.p, HT.data(), sizeof(double) * HT.size(), hipMemcpyHostToDevice));
  OCS_TRY(g->d_T.ensure(sizeof(double) * g->t.size()));
  HIP_TRY(hipMemcpy(g->d_T.p, g->t.data(), sizeof(double) * g->t.size(), hipMemcpyHostToDevice));
  if (!g->stream) HIP_TRY(hipStreamCreate(&g->stream));
  g->grid_uploaded = true;
  return OCS_OK;
}

static GridDesc describe(const ocs_integrator_s* g) {
  GridDesc d;
  d.N = g->N;
  d.HT = g->d_HT.d();
  d.T = g->d_T.d();
  d.TC = g->d_TC.d();
  d.TU = g->d_TU.d();
  d.REC = g->d_REC.d();
  return d;
}

// make sure the time-coefficient table of (g, p) is current; enqueued on `s`
static int bind_problem(ocs_integrator_s* g, ocs_problem_s* p, int batch, hipStream_t s) {
  OCS_TRY(upload_problem(p));
  OCS_TRY(upload_grid(g));
  if (p->pmask && p->pb_batch != batch)
    return fail(OCS_ERR_SHAPE, "problem has per-trajectory parameters for batch %d, call has batch %d",
                p->pb_batch, batch);
  if (g->tc_prob != p || g->tc_version != p->version) {
    const int ntc = functor_ntc(p->functor, p->nS);
    const int ntu = functor_ntu(p->functor, p->nS);
    OCS_TRY(g->d_TC.ensure(sizeof(double) * (size_t)(2 * g->N + 1) * ntc));
    OCS_TRY(g->d_TU.ensure(sizeof(double) * (size_t)(2 * g->N + 1) * (ntu > 0 ? ntu : 1)));
    OCS_TRY(g->d_REC.ensure(sizeof(double) * (size_t)g->N * rec_stride_host(ntc)));
    LAUNCH_TRY(launch_tcoef(describe(p), describe(g), s));
    g->tc_prob = p;
    g->tc_version = p->version;
  }
  return OCS_OK;
}

// ------------------------------------------------------------------------------------
// library
// ------------------------------------------------------------------------------------
extern "C" {

const char* ocs_version(void) { return "ocs-mi355x 0.1 (gfx950, fp64)"; }
const char* ocs_last_error(void) { return g_err.c_str(); }

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
  } else {
    delete p;
    return fail(OCS_ERR_UNSUPPORTED, "unknown problem id %d", problem_id);
  }
  if (!functor_supported(p->functor, nS, nC)) {
    delete p;
    return fail(OCS_ERR_UNSUPPORTED, "no kernel instantiated for nS=%d nC=%d", nS, nC);
  }
  p->bounds.assign(control_bounds, control_bounds + 2 * nC);
  p->version = g_version_counter++;
  *out = p;
  return OCS_OK;
}

int ocs_problem_destroy(ocs_problem p) {
  if (!p) return OCS_OK;
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
    p->version = g_version_counter++;
    return OCS_OK;
  }
  if (!param_index || !values || batch < 1 || nidx < 0) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(upload_problem(p));
  const int npar = (int)p->par.size();
  if (npar > 32) return fail(OCS_ERR_UNSUPPORTED, "per-trajectory parameters need <= 32 parameters");
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
  p->version = g_version_counter++;
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

int ocs_integrator_destroy(ocs_integrator g) {
  if (!g) return OCS_OK;
  if (g->stream) (void)hipStreamDestroy(g->stream);
  DevBuf* bufs[] = {&g->d_HT, &g->d_T, &g->d_TC, &g->d_TU, &g->d_REC, &g->d_x0, &g->d_u, &g->d_x, &g->d_J,
                    &g->d_lam, &g->d_dJdu, &g->d_lamT, &g->d_stage, &g->d_ck};
  for (DevBuf* b : bufs) b->release();
  delete g;
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

int ocs_compute_states_dev(ocs_integrator g, ocs_problem p, int batch, const double* x0, const double* u,
                           double* x, double* J, void* stream) {
  if (!g || !p || !x0 || !u || !J) return fail(OCS_ERR_INVALID, "null argument");
  if (batch < 1) return fail(OCS_ERR_SHAPE, "batch must be >= 1");
  hipStream_t s = (hipStream_t)stream;
  OCS_TRY(bind_problem(g, p, batch, s));
  double* ck = x;
  if (!ck) {  // J-only call: checkpoints go to handle-owned scratch so the adjoint pass can still run
    OCS_TRY(g->d_ck.ensure(sizeof(double) * (size_t)(p->nS + 1) * (g->N + 1) * batch));
    ck = g->d_ck.d();
  }
  g->ck = nullptr;
  LAUNCH_TRY(launch_forward(describe(p), describe(g), batch, x0, u, ck, J, s));
  g->ck = ck;
  g->ck_batch = batch;
  g->ck_prob = p;
  return OCS_OK;
}

int ocs_compute_adjoints_dev(ocs_integrator g, ocs_problem p, int batch, const double* u, const double* lamT,
                             double* lam, double* dJdu, void* stream) {
  if (!g || !p || !u) return fail(OCS_ERR_INVALID, "null argument");
  if (!lam && !dJdu) return fail(OCS_ERR_INVALID, "at least one of lam, dJdu must be requested");
  if (!g->ck || g->ck_prob != p || g->ck_batch != batch)
    return fail(OCS_ERR_ORDER, "compute_adjoints needs compute_states first on the same handle/problem/batch");
  hipStream_t s = (hipStream_t)stream;
  OCS_TRY(bind_problem(g, p, batch, s));
  LAUNCH_TRY(launch_backward(describe(p), describe(g), batch, g->ck, u, lamT, lam, dJdu, s));
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
  for (int b = 0; b < batch; ++b)
    if (!std::isfinite(J[b])) return OCS_NUM_NONFINITE;
  return OCS_OK;
}

int ocs_compute_adjoints(ocs_integrator g, ocs_problem p, int batch, const double* u, const double* lamT,
                         double* lam, double* dJdu) {
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
