// ocs_handles.hpp -- private: handle structs, error plumbing and staging helpers shared by
// the C-ABI translation units (ocs_api.cpp, ocs_control.cpp, ocs_fbs.cpp).
#pragma once
#include "../../include/ocs.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "ocs_internal.hpp"
#include "ocs_jit.hpp"

namespace ocs {

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
inline std::string& err_string() {
  static thread_local std::string s;
  return s;
}

inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  err_string() = buf;
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

inline int require_device() {
  static std::once_flag once;   // (reachable from the worker threads of ocs_multi_*)
  static int state = -1;        // 1 ok, -1 none
  std::call_once(once, [] {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    state = (e == hipSuccess && n > 0) ? 1 : -1;
  });
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
// Globally unique: the integrator tables are cached under (problem pointer, version), and problems are created and
// re-parametrised concurrently from the worker threads of ocs_multi_*.
inline unsigned long long next_version() {
  static std::atomic<unsigned long long> c{1};
  return c.fetch_add(1, std::memory_order_relaxed);
}

// The HIP current device is per host thread.  Every handle records the device it was created on; entry points that
// switch devices (ocs_multi_*) restore the caller's on every return path through this guard.
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

inline int current_device_or(int dflt) {
  int d = dflt;
  return hipGetDevice(&d) == hipSuccess ? d : dflt;
}

}  // namespace ocs

struct ocs_problem_s {
  using DevBuf = ocs::DevBuf;
  using Functor = ocs::Functor;
  int id = 0, nS = 0, nC = 0;
  int device = -1;              // HIP device current when the handle was created (-1: created without a device)
  Functor functor = Functor::Logistic;
  ocs::UserModule* user = nullptr;  // hipRTC module of a user-supplied problem
  std::vector<double> par;      // functor order
  std::vector<int> user2func;   // user parameter index -> functor parameter index
  std::vector<double> bounds;   // nC x 2
  DevBuf d_ps, d_pb, d_lb, d_ub;
  unsigned pmask = 0;
  int pb_batch = 0;
  unsigned long long version = 0;  // bumps whenever device-visible parameters change
  bool uploaded = false;
  // OCS_PROBLEM_LQ only: the same problem as generated device source (hipRTC), built on first use by the
  // forward-backward-sweep entry points, whose kernels are instantiated per functor (ocs_fbs.cpp: lq_shadow)
  ocs_problem_s* shadow = nullptr;
  unsigned long long shadow_version = 0;
};

struct ocs_fbs_state;  // forward-backward-sweep workspace (ocs_fbs.cpp)
void ocs_fbs_state_free(ocs_fbs_state* s);

struct ocs_integrator_s {
  using DevBuf = ocs::DevBuf;
  ocs_fbs_state* fbs = nullptr;
  ocs::LqWorkspace* lqws = nullptr;   // time-parallel LQ passes: chunk matrices of the bound problem + scratch
  int device = -1;   // HIP device current when the handle was created
  int rec_stride = 8;  // doubles per step record of the bound problem
  bool uniform = false;  // all steps of the grid have the same size
  hipEvent_t tc_event = nullptr;   // recorded behind the kernels that build TC / TU / REC / RECS ...
  hipStream_t tc_stream = nullptr; // ... on this stream: a call on another stream waits for it first
  std::vector<int> traj_status;    // per-trajectory flags of the last host compute_states / nlp_objective
  int mapping = 0;  // ocs::Mapping requested through ocs_integrator_set_mapping (0 = automatic)
  int kind = 0;  // 0 RK4Integrator, 1 RK4InfiniteIntegrator (then `leg2` and `ustar` are set)
  ocs_integrator_s* leg2 = nullptr;   // integrator2 of RK4InfiniteIntegrator.m:13-14
  std::vector<double> ustar;          // uStar (nC)
  DevBuf d_ustar, d_lam2;             // device uStar [nC]; lam2(:,1) [nAug][B]
  // tail leg of RK4InfiniteIntegrator on the wave-specialised kernels (small batches): the constant control as samples
  // [2N+1][nC][B], the tail's lam [N+1][nAug][B] (its first column is the main leg's lamT), the tail's objective [B]
  DevBuf d_utail, d_lamtail, d_J2;
  int utail_batch = 0;
  bool tail_wave = false;
  int N = 0;
  std::vector<double> tspan, t, h;
  DevBuf d_HT, d_T, d_TC, d_TU, d_REC, d_RECS, d_split;
  bool grid_uploaded = false;
  unsigned long long tc_version = 0;  // version of the problem TC was built for
  const ocs_problem_s* tc_prob = nullptr;
  bool want_ck = true;          // keep checkpoints of J-only forward passes (adjoint may follow)
  double* want_lam0 = nullptr;  // if set, the next adjoint pass also writes lam(:,1) here ([nAug][B])
  // state of the last forward pass (the xK contract of RK4Integrator.m:10,32)
  const double* ck = nullptr;
  int ck_batch = 0;
  const ocs_problem_s* ck_prob = nullptr;
  // staging for the host entry points
  hipStream_t stream = nullptr;
  DevBuf d_x0, d_u, d_x, d_J, d_lam, d_dJdu, d_lamT, d_stage, d_ck;
};

struct ocs_control_s;
const double* ocs_control_device_J(const ocs_control_s* c);   // ocs_control.cpp
int ocs_control_device_id(const ocs_control_s* c);             // device owning the handle's memory, -1 before the first use

namespace ocs {

inline int upload_problem(ocs_problem_s* p) {
  if (p->uploaded) return OCS_OK;
  OCS_TRY(require_device());
  if (p->device < 0) p->device = current_device_or(-1);   // the device that owns the handle's memory from here on
  OCS_TRY(p->d_ps.ensure(sizeof(double) * p->par.size()));
  HIP_TRY(hipMemcpy(p->d_ps.p, p->par.data(), sizeof(double) * p->par.size(), hipMemcpyHostToDevice));
  OCS_TRY(p->d_lb.ensure(sizeof(double) * p->nC));
  OCS_TRY(p->d_ub.ensure(sizeof(double) * p->nC));
  HIP_TRY(hipMemcpy(p->d_lb.p, p->bounds.data(), sizeof(double) * p->nC, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_ub.p, p->bounds.data() + p->nC, sizeof(double) * p->nC, hipMemcpyHostToDevice));
  p->uploaded = true;
  return OCS_OK;
}

inline ProblemDesc describe(const ocs_problem_s* p) {
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
  d.user = p->user;
  d.version = p->version;
  return d;
}

inline int upload_grid(ocs_integrator_s* g) {
  if (g->grid_uploaded) return OCS_OK;
  OCS_TRY(require_device());
  if (g->device < 0) g->device = current_device_or(-1);
  const int N = g->N;
  std::vector<double> HT((size_t)4 * N);
  for (int i = 0; i < N; ++i) {  // the divisions the reference performs per step, done once in IEEE fp64
    HT[4 * i + 0] = g->h[i];
    HT[4 * i + 1] = g->h[i] / 2;  // RK4Integrator.m:40
    HT[4 * i + 2] = g->h[i] / 6;  // :50, :73
    HT[4 * i + 3] = g->h[i] / 3;  // :77
  }
  g->uniform = true;
  for (int i = 1; i < N; ++i) g->uniform = g->uniform && g->h[i] == g->h[0];
  OCS_TRY(g->d_HT.ensure(sizeof(double) * HT.size()));
  HIP_TRY(hipMemcpy(g->d_HT.p, HT.data(), sizeof(double) * HT.size(), hipMemcpyHostToDevice));
  OCS_TRY(g->d_T.ensure(sizeof(double) * g->t.size()));
  HIP_TRY(hipMemcpy(g->d_T.p, g->t.data(), sizeof(double) * g->t.size(), hipMemcpyHostToDevice));
  if (!g->stream) HIP_TRY(hipStreamCreate(&g->stream));
  if (g->kind == 1) {
    OCS_TRY(g->d_ustar.ensure(sizeof(double) * g->ustar.size()));
    HIP_TRY(hipMemcpy(g->d_ustar.p, g->ustar.data(), sizeof(double) * g->ustar.size(), hipMemcpyHostToDevice));
  }
  g->grid_uploaded = true;
  return OCS_OK;
}

inline GridDesc describe(const ocs_integrator_s* g) {
  GridDesc d;
  d.N = g->N;
  d.HT = g->d_HT.d();
  d.T = g->d_T.d();
  d.TC = g->d_TC.d();
  d.TU = g->d_TU.d();
  d.REC = g->d_REC.d() ? g->d_REC.d() + (size_t)rec_pad_host() * g->rec_stride : nullptr;
  d.RECS = g->d_RECS.d() ? g->d_RECS.d() + scan_recs_front() : nullptr;
  d.uniform = g->uniform;
  d.lqws = const_cast<LqWorkspace**>(&g->lqws);
  return d;
}

// make sure the time-coefficient table of (g, p) is current; enqueued on `s`
inline int bind_problem(ocs_integrator_s* g, ocs_problem_s* p, int batch, hipStream_t s) {
  OCS_TRY(upload_problem(p));
  OCS_TRY(upload_grid(g));
  if (p->pmask && p->pb_batch != batch)
    return fail(OCS_ERR_SHAPE, "problem has per-trajectory parameters for batch %d, call has batch %d",
                p->pb_batch, batch);
  if (g->tc_prob != p || g->tc_version != p->version) {
    const int ntc = functor_ntc(p->functor, p->nS);
    const int ntu = functor_ntu(p->functor, p->nS);
    OCS_TRY(g->d_TC.ensure(sizeof(double) * (size_t)(2 * g->N + 1) * ntc));
    // (+128: the fold kernels stream this table in 1 KiB pieces, the last of which reaches past the horizon)
    OCS_TRY(g->d_TU.ensure(sizeof(double) * ((size_t)(2 * g->N + 1) * (ntu > 0 ? ntu : 1) + 128)));
    g->rec_stride = rec_stride_host(ntc);
    OCS_TRY(g->d_REC.ensure(sizeof(double) * (size_t)(g->N + 2 * rec_pad_host()) * g->rec_stride));
    LAUNCH_TRY(launch_tcoef(describe(p), describe(g), s));
    if (scan_problem_ok(describe(p)) || vector_problem_ok(describe(p))) {
      OCS_TRY(g->d_RECS.ensure(sizeof(double) * scan_recs_doubles(g->N)));
      LAUNCH_TRY(launch_build_recs(g->N, g->rec_stride, rec_sc_offset_host(ntc), describe(g).REC, g->d_RECS.d(), s));
    }
    g->tc_prob = p;
    g->tc_version = p->version;
    // the tables are valid for every later call on this stream; a call on another stream must wait for them
    if (!g->tc_event) HIP_TRY(hipEventCreateWithFlags(&g->tc_event, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(g->tc_event, s));
    g->tc_stream = s;
  } else if (g->tc_event && s != g->tc_stream) {
    HIP_TRY(hipStreamWaitEvent(s, g->tc_event, 0));
  }
  return OCS_OK;
}


}  // namespace ocs
