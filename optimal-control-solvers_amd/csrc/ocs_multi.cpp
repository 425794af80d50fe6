// ocs_multi.cpp -- the batch axis sharded over the GPUs of one node, behind the C-ABI (north_star: "the
// batch axis shards trivially across the 8 GPUs of one node with an RCCL all-reduce of the objective J over xGMI").
//
// The reference has no batch axis and no parallelism (every entry point integrates one trajectory); trajectories are
// independent, so a single-process caller (MATLAB loadlibrary, tests/solve_test_problem.m:37; the iteration loop of
// functions/single_shooting.m:114,137-150) gets N GPUs by handing whole batches to the entry points below.
//
//  * one stream and one PERSISTENT host thread per device (the HIP current device is per thread: a worker sets its device
//    once and keeps it); a call posts one task per device and waits for all of them.  With one device the task runs on
//    the caller's thread.  The caller's current device is restored on every return path.
//  * host entry points (ocs_multi_compute_states ...): MATLAB-shaped host arrays of the whole batch, cut into contiguous
//    blocks (ocs_multi_shard), the one-device host entry point of include/ocs.h on every block -- NO data-path exchange;
//  * device entry points (ocs_multi_*_dev): the blocks are already resident, one device pointer per device, batch-minor
//    like the one-device _dev entry points; the kernels are enqueued on the per-device streams and the call returns
//    without waiting; the O(1)-size reductions are enqueued BEHIND them on the same streams:
//       all-reduce(SUM) of [sum J, count of finite J (or of converged instances)]     2 doubles per device
//       all-gather of (min J, global argmin)                                           2 doubles per device (no MINLOC)
//    ocs_multi_stats waits for the streams and returns the four numbers.
//  * RCCL is loaded at run time (dlopen, like hipRTC in ocs_jit.cpp): libocs.so itself does not depend on it.  Without a
//    communicator (library missing, ncclCommInitAll failing, OCS_MULTI_NO_RCCL=1) the same reductions run on the host
//    over the per-device partial results.
// Handles (problem / integrator / control) own device memory, so the caller creates one set per device (under
// ocs_set_device(ocs_multi_device(m, k))) and passes them as arrays indexed like the devices; a handle whose memory lives
// on another device than its slot is rejected.
#include <rccl/rccl.h>   // types and enumerators only: the functions are looked up with dlsym

#include <dlfcn.h>

#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ocs.h"
#include "ocs_handles.hpp"
#include "ocs_internal.hpp"

using namespace ocs;

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool ok = false;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
#define L(field, sym) *(void**)(&r.field) = dlsym(r.lib, sym)
    L(CommInitAll, "ncclCommInitAll");
    L(CommDestroy, "ncclCommDestroy");
    L(AllReduce, "ncclAllReduce");
    L(AllGather, "ncclAllGather");
    L(GroupStart, "ncclGroupStart");
    L(GroupEnd, "ncclGroupEnd");
    L(GetErrorString, "ncclGetErrorString");
#undef L
    ok = r.CommInitAll && r.CommDestroy && r.AllReduce && r.AllGather && r.GroupStart && r.GroupEnd && r.GetErrorString;
  });
  return ok ? &r : nullptr;
}

// one persistent host thread bound to one device
struct Worker {
  int dev = 0;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> task;
  bool has_task = false, done = false, quit = false, dev_ok = true;
  int rc = OCS_OK;
  std::string msg;

  void loop() {
    dev_ok = hipSetDevice(dev) == hipSuccess;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return has_task || quit; });
      if (quit) return;
      std::function<int()> fn = std::move(task);
      has_task = false;
      lk.unlock();
      int r;
      std::string m;
      if (!dev_ok) {
        r = OCS_ERR_HIP;
        m = "hipSetDevice failed";
      } else {
        r = fn();
        if (r < 0) m = ocs_last_error();   // (thread-local: copy it out of the worker)
      }
      lk.lock();
      rc = r;
      msg = std::move(m);
      done = true;
      cv.notify_all();
    }
  }
  void post(std::function<int()> fn) {
    {
      std::lock_guard<std::mutex> lk(mu);
      task = std::move(fn);
      has_task = true;
      done = false;
    }
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

}  // namespace

struct ocs_multi_s {
  std::vector<int> dev;
  std::vector<hipStream_t> stream;
  std::vector<ncclComm_t> comm;   // empty if there is no communicator (reductions then run on the host)
  std::vector<double*> d_red;     // per device: [0..3] local {sum, count, min, argmin}, [4..5] all-reduced {sum, count},
                                  // [6 .. 6 + 2n) all-gathered (min, argmin) pairs
  std::vector<double*> d_J;       // per device: staging of the shard's J when the entry point keeps none
  std::vector<size_t> d_J_cap;
  std::vector<std::unique_ptr<Worker>> worker;   // n > 1 only
  std::string comm_note;          // why there is no communicator, if there is none
  bool stats_pending = false;     // reductions enqueued by a _dev entry point and not fetched yet
};

namespace {

#define NCCL_TRY(x)                                                                        \
  do {                                                                                     \
    ncclResult_t r_ = (x);                                                                 \
    if (r_ != ncclSuccess) return fail(OCS_ERR_HIP, "RCCL: %s (%s)", rccl()->GetErrorString(r_), #x); \
  } while (0)

void shard(int batch, int n, int k, int* lo, int* hi) {   // as optimal-control-solvers_amd/distributed.py shard_bounds
  const int base = batch / n, rem = batch % n;
  *lo = k * base + (k < rem ? k : rem);
  *hi = *lo + base + (k < rem ? 1 : 0);
}

// runs fn(k) for every device with that device current -- on the persistent worker of the device, or, with one device,
// on the caller's thread (whose device the caller of this function restores).  Returns the first error, else the
// largest (numerical, > 0) status.
template <class F>
int on_devices(ocs_multi_s* m, F fn) {
  const int n = (int)m->dev.size();
  std::vector<int> rc(n, OCS_OK);
  std::vector<std::string> msg(n);
  if (n == 1) {
    if (hipSetDevice(m->dev[0]) != hipSuccess) return fail(OCS_ERR_HIP, "hipSetDevice(%d) failed", m->dev[0]);
    rc[0] = fn(0);
    if (rc[0] < 0) msg[0] = ocs_last_error();
  } else {
    for (int k = 0; k < n; ++k) m->worker[k]->post([&fn, k]() { return fn(k); });
    for (int k = 0; k < n; ++k) {
      m->worker[k]->wait();
      rc[k] = m->worker[k]->rc;
      msg[k] = m->worker[k]->msg;
    }
  }
  int worst = OCS_OK;
  for (int k = 0; k < n; ++k) {
    if (rc[k] < 0) return fail(rc[k], "device %d: %s", m->dev[k], msg[k].c_str());
    if (rc[k] > worst) worst = rc[k];
  }
  return worst;
}

// Enqueues the reductions on the devices' streams, behind whatever produced dJ[k] there (dJ[k]: the shard's objectives
// on device k, nk[k] of them, first global index lo[k]; mask[k]: optional int array, entries <= 0 are left out).
int enqueue_reductions(ocs_multi_s* m, const std::vector<const double*>& dJ, const std::vector<int>& nk,
                       const std::vector<int>& lo, const std::vector<const int*>* mask = nullptr) {
  const int n = (int)m->dev.size();
  for (int k = 0; k < n; ++k) {
    HIP_TRY(hipSetDevice(m->dev[k]));
    LAUNCH_TRY(launch_objective_stats(dJ[k], nk[k], lo[k], m->d_red[k], m->stream[k], mask ? (*mask)[k] : nullptr));
  }
  if (!m->comm.empty()) {
    Rccl* r = rccl();
    NCCL_TRY(r->GroupStart());
    for (int k = 0; k < n; ++k)
      NCCL_TRY(r->AllReduce(m->d_red[k], m->d_red[k] + 4, 2, ncclDouble, ncclSum, m->comm[k], m->stream[k]));
    NCCL_TRY(r->GroupEnd());
    NCCL_TRY(r->GroupStart());
    for (int k = 0; k < n; ++k)
      NCCL_TRY(r->AllGather(m->d_red[k] + 2, m->d_red[k] + 6, 2, ncclDouble, m->comm[k], m->stream[k]));
    NCCL_TRY(r->GroupEnd());
  }
  m->stats_pending = true;
  return OCS_OK;
}

// Waits for the streams and assembles {sum J, count, min J, argmin} over the whole batch.
int fetch_reductions(ocs_multi_s* m, double out[4]) {
  const int n = (int)m->dev.size();
  std::vector<double> h((size_t)6 + 2 * n);
  if (!m->comm.empty()) {
    HIP_TRY(hipSetDevice(m->dev[0]));
    HIP_TRY(hipMemcpyAsync(h.data(), m->d_red[0], sizeof(double) * h.size(), hipMemcpyDeviceToHost, m->stream[0]));
    for (int k = 0; k < n; ++k) {
      HIP_TRY(hipSetDevice(m->dev[k]));
      HIP_TRY(hipStreamSynchronize(m->stream[k]));
    }
  } else {   // no communicator: the same reductions over the per-device partial results, on the host
    h[4] = h[5] = 0.0;
    for (int k = 0; k < n; ++k) {
      double p[4];
      HIP_TRY(hipSetDevice(m->dev[k]));
      HIP_TRY(hipMemcpyAsync(p, m->d_red[k], sizeof(p), hipMemcpyDeviceToHost, m->stream[k]));
      HIP_TRY(hipStreamSynchronize(m->stream[k]));
      h[4] += p[0];
      h[5] += p[1];
      h[6 + 2 * k] = p[2];
      h[7 + 2 * k] = p[3];
    }
  }
  m->stats_pending = false;
  out[0] = h[4];
  out[1] = h[5];
  out[2] = INFINITY;
  out[3] = -1.0;
  for (int k = 0; k < n; ++k)
    if (h[6 + 2 * k] < out[2]) {   // (the first device wins a tie: the smallest global index)
      out[2] = h[6 + 2 * k];
      out[3] = h[7 + 2 * k];
    }
  return OCS_OK;
}

int reduce_objectives(ocs_multi_s* m, const std::vector<const double*>& dJ, const std::vector<int>& nk,
                      const std::vector<int>& lo, double out[4]) {
  OCS_TRY(enqueue_reductions(m, dJ, nk, lo));
  return fetch_reductions(m, out);
}

// every slot has its handles, and no handle's memory lives on another device than its slot's
int check_handles(ocs_multi_s* m, const ocs_integrator* g, const ocs_problem* p, const ocs_control* c) {
  if (!m) return fail(OCS_ERR_INVALID, "null multi-device handle");
  if (!g || !p) return fail(OCS_ERR_INVALID, "null handle array");
  for (size_t k = 0; k < m->dev.size(); ++k) {
    if (!g[k] || !p[k] || (c && !c[k])) return fail(OCS_ERR_INVALID, "null handle for device %d", m->dev[k]);
    const int dg = g[k]->device, dp = p[k]->device, dc = c ? ocs_control_device_id(c[k]) : -1;
    if ((dg >= 0 && dg != m->dev[k]) || (dp >= 0 && dp != m->dev[k]) || (dc >= 0 && dc != m->dev[k]))
      return fail(OCS_ERR_INVALID, "slot %d runs on device %d, but its handles live on device %d (create them under "
                  "ocs_set_device(ocs_multi_device(m, k)))", (int)k, m->dev[k], dg >= 0 && dg != m->dev[k] ? dg : (dp >= 0 && dp != m->dev[k] ? dp : dc));
  }
  return OCS_OK;
}

int blocks(ocs_multi_s* m, int batch, std::vector<int>& lo, std::vector<int>& nk) {
  const int n = (int)m->dev.size();
  if (batch < n) return fail(OCS_ERR_SHAPE, "batch %d is smaller than the number of devices %d", batch, n);
  lo.resize(n);
  nk.resize(n);
  for (int k = 0; k < n; ++k) {
    int hi;
    shard(batch, n, k, &lo[k], &hi);
    nk[k] = hi - lo[k];
  }
  return OCS_OK;
}

int dev_blocks(ocs_multi_s* m, const int* batch, std::vector<int>& lo, std::vector<int>& nk) {
  const int n = (int)m->dev.size();
  if (!batch) return fail(OCS_ERR_INVALID, "null batch array");
  lo.assign(n, 0);
  nk.assign(batch, batch + n);
  for (int k = 0; k < n; ++k) {
    if (nk[k] < 1) return fail(OCS_ERR_SHAPE, "device %d: block of %d trajectories", m->dev[k], nk[k]);
    if (k) lo[k] = lo[k - 1] + nk[k - 1];
  }
  return OCS_OK;
}

}  // namespace

extern "C" {

int ocs_multi_create(ocs_multi* out, const int* devices, int n) {
  if (!out || n < 1 || n > 64) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(require_device());
  DeviceGuard guard;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  // OCS_MULTI_ALLOW_DUPLICATES=1 (testing): a device may be listed more than once, so that the N > 1 code path (worker
  // threads, block arithmetic, reductions) can run on a box with one GPU.  RCCL refuses such a list; the reductions
  // then run on the host.
  const char* dup_env = getenv("OCS_MULTI_ALLOW_DUPLICATES");
  const bool allow_dup = dup_env && dup_env[0] == '1';
  auto* m = new ocs_multi_s;
  bool dup = false;
  for (int k = 0; k < n; ++k) {
    const int d = devices ? devices[k] : k;
    if (d < 0 || d >= ndev) {
      delete m;
      return fail(OCS_ERR_INVALID, "device %d of %d does not exist", d, ndev);
    }
    for (int q : m->dev)
      if (q == d) {
        if (!allow_dup) {
          delete m;
          return fail(OCS_ERR_INVALID, "device %d listed twice", d);
        }
        dup = true;
      }
    m->dev.push_back(d);
  }
  m->stream.assign(n, nullptr);
  m->d_red.assign(n, nullptr);
  m->d_J.assign(n, nullptr);
  m->d_J_cap.assign(n, 0);
  for (int k = 0; k < n; ++k) {
    if (hipSetDevice(m->dev[k]) != hipSuccess || hipStreamCreate(&m->stream[k]) != hipSuccess ||
        hipMalloc((void**)&m->d_red[k], sizeof(double) * (6 + 2 * (size_t)n)) != hipSuccess) {
      ocs_multi_destroy(m);
      return fail(OCS_ERR_HIP, "stream / buffer creation on device %d failed", m->dev[k]);
    }
  }
  // one communicator over the local devices; also for n = 1, so that the collective path is the same
  const char* no_env = getenv("OCS_MULTI_NO_RCCL");
  Rccl* r = rccl();
  if (no_env && no_env[0] == '1') {
    m->comm_note = "OCS_MULTI_NO_RCCL=1";
  } else if (dup) {
    m->comm_note = "a device is listed twice";
  } else if (!r) {
    m->comm_note = "librccl.so could not be loaded";
  } else {
    m->comm.assign(n, nullptr);
    const ncclResult_t e = r->CommInitAll(m->comm.data(), n, m->dev.data());
    if (e != ncclSuccess) {
      m->comm.clear();
      m->comm_note = std::string("ncclCommInitAll: ") + r->GetErrorString(e);
    }
  }
  if (n > 1) {
    for (int k = 0; k < n; ++k) {
      m->worker.emplace_back(new Worker);
      Worker* w = m->worker.back().get();
      w->dev = m->dev[k];
      w->th = std::thread([w] { w->loop(); });
    }
  }
  *out = m;
  return OCS_OK;
}

int ocs_multi_destroy(ocs_multi m) {
  if (!m) return OCS_OK;
  DeviceGuard guard;
  for (auto& w : m->worker) w->stop();
  for (size_t k = 0; k < m->dev.size(); ++k) {
    (void)hipSetDevice(m->dev[k]);
    if (k < m->stream.size() && m->stream[k]) (void)hipStreamSynchronize(m->stream[k]);
    if (k < m->comm.size() && m->comm[k] && rccl()) (void)rccl()->CommDestroy(m->comm[k]);
    if (k < m->stream.size() && m->stream[k]) (void)hipStreamDestroy(m->stream[k]);
    if (k < m->d_red.size() && m->d_red[k]) (void)hipFree(m->d_red[k]);
    if (k < m->d_J.size() && m->d_J[k]) (void)hipFree(m->d_J[k]);
  }
  delete m;
  return OCS_OK;
}

int ocs_multi_size(ocs_multi m) { return m ? (int)m->dev.size() : 0; }

int ocs_multi_device(ocs_multi m, int k) {
  if (!m || k < 0 || k >= (int)m->dev.size()) return fail(OCS_ERR_INVALID, "bad argument");
  return m->dev[k];
}

int ocs_multi_has_communicator(ocs_multi m) {
  if (!m) return fail(OCS_ERR_INVALID, "null multi-device handle");
  if (m->comm.empty()) err_string() = "no RCCL communicator (" + m->comm_note + "): reductions on the host";
  return m->comm.empty() ? 0 : 1;
}

int ocs_multi_stream(ocs_multi m, int k, void** stream) {
  if (!m || !stream || k < 0 || k >= (int)m->dev.size()) return fail(OCS_ERR_INVALID, "bad argument");
  *stream = (void*)m->stream[k];
  return OCS_OK;
}

int ocs_multi_shard(ocs_multi m, int batch, int k, int* lo, int* hi) {
  if (!m || !lo || !hi || batch < 0 || k < 0 || k >= (int)m->dev.size()) return fail(OCS_ERR_INVALID, "bad argument");
  shard(batch, (int)m->dev.size(), k, lo, hi);
  return OCS_OK;
}

int ocs_multi_synchronize(ocs_multi m) {
  if (!m) return fail(OCS_ERR_INVALID, "null multi-device handle");
  DeviceGuard guard;
  for (size_t k = 0; k < m->dev.size(); ++k) {
    HIP_TRY(hipSetDevice(m->dev[k]));
    HIP_TRY(hipStreamSynchronize(m->stream[k]));
  }
  return OCS_OK;
}

int ocs_multi_stats(ocs_multi m, double* stats) {
  if (!m || !stats) return fail(OCS_ERR_INVALID, "bad argument");
  if (!m->stats_pending) return fail(OCS_ERR_ORDER, "no reductions enqueued: call an ocs_multi_*_dev entry point with reduce != 0 first");
  DeviceGuard guard;
  return fetch_reductions(m, stats);
}

// [x, J] = compute_states(obj, prob, x0, u) for a batch (Integrator/RK4Integrator.m:28-56), MATLAB shapes as
// ocs_compute_states; stats (optional, 4 doubles): {sum J, number of finite J, min J, index of the minimum}
int ocs_multi_compute_states(ocs_multi m, const ocs_integrator* g, const ocs_problem* p, int batch, const double* x0,
                             const double* u, double* x, double* J, double* stats) {
  OCS_TRY(check_handles(m, g, p, nullptr));
  if (!x0 || !u || !J || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  DeviceGuard guard;
  const int n = (int)m->dev.size();
  std::vector<int> lo, nk;
  OCS_TRY(blocks(m, batch, lo, nk));
  const int rc = on_devices(m, [&](int k) {
    const size_t nS = p[k]->nS, nC = p[k]->nC, nAug = nS + 1;
    const size_t nT = 2 * (size_t)g[k]->N + 1, nN = (size_t)g[k]->N + 1;
    return ocs_compute_states(g[k], p[k], nk[k], x0 + nS * lo[k], u + nC * nT * lo[k],
                              x ? x + nAug * nN * lo[k] : nullptr, J + lo[k]);
  });
  if (rc < 0) return rc;
  if (stats) {
    std::vector<const double*> dJ(n);
    for (int k = 0; k < n; ++k) dJ[k] = g[k]->d_J.d();
    OCS_TRY(reduce_objectives(m, dJ, nk, lo, stats));
  }
  return rc;
}

// [lam, dJdu] = compute_adjoints(obj, prob, u, lamT) for the batch of the preceding ocs_multi_compute_states
// (RK4Integrator.m:59-121); no reduction: nothing of the adjoint pass is summed over trajectories
int ocs_multi_compute_adjoints(ocs_multi m, const ocs_integrator* g, const ocs_problem* p, int batch, const double* u,
                               const double* lamT, double* lam, double* dJdu) {
  OCS_TRY(check_handles(m, g, p, nullptr));
  if (!u || !lam || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  DeviceGuard guard;
  std::vector<int> lo, nk;
  OCS_TRY(blocks(m, batch, lo, nk));
  return on_devices(m, [&](int k) {
    const size_t nC = p[k]->nC, nAug = (size_t)p[k]->nS + 1, nT = 2 * (size_t)g[k]->N + 1, nN = (size_t)g[k]->N + 1;
    return ocs_compute_adjoints(g[k], p[k], nk[k], u + nC * nT * lo[k], lamT ? lamT + nAug * lo[k] : nullptr,
                                lam + nAug * nN * lo[k], dJdu ? dJdu + nC * nT * lo[k] : nullptr);
  });
}

// [J, dJdv] = nlpObjective(v) for a batch of candidates (functions/single_shooting.m:137-150), shapes as
// ocs_nlp_objective; stats as above (the best candidate of the whole batch in stats[2..3])
int ocs_multi_nlp_objective(ocs_multi m, const ocs_integrator* g, const ocs_problem* p, const ocs_control* c, int batch,
                            double* x0, const double* v, int nFree, const int* FreeInitStates, double* J, double* dJdv,
                            double* stats) {
  if (!c) return fail(OCS_ERR_INVALID, "null handle array");
  OCS_TRY(check_handles(m, g, p, c));
  if (!x0 || !v || !J || !dJdv || batch < 1 || nFree < 0) return fail(OCS_ERR_INVALID, "bad argument");
  DeviceGuard guard;
  const int n = (int)m->dev.size();
  std::vector<int> lo, nk;
  OCS_TRY(blocks(m, batch, lo, nk));
  const int rc = on_devices(m, [&](int k) {
    int nB = 0, nCc = 0, nt = 0;
    OCS_TRY(ocs_control_dims(c[k], &nB, &nCc, &nt));
    const size_t nS = p[k]->nS, nV = (size_t)nCc * nB + nFree;
    return ocs_nlp_objective(g[k], p[k], c[k], nk[k], x0 + nS * lo[k], v + nV * lo[k], nFree, FreeInitStates, J + lo[k],
                             dJdv + nV * lo[k]);
  });
  if (rc < 0) return rc;
  if (stats) {
    std::vector<const double*> dJ(n);
    for (int k = 0; k < n; ++k) dJ[k] = ocs_control_device_J(c[k]);
    OCS_TRY(reduce_objectives(m, dJ, nk, lo, stats));
  }
  return rc;
}

// soln = fb_sweep(prob, x0, tspan, options) for a batch of instances (functions/fb_sweep.m:1-126), shapes as
// ocs_fb_sweep; stats: {sum J over the converged instances, number of converged instances, min J, its index}
int ocs_multi_fb_sweep(ocs_multi m, const ocs_integrator* g, const ocs_problem* p, int batch, const double* x0,
                       const ocs_fbs_options* opt, const double* u0grid, const double* u0err, double* x, double* lam,
                       double* uInterp, double* J, int* sweeps, double* maxChange, double* stats) {
  OCS_TRY(check_handles(m, g, p, nullptr));
  if (!x0 || !opt || !x || !lam || !uInterp || !J || !sweeps || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  DeviceGuard guard;
  const int n = (int)m->dev.size();
  std::vector<int> lo, nk;
  OCS_TRY(blocks(m, batch, lo, nk));
  const int rc = on_devices(m, [&](int k) {
    const size_t nS = p[k]->nS, nC = p[k]->nC, nT = 2 * (size_t)g[k]->N + 1, nN = (size_t)g[k]->N + 1;
    const size_t L = lo[k];
    return ocs_fb_sweep(g[k], p[k], nk[k], x0 + nS * L, opt, u0grid ? u0grid + nC * nT * L : nullptr,
                        u0err ? u0err + nC * (size_t)opt->nERROR_PTS * L : nullptr, x + nS * nN * L, lam + nS * nN * L,
                        uInterp + nC * (size_t)opt->nINTERP_PTS * L, J + L, sweeps + L,
                        maxChange ? maxChange + (size_t)opt->nSWEEPS * L : nullptr);
  });
  if (rc < 0) return rc;
  if (stats) {
    // J of an instance that did not converge is NaN on return (the reference's empty struct): the statistics count the
    // finite ones, i.e. the converged instances.  The shard's J goes back to its device for the reduction.
    std::vector<const double*> dJ(n);
    for (int k = 0; k < n; ++k) {
      HIP_TRY(hipSetDevice(m->dev[k]));
      if (m->d_J_cap[k] < (size_t)nk[k]) {
        if (m->d_J[k]) (void)hipFree(m->d_J[k]);
        m->d_J[k] = nullptr;
        m->d_J_cap[k] = 0;
        HIP_TRY(hipMalloc((void**)&m->d_J[k], sizeof(double) * nk[k]));
        m->d_J_cap[k] = nk[k];
      }
      std::vector<double> Jk(J + lo[k], J + lo[k] + nk[k]);
      for (int b = 0; b < nk[k]; ++b)
        if (sweeps[lo[k] + b] <= 0) Jk[b] = NAN;
      HIP_TRY(hipMemcpy(m->d_J[k], Jk.data(), sizeof(double) * nk[k], hipMemcpyHostToDevice));
      dJ[k] = m->d_J[k];
    }
    OCS_TRY(reduce_objectives(m, dJ, nk, lo, stats));
  }
  return rc;
}

// ---- device-resident blocks: one pointer per device, batch-minor as the one-device _dev entry points --------------------
// The iteration loop of functions/single_shooting.m:114,137-150 keeps its iterates where they are evaluated: nothing
// crosses PCIe, the kernels of a call are enqueued on the per-device streams and the call returns; the reductions (if
// asked for) follow on the same streams, ocs_multi_stats fetches them.

int ocs_multi_compute_states_dev(ocs_multi m, const ocs_integrator* g, const ocs_problem* p, const int* batch,
                                 const double* const* x0, const double* const* u, double* const* x, double* const* J,
                                 int reduce) {
  OCS_TRY(check_handles(m, g, p, nullptr));
  if (!x0 || !u || !J) return fail(OCS_ERR_INVALID, "bad argument");
  DeviceGuard guard;
  const int n = (int)m->dev.size();
  std::vector<int> lo, nk;
  OCS_TRY(dev_blocks(m, batch, lo, nk));
  for (int k = 0; k < n; ++k)
    if (!x0[k] || !u[k] || !J[k]) return fail(OCS_ERR_INVALID, "null array for device %d", m->dev[k]);
  const int rc = on_devices(m, [&](int k) {
    return ocs_compute_states_dev(g[k], p[k], nk[k], x0[k], u[k], x ? x[k] : nullptr, J[k], (void*)m->stream[k]);
  });
  if (rc < 0) return rc;
  if (reduce) {
    std::vector<const double*> dJ(J, J + n);
    OCS_TRY(enqueue_reductions(m, dJ, nk, lo));
  }
  return rc;
}

int ocs_multi_compute_adjoints_dev(ocs_multi m, const ocs_integrator* g, const ocs_problem* p, const int* batch,
                                   const double* const* u, const double* const* lamT, double* const* lam,
                                   double* const* dJdu) {
  OCS_TRY(check_handles(m, g, p, nullptr));
  if (!u || (!lam && !dJdu)) return fail(OCS_ERR_INVALID, "bad argument");
  DeviceGuard guard;
  std::vector<int> lo, nk;
  OCS_TRY(dev_blocks(m, batch, lo, nk));
  return on_devices(m, [&](int k) {
    return ocs_compute_adjoints_dev(g[k], p[k], nk[k], u[k], lamT ? lamT[k] : nullptr, lam ? lam[k] : nullptr,
                                    dJdu ? dJdu[k] : nullptr, (void*)m->stream[k]);
  });
}

int ocs_multi_nlp_objective_dev(ocs_multi m, const ocs_integrator* g, const ocs_problem* p, const ocs_control* c,
                                const int* batch, double* const* x0, const double* const* v, int nFree,
                                const int* FreeInitStates, double* const* J, double* const* dJdv, int reduce) {
  if (!c) return fail(OCS_ERR_INVALID, "null handle array");
  OCS_TRY(check_handles(m, g, p, c));
  if (!x0 || !v || !J || !dJdv || nFree < 0) return fail(OCS_ERR_INVALID, "bad argument");
  DeviceGuard guard;
  const int n = (int)m->dev.size();
  std::vector<int> lo, nk;
  OCS_TRY(dev_blocks(m, batch, lo, nk));
  for (int k = 0; k < n; ++k)
    if (!x0[k] || !v[k] || !J[k] || !dJdv[k]) return fail(OCS_ERR_INVALID, "null array for device %d", m->dev[k]);
  const int rc = on_devices(m, [&](int k) {
    return ocs_nlp_objective_dev(g[k], p[k], c[k], nk[k], x0[k], v[k], nFree, FreeInitStates, J[k], dJdv[k],
                                 (void*)m->stream[k]);
  });
  if (rc < 0) return rc;
  if (reduce) {
    std::vector<const double*> dJ(J, J + n);
    OCS_TRY(enqueue_reductions(m, dJ, nk, lo));
  }
  return rc;
}

// (the sweep loop waits for the count of active instances of every sweep, so each device's loop runs on its worker
//  thread; the call returns when every device has finished its loop, the final kernels may still be in flight)
int ocs_multi_fb_sweep_dev(ocs_multi m, const ocs_integrator* g, const ocs_problem* p, const int* batch,
                           const double* const* x0, const ocs_fbs_options* opt, double* const* xaug, double* const* lam,
                           double* const* uInterp, double* const* J, int* const* sweeps, double* const* maxChange,
                           int reduce) {
  OCS_TRY(check_handles(m, g, p, nullptr));
  if (!x0 || !opt || !xaug || !lam || !uInterp || !J || !sweeps) return fail(OCS_ERR_INVALID, "bad argument");
  DeviceGuard guard;
  const int n = (int)m->dev.size();
  std::vector<int> lo, nk;
  OCS_TRY(dev_blocks(m, batch, lo, nk));
  for (int k = 0; k < n; ++k)
    if (!x0[k] || !xaug[k] || !lam[k] || !uInterp[k] || !J[k] || !sweeps[k])
      return fail(OCS_ERR_INVALID, "null array for device %d", m->dev[k]);
  const int rc = on_devices(m, [&](int k) {
    return ocs_fb_sweep_dev(g[k], p[k], nk[k], x0[k], opt, nullptr, nullptr, xaug[k], lam[k], uInterp[k], J[k], sweeps[k],
                            maxChange ? maxChange[k] : nullptr, (void*)m->stream[k]);
  });
  if (rc < 0) return rc;
  if (reduce) {   // the converged instances only (sweeps > 0)
    std::vector<const double*> dJ(J, J + n);
    std::vector<const int*> mask(sweeps, sweeps + n);
    OCS_TRY(enqueue_reductions(m, dJ, nk, lo, &mask));
  }
  return rc;
}

}  // extern "C"
