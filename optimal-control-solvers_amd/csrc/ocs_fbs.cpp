// ocs_fbs.cpp -- forward-backward sweep driver (functions/fb_sweep.m, compute_x_lam.m,
// compute_x_lam_J.m) on the integrator grid, batch-aware: every instance carries its own
// convergence state, the whole batch advances sweep by sweep until no instance is active.
#include "ocs_trace.hpp"
#include <cstdio>
#include <cstdlib>
#include <string>
#include "ocs_handles.hpp"

#include <algorithm>

using namespace ocs;

struct ocs_fbs_state {
  // pchip node tables
  DevBuf TN, HN, W1, W2, TM, IH, PR;
  bool tables = false;
  // query-point tables (error points / interp points), rebuilt when the options change
  int nerr = 0, nint = 0;
  bool err_on_nodes = false;  // the error points are the grid nodes (the default on a linspace tspan)
  int last_path = 0;          // ocs_fb_sweep_path
  DevBuf KE, SE, TE, TUE, KI, SI, TI, TUI;
  DevBuf QSE;   // error points by interval: offsets [n] (they are sorted: linspace)
  unsigned long long tu_version = 0;
  const ocs_problem_s* tu_prob = nullptr;
  // windows of the batch on their own streams (fb_sweep with the fused control update)
  std::vector<hipStream_t> wstreams;
  std::vector<hipEvent_t> wevents;  // two per window: "sweep done" (ping-pong)
  hipEvent_t fork = nullptr, stag = nullptr;
  int* h_nact = nullptr;            // pinned: [windows][nSWEEPS] instances still active after each sweep
  int h_nact_cap = 0;
  DevBuf nact_slots;                // device: the same counters
  // work arrays
  DevBuf xaug, xmid, lam, lmid, ugrid, uerr, uint_, J, usel, status, maxchange, nactive, x0, stage, metric, anyvalid, dump;
};

void ocs_fbs_state_free(ocs_fbs_state* s) {
  if (!s) return;
  s->QSE.release();
  DevBuf* bufs[] = {&s->TN, &s->HN, &s->W1, &s->W2, &s->TM, &s->IH, &s->PR, &s->KE, &s->SE, &s->TE, &s->TUE, &s->KI, &s->SI,
                    &s->TI, &s->TUI, &s->xaug, &s->xmid, &s->lam, &s->lmid, &s->ugrid, &s->uerr, &s->uint_, &s->J,
                    &s->usel, &s->status, &s->maxchange, &s->nactive, &s->x0, &s->stage, &s->metric, &s->anyvalid, &s->dump};
  for (DevBuf* b : bufs) b->release();
  s->nact_slots.release();
  for (hipStream_t st : s->wstreams) (void)hipStreamDestroy(st);
  for (hipEvent_t e : s->wevents) (void)hipEventDestroy(e);
  if (s->fork) (void)hipEventDestroy(s->fork);
  if (s->stag) (void)hipEventDestroy(s->stag);
  if (s->h_nact) (void)hipHostFree(s->h_nact);
  delete s;
}

static void matlab_linspace(double a, double b, int n, std::vector<double>& out) {
  out.resize(n);
  if (n == 1) {
    out[0] = b;
    return;
  }
  const int n1 = n - 1;
  for (int k = 0; k <= n1; ++k) out[k] = a + ((double)k * (b - a)) / (double)n1;
  out[0] = a;
  out[n1] = b;
}

static int upload(DevBuf& d, const void* src, size_t bytes) {
  OCS_TRY(d.ensure(bytes ? bytes : 8));
  if (bytes) HIP_TRY(hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice));
  return OCS_OK;
}

static int ensure_tables(ocs_integrator_s* g) {
  if (!g->fbs) g->fbs = new ocs_fbs_state();
  ocs_fbs_state* f = g->fbs;
  if (f->tables) return OCS_OK;
  const int N = g->N, n = N + 1;
  std::vector<double> tn(n), tm(N), w1(n, 0.0), w2(n, 0.0);
  for (int i = 0; i < n; ++i) tn[i] = g->t[2 * (size_t)i];
  for (int i = 0; i < N; ++i) tm[i] = g->t[2 * (size_t)i + 1];
  const std::vector<double>& h = g->h;
  for (int k = 1; k + 1 < n; ++k) {  // pchip interior weights (MATLAB pchipslopes)
    const double hs = h[k - 1] + h[k];
    w1[k] = (h[k - 1] + hs) / (3 * hs);
    w2[k] = (hs + h[k]) / (3 * hs);
  }
  OCS_TRY(upload(f->TN, tn.data(), sizeof(double) * n));
  OCS_TRY(upload(f->HN, h.data(), sizeof(double) * N));
  OCS_TRY(upload(f->W1, w1.data(), sizeof(double) * n));
  OCS_TRY(upload(f->W2, w2.data(), sizeof(double) * n));
  OCS_TRY(upload(f->TM, tm.data(), sizeof(double) * N));
  std::vector<double> ih(N);
  for (int i = 0; i < N; ++i) ih[i] = 1.0 / h[i];
  OCS_TRY(upload(f->IH, ih.data(), sizeof(double) * N));
  {  // per-interval records: {h(i-1),h(i),h(i+1), their reciprocals, W1(i),W2(i),W1(i+1),W2(i+1), tmid-t(i), h(i)/8, pad}
    const int R = costate_prec();
    std::vector<double> pr((size_t)N * R, 0.0);
    auto cl = [&](int k) { return k < 0 ? 0 : (k > N - 1 ? N - 1 : k); };
    for (int i = 0; i < N; ++i) {
      double* q = &pr[(size_t)i * R];
      for (int e = 0; e < 3; ++e) {
        q[e] = h[cl(i - 1 + e)];
        q[3 + e] = ih[cl(i - 1 + e)];
      }
      q[6] = w1[i];
      q[7] = w2[i];
      q[8] = w1[i + 1];
      q[9] = w2[i + 1];
      q[10] = tm[i] - tn[i];
      q[11] = 0.125 * h[i];   // the Hermite cubic at the middle of its interval: (y0 + y1)/2 + h/8 (d0 - d1)
    }
    OCS_TRY(upload(f->PR, pr.data(), sizeof(double) * pr.size()));
  }
  f->tables = true;
  return OCS_OK;
}

static FbsTables tabs(const ocs_integrator_s* g) {
  const ocs_fbs_state* f = g->fbs;
  return FbsTables{g->N + 1, f->TN.d(), f->HN.d(), f->W1.d(), f->W2.d(), f->TM.d(), f->IH.d(), f->PR.d()};
}

// interval index and local coordinate of query points linspace(T0, TF, nq)
// *on_nodes (optional): the query points are exactly the grid nodes, in order
static int build_points(ocs_integrator_s* g, int nq, DevBuf& K, DevBuf& S, DevBuf& T, bool* on_nodes = nullptr,
                        DevBuf* QS = nullptr) {
  const int n = g->N + 1;
  std::vector<double> q, tn(n), s(nq);
  std::vector<int> k(nq);
  for (int i = 0; i < n; ++i) tn[i] = g->t[2 * (size_t)i];
  matlab_linspace(tn[0], tn[n - 1], nq, q);  // fb_sweep.m:69-70
  for (int j = 0; j < nq; ++j) {
    int lo = 0, hi = n - 1;
    if (q[j] <= tn[0])
      lo = 0;
    else if (q[j] >= tn[n - 1])
      lo = n - 2;
    else {
      while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        if (tn[mid] <= q[j])
          lo = mid;
        else
          hi = mid;
      }
    }
    k[j] = lo;
    s[j] = q[j] - tn[lo];
  }
  if (on_nodes) {
    // "the same" up to a few units in the last place of the horizon: a tspan from another linspace than MATLAB's (numpy's
    // differs from it in the last bit of some nodes) still has its error points on the nodes; sampling the control at
    // the node instead of a point 1e-16 beside it changes the weighted change of :107 at round-off level only
    const double tol = 8.0 * 2.220446049250313e-16 * std::max(std::fabs(tn[0]), std::fabs(tn[n - 1]));
    bool same = nq == n;
    for (int j = 0; same && j < n; ++j) same = std::fabs(q[j] - tn[j]) <= tol;
    *on_nodes = same;
  }
  if (QS) {   // offsets of the points of every interval (the points are in ascending order)
    std::vector<int> qs((size_t)n, 0);
    for (int j = 0; j < nq; ++j) ++qs[k[j] + 1];
    for (int i = 1; i < n; ++i) qs[i] += qs[i - 1];
    OCS_TRY(upload(*QS, qs.data(), sizeof(int) * n));
  }
  OCS_TRY(upload(K, k.data(), sizeof(int) * nq));
  OCS_TRY(upload(S, s.data(), sizeof(double) * nq));
  OCS_TRY(upload(T, q.data(), sizeof(double) * nq));
  return OCS_OK;
}

extern "C" {

// vectorInterpolant(x, v, method)(tq) for a batch of sample sets on the device (functions/vectorInterpolant.m:1-12;
// the step right after the solvers: single_shooting.m:128-130, fb_sweep.m:123).  x [n], tq [nq] host;
// v [n][nComp][B] -> out [nq][nComp][B] device, batch-minor.  Same rules as the host ocs_interp: 'linear' and
// 'pchip' continue their end pieces outside the grid, 'previous' gives NaN before the first sample.
int ocs_interp_dev(int method, int nComp, int n, const double* x, const double* v, int nq, const double* tq,
                   double* out, int batch, void* stream) {
  OCS_TRACE("ocs_interp_dev");
  if (!x || !v || !tq || !out || nComp < 1 || n < 2 || nq < 0 || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  if (method != OCS_INTERP_LINEAR && method != OCS_INTERP_PREVIOUS && method != OCS_INTERP_PCHIP &&
      method != OCS_INTERP_NEAREST && method != OCS_INTERP_NEXT)
    return fail(OCS_ERR_UNSUPPORTED, "unknown interpolation method %d", method);
  for (int i = 0; i + 1 < n; ++i)
    if (!(x[i + 1] > x[i])) return fail(OCS_ERR_INVALID, "sample points must increase strictly");
  if (nq == 0) return OCS_OK;
  hipStream_t s = (hipStream_t)stream;
  std::vector<double> h(n - 1), ih(n - 1), w1(n, 0.0), w2(n, 0.0), sq(nq);
  std::vector<int> kq(nq);
  for (int i = 0; i + 1 < n; ++i) {
    h[i] = x[i + 1] - x[i];
    ih[i] = 1.0 / h[i];
  }
  for (int k = 1; k + 1 < n; ++k) {  // pchip interior weights (MATLAB pchipslopes)
    const double hs = h[k - 1] + h[k];
    w1[k] = (h[k - 1] + hs) / (3 * hs);
    w2[k] = (hs + h[k]) / (3 * hs);
  }
  for (int j = 0; j < nq; ++j) {
    const double q = tq[j];
    int lo = 0, hi = n - 1;
    if (q <= x[0])
      lo = 0;
    else if (q >= x[n - 1])
      lo = n - 2;
    else
      while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        if (x[mid] <= q)
          lo = mid;
        else
          hi = mid;
      }
    if (method == OCS_INTERP_PREVIOUS) lo = q < x[0] ? -1 : (q >= x[n - 1] ? n - 1 : lo);
    if (method == OCS_INTERP_NEAREST || method == OCS_INTERP_NEXT) lo = interp_sample_index(method, n, x, q);
    kq[j] = lo;
    sq[j] = lo >= 0 && lo < n - 1 ? q - x[lo] : 0.0;
  }
  // The seven small tables in ONE device buffer that the calling thread keeps between calls (grow-only, per device): seven
  // hipMalloc / hipFree pairs per call were most of the call's 250-450 us at batch 4096.  One staged host array, one copy.
  // 'pchip': the query points sorted by interval (counting sort), so that a thread evaluates all points of its intervals from
  // one set of node slopes (k_interp_pchip_sorted): QS [n] offsets per interval, QI [nq] original indices, SS [nq] coordinates
  const bool sorted = method == OCS_INTERP_PCHIP;
  std::vector<int> qs, qi;
  std::vector<double> ss;
  if (sorted) {
    qs.assign((size_t)n, 0);
    for (int j = 0; j < nq; ++j) ++qs[kq[j] + 1];
    for (int k = 1; k < n; ++k) qs[k] += qs[k - 1];
    qi.resize(nq);
    ss.resize(nq);
    std::vector<int> at(qs.begin(), qs.end() - 1);
    for (int j = 0; j < nq; ++j) {
      const int pos = at[kq[j]]++;
      qi[pos] = j;
      ss[pos] = sq[j];
    }
  }
  const size_t nd = (size_t)n * 5 + (size_t)nq * 2 + 8 + (sorted ? (size_t)n + 2 * (size_t)nq + 8 : 0);
  // doubles: TN n | HN n | W1 n | W2 n | IH n | SQ nq | KQ (ints) nq [| SS nq | QS (ints) n | QI (ints) nq]
  std::vector<double> stage(nd, 0.0);
  double* hp = stage.data();
  memcpy(hp, x, sizeof(double) * n);
  memcpy(hp + n, h.data(), sizeof(double) * (n - 1));
  memcpy(hp + 2 * (size_t)n, w1.data(), sizeof(double) * n);
  memcpy(hp + 3 * (size_t)n, w2.data(), sizeof(double) * n);
  memcpy(hp + 4 * (size_t)n, ih.data(), sizeof(double) * (n - 1));
  memcpy(hp + 5 * (size_t)n, sq.data(), sizeof(double) * nq);
  memcpy(hp + 5 * (size_t)n + nq, kq.data(), sizeof(int) * nq);
  const size_t oSS = 5 * (size_t)n + 2 * (size_t)nq + 8, oQS = oSS + nq, oQI = oQS + (size_t)(n + 1) / 2 + 1;
  if (sorted) {
    memcpy(hp + oSS, ss.data(), sizeof(double) * nq);
    memcpy(hp + oQS, qs.data(), sizeof(int) * n);
    memcpy(hp + oQI, qi.data(), sizeof(int) * nq);
  }
  struct Scratch {
    DevBuf buf;
    int device = -1;
  };
  static thread_local Scratch scratch;
  int devnow = 0;
  HIP_TRY(hipGetDevice(&devnow));
  if (scratch.device != devnow) {   // (a buffer of another device: dropped, not freed from here)
    scratch.buf = DevBuf();
    scratch.device = devnow;
  }
  OCS_TRY(scratch.buf.ensure(sizeof(double) * nd));
  HIP_TRY(hipMemcpyAsync(scratch.buf.p, hp, sizeof(double) * nd, hipMemcpyHostToDevice, s));
  double* dp = scratch.buf.d();
  const FbsTables tb{n, dp, dp + n, dp + 2 * (size_t)n, dp + 3 * (size_t)n, nullptr, dp + 4 * (size_t)n, nullptr};
  // ('nearest' and 'next' pick a sample like 'previous' does: the kernel's sample-index mode)
  const int kmethod = (method == OCS_INTERP_NEAREST || method == OCS_INTERP_NEXT) ? OCS_INTERP_PREVIOUS : method;
  if (sorted)
    LAUNCH_TRY(launch_interp_pchip_sorted(tb, nComp, (const int*)(dp + oQS), (const int*)(dp + oQI), dp + oSS, batch, v, out, s));
  else
    LAUNCH_TRY(launch_interp(kmethod, tb, nComp, nq, (const int*)(dp + 5 * (size_t)n + nq), dp + 5 * (size_t)n, batch, v, out, s));
  HIP_TRY(hipStreamSynchronize(s));   // (the staged host array lives until here)
  return OCS_OK;
}

int ocs_fbs_default_options(ocs_fbs_options* o) {
  if (!o) return fail(OCS_ERR_INVALID, "null argument");
  o->uRelTol = 1e-7;      // fb_sweep.m:16
  o->uAbsTol = 1e-7;      // :17
  o->nSWEEPS = 50;        // :20
  o->nERROR_PTS = 1001;   // :21
  o->nINTERP_PTS = 1001;  // :22
  o->fused_update_off = 0;
  o->nWINDOWS = 0;
  o->cost_row = 0;
  o->uRelax = 0.0;
  return OCS_OK;
}

// The sweep kernels (costate, control update, ControlChar) are templates over the problem functor; the build-defined
// LQ problem lives in the matrix-core kernels only (ocs_lq_kernels.hip).  For the sweep it is handed on as the SAME
// problem written as plugin source -- F, dFdx_times_vec, dFdu_times_vec of ocs_oracle.c / ocs_lq_kernels.hip and the
// Gen-1 ControlChar of the A9 adapter, u = clamp(-Bu' lam e^{rt} / (2 R), bounds) (make_from_symbolic.m:19-23,111) --
// compiled once per handle with hipRTC.  Parameter block [r | A | Bu | q | rdiag] as in the registry problem.
static std::string lq_plugin_source(int nS, int nC) {
  const int oA = 1, oB = 1 + nS * nS, oq = oB + nS * nC, oR = oq + nS;
  char buf[4096];
  snprintf(buf, sizeof(buf), R"SRC(
__device__ void ocs_F(double t, const double* y, const double* u, OCS_PARAMS p, double* f) {
  double s = 0.0;
  for (int i = 0; i < %d; ++i) {
    double a = 0.0;
    for (int l = 0; l < %d; ++l) a += p[%d + i + %d * l] * y[l];
    for (int l = 0; l < %d; ++l) a += p[%d + i + %d * l] * u[l];
    f[i] = a;
    s += p[%d + i] * (y[i] * y[i]);
  }
  for (int l = 0; l < %d; ++l) s += p[%d + l] * (u[l] * u[l]);
  f[%d] = exp(-p[0] * t) * s;
}
__device__ void ocs_dFdx_times_vec(double t, const double* y, const double* u, OCS_PARAMS p, const double* v, double* g) {
  const double e = exp(-p[0] * t);
  for (int i = 0; i < %d; ++i) {
    double a = 0.0;
    for (int l = 0; l < %d; ++l) a += p[%d + l + %d * i] * v[l];
    g[i] = a + 2 * e * p[%d + i] * y[i] * v[%d];
  }
}
__device__ void ocs_dFdu_times_vec(double t, const double* y, const double* u, OCS_PARAMS p, const double* v, double* g) {
  const double e = exp(-p[0] * t);
  for (int l = 0; l < %d; ++l) {
    double a = 0.0;
    for (int i = 0; i < %d; ++i) a += p[%d + i + %d * l] * v[i];
    g[l] = a + 2 * e * p[%d + l] * u[l] * v[%d];
  }
}
__device__ void ocs_ControlChar(double t, const double* x, const double* lam, OCS_PARAMS p, const double* lb,
                                const double* ub, double* u) {
  const double e = exp(p[0] * t);
  for (int l = 0; l < %d; ++l) {
    double a = 0.0;
    for (int i = 0; i < %d; ++i) a += p[%d + i + %d * l] * lam[i];
    u[l] = fmin(ub[l], fmax(lb[l], -a * e / (2 * p[%d + l])));
  }
}
)SRC",
           nS, nS, oA, nS, nC, oB, nS, oq, nC, oR, nS,           // F
           nS, nS, oA, nS, oq, nS,                                 // dFdx
           nC, nS, oB, nS, oR, nS,                                 // dFdu
           nC, nS, oB, nS, oR);                                    // ControlChar
  return std::string(buf);
}
// the problem the sweep kernels run: `p` itself, or for OCS_PROBLEM_LQ its plugin-source twin
static int sweep_problem(ocs_problem_s* p, ocs_problem_s** out) {
  *out = p;
  if (p->functor != Functor::LQ) return OCS_OK;
  if (p->pmask) return fail(OCS_ERR_UNSUPPORTED, "fb_sweep on the LQ problem: per-trajectory parameters are not supported");
  // Built once: the generated source depends on (nS, nC) only, and the parameter block and bounds of a registry problem
  // never change after creation (a version bump of `p` comes from setting or clearing per-trajectory parameters, which
  // this path rejects above) -- recompiling with hipRTC on every bump cost seconds for nS = 32.  The shadow has its own
  // version (globally unique, ocs_handles.hpp next_version), which is what the integrator's cached tables are keyed on:
  // a freed shadow's address may be reused, its version is not.
  if (!p->shadow) {
    const std::string src = lq_plugin_source(p->nS, p->nC);
    ocs_problem q = nullptr;
    const int rc = ocs_problem_create_from_source(&q, src.c_str(), p->nS, p->nC, p->par.data(), (int)p->par.size(),
                                                  p->bounds.data(), 1);
    if (rc != OCS_OK) return rc;
    p->shadow = q;
    p->shadow_version = p->version;
  }
  *out = p->shadow;
  return OCS_OK;
}

// value = ControlChar(t, x, lam): the Gen-1 problem method of make_from_symbolic.m:33-38 with the clamp of :111, which fb_sweep
// evaluates on pchip x, lam (fb_sweep.m:96, 123); host pointers, MATLAB shapes: t k, x and lam nS x k, out nC x k.
int ocs_problem_ControlChar(ocs_problem p, int k, const double* t, const double* x, const double* lam, double* out) {
  OCS_TRACE("ocs_problem_ControlChar");
  if (!p || !t || !x || !lam || !out || k < 1) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(sweep_problem(p, &p));   // (the LQ problem: its plugin twin)
  if (p->functor == Functor::User && !(p->user && p->user->has_cc))
    return fail(OCS_ERR_UNSUPPORTED, "the problem's source defines no ocs_ControlChar");
  if (p->pmask) return fail(OCS_ERR_UNSUPPORTED, "ControlChar: per-trajectory parameters have no meaning for k free columns");
  OCS_TRY(upload_problem(p));
  const int nS = p->nS, nC = p->nC;
  DevBuf dt, dx, dl, dout;
  auto body = [&]() -> int {
    OCS_TRY(dt.ensure(sizeof(double) * k));
    OCS_TRY(dx.ensure(sizeof(double) * (size_t)nS * k));
    OCS_TRY(dl.ensure(sizeof(double) * (size_t)nS * k));
    OCS_TRY(dout.ensure(sizeof(double) * (size_t)nC * k));
    HIP_TRY(hipMemcpy(dt.p, t, sizeof(double) * k, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dx.p, x, sizeof(double) * (size_t)nS * k, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dl.p, lam, sizeof(double) * (size_t)nS * k, hipMemcpyHostToDevice));
    LAUNCH_TRY(launch_eval(describe(p), 3, k, dt.d(), dx.d(), nullptr, dl.d(), dout.d(), nullptr));
    HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * (size_t)nC * k, hipMemcpyDeviceToHost));
    return OCS_OK;
  };
  const int rc = body();
  dt.release();
  dx.release();
  dl.release();
  dout.release();
  return rc;
}

// [x, lam] = compute_x_lam(prob, x0, tspan, u, ...) / [x, lam, J] = compute_x_lam_J(...) on the grid.
// device: x0 [nS][B], ugrid [2N+1][nC][B] -> xaug [N+1][nAug][B] (states + running objective, the
// augmented system of compute_x_lam_J.m:6-15), lam [N+1][nS][B], J [B] (may be NULL).
int ocs_compute_x_lam_dev(ocs_integrator g, ocs_problem p, int batch, const double* x0, const double* ugrid,
                          double* xaug, double* lam, double* J, void* stream) {
  OCS_TRACE("ocs_compute_x_lam_dev");
  if (!g || !p || !x0 || !ugrid || !xaug || !lam || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  if (g->kind != 0) return fail(OCS_ERR_UNSUPPORTED, "compute_x_lam needs an RK4Integrator grid");
  OCS_TRY(sweep_problem(p, &p));
  hipStream_t s = (hipStream_t)stream;
  OCS_TRY(bind_problem(g, p, batch, s));
  OCS_TRY(ensure_tables(g));
  ocs_fbs_state* f = g->fbs;
  const int N = g->N, nS = p->nS, nAug = nS + 1;
  const size_t B = (size_t)batch;
  OCS_TRY(f->xmid.ensure(sizeof(double) * (size_t)N * nS * B));
  double* Jd = J;
  if (!Jd) {
    OCS_TRY(f->J.ensure(sizeof(double) * B));
    Jd = f->J.d();
  }
  LAUNCH_TRY(launch_forward(describe(p), describe(g), batch, x0, ugrid, xaug, Jd, FwdOpts(), s));
  const FbsTables tb = tabs(g);
  if (costate_forms_midpoints(describe(p), N, batch)) {  // pchip midpoints of x inside the costate kernel
    LAUNCH_TRY(launch_costate(describe(p), describe(g), batch, xaug, nAug, nullptr, ugrid, nullptr, 0, lam, s, 0, tb.PR));
    return OCS_OK;
  }
  LAUNCH_TRY(launch_pchip_mid(tb, nS, nAug, batch, xaug, f->xmid.d(), s));
  LAUNCH_TRY(launch_costate(describe(p), describe(g), batch, xaug, nAug, f->xmid.d(), ugrid, nullptr, 0, lam, s));
  return OCS_OK;
}

// soln = fb_sweep(prob, x0, tspan, options)   functions/fb_sweep.m:1-126 on the grid of `g`.
// device: x0 [nS][B]; u0grid [2N+1][nC][B] / u0err [nERR][nC][B] or NULL (lower bound, :23);
// outputs xaug [N+1][nAug][B], lam [N+1][nS][B], uInterp [nINTERP][nC][B], J [B],
// sweeps int[B] (sweep index at which the instance converged, 0 = never: soln stays empty, :77),
// maxChange [nSWEEPS][B] or NULL (the per-sweep value printed at :109; NaN where not run).
// Returns OCS_NUM_NOT_CONVERGED if any instance used all nSWEEPS sweeps.
int ocs_fb_sweep_dev(ocs_integrator g, ocs_problem p, int batch, const double* x0, const ocs_fbs_options* opt,
                     const double* u0grid, const double* u0err, double* xaug, double* lam, double* uInterp,
                     double* J, int* sweeps, double* maxChange, void* stream) {
  OCS_TRACE("ocs_fb_sweep_dev");
  if (!g || !p || !x0 || !opt || !xaug || !lam || !uInterp || !J || !sweeps || batch < 1)
    return fail(OCS_ERR_INVALID, "bad argument");
  if (g->kind != 0) return fail(OCS_ERR_UNSUPPORTED, "fb_sweep needs an RK4Integrator grid");
  OCS_TRY(sweep_problem(p, &p));
  if (p->user && !p->user->has_cc)
    return fail(OCS_ERR_UNSUPPORTED, "fb_sweep needs ocs_ControlChar in the user problem source (has_control_char)");
  if (opt->nSWEEPS < 1 || opt->nERROR_PTS < 2 || opt->nINTERP_PTS < 2) return fail(OCS_ERR_INVALID, "bad options");
  if ((u0grid == nullptr) != (u0err == nullptr)) return fail(OCS_ERR_INVALID, "give both u0grid and u0err or neither");
  hipStream_t s = (hipStream_t)stream;
  OCS_TRY(bind_problem(g, p, batch, s));
  OCS_TRY(ensure_tables(g));
  ocs_fbs_state* f = g->fbs;
  const int N = g->N, nS = p->nS, nC = p->nC, nAug = nS + 1, nT = 2 * N + 1;
  const int nE = opt->nERROR_PTS, nI = opt->nINTERP_PTS;
  const size_t B = (size_t)batch;
  const int ntu = std::max(1, functor_ntu(p->functor, p->nS));
  bool newpts = false;
  if (f->nerr != nE) {
    OCS_TRY(build_points(g, nE, f->KE, f->SE, f->TE, &f->err_on_nodes, &f->QSE));
    f->nerr = nE;
    newpts = true;
  }
  if (f->nint != nI) {
    OCS_TRY(build_points(g, nI, f->KI, f->SI, f->TI));
    f->nint = nI;
    newpts = true;
  }
  if (newpts || f->tu_prob != p || f->tu_version != p->version) {
    OCS_TRY(f->TUE.ensure(sizeof(double) * (size_t)nE * ntu));
    OCS_TRY(f->TUI.ensure(sizeof(double) * (size_t)nI * ntu));
    LAUNCH_TRY(launch_tu_at(describe(p), nE, f->TE.d(), f->TUE.d(), s));
    LAUNCH_TRY(launch_tu_at(describe(p), nI, f->TI.d(), f->TUI.d(), s));
    f->tu_prob = p;
    f->tu_version = p->version;
  }
  const size_t ugridN = (size_t)nT * nC * B, uerrN = (size_t)nE * nC * B;
  OCS_TRY(f->xmid.ensure(sizeof(double) * (size_t)N * nS * B));
  OCS_TRY(f->ugrid.ensure(sizeof(double) * ugridN));
  OCS_TRY(f->uerr.ensure(sizeof(double) * uerrN));
  OCS_TRY(f->usel.ensure(sizeof(int) * B));
  OCS_TRY(f->dump.ensure(sizeof(double) * B));
  OCS_TRY(f->nactive.ensure(sizeof(int)));
  // Error points on the grid nodes and the default start: the control at the error points is the node samples of the
  // grid control, so the weighted change is taken while the grid control is replaced (one kernel, one pass over x and
  // lam) instead of in a separate error-point kernel with its own copy of the control.
  // damped update (extension, ocs.h): u = u + om (uNew - u); om = 1 is the reference's u = uNew
  if (opt->uRelax < 0.0 || opt->uRelax > 1.0) return fail(OCS_ERR_INVALID, "uRelax must be in [0, 1]");
  const double om = opt->uRelax > 0.0 ? opt->uRelax : 1.0;
  const int fuo = opt->fused_update_off == 3 ? 0 : opt->fused_update_off;   // 3: as 0, without the fold below
  const bool fusedup = f->err_on_nodes && !u0grid && fuo != 1;
  // error points off the nodes: by runs of intervals where the sorted kernel applies (registry problems), else point by point
  static const bool cps_off = getenv("OCS_CONTROL_PTS_SORTED") && getenv("OCS_CONTROL_PTS_SORTED")[0] == '0';
  const bool cps = !fusedup && !cps_off && control_pts_sorted_ok(describe(p)) && f->QSE.p;
  const int nparts = fusedup ? control_grid_parts(N) : (cps ? control_pts_run_parts(N) : control_pts_parts(nE));
  OCS_TRY(f->metric.ensure(sizeof(double) * (size_t)nparts * B));
  OCS_TRY(f->anyvalid.ensure(sizeof(int) * B));
  double* mc = maxChange;
  if (!mc) {
    OCS_TRY(f->maxchange.ensure(sizeof(double) * (size_t)opt->nSWEEPS * B));
    mc = f->maxchange.d();
  }
  int* status = sweeps;
  LAUNCH_TRY(launch_fbs_init(batch, opt->nSWEEPS, (int*)f->usel.p, status, mc, s));  // usel = 0, status = 0, maxChange = NaN
  const ProblemDesc pd = describe(p);
  const GridDesc gd = describe(g);
  const FbsTables tb = tabs(g);
  // every sweep with the control update folded into the state pass (see below): the grid samples of u are never formed
  // (not with a damped update: that needs the samples of the control it damps)
  // (registry problems: where the wave-specialised costate kernels and the gated state pass apply; hipRTC problems: row
  //  functions whose ocs_ControlChar reads the costate alone, fold_supported)
  const bool userfold = p->user != nullptr && fold_supported(pd, gd, batch);
  const bool fold = fusedup && om == 1.0 && opt->nWINDOWS <= 1 && opt->fused_update_off == 0 &&
                    (userfold || (costate_forms_midpoints(pd, N, batch) && forward_gate_supported(pd, gd, batch) &&
                                  fold_supported(pd, gd, batch)));
  if (u0grid) {  // u = u0  :76
    HIP_TRY(hipMemcpyAsync(f->ugrid.p, u0grid, sizeof(double) * ugridN, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(f->uerr.p, u0err, sizeof(double) * uerrN, hipMemcpyDeviceToDevice, s));
  } else if (!fold) {  // u0 = ControlBounds(:,1)*ones(1,length(t))  :23
    LAUNCH_TRY(launch_fill_rows(nT, nC, batch, p->d_lb.d(), f->ugrid.d(), s));
    // (with the change measured on the grid nodes the separate error-point samples are never read)
    if (!fusedup) LAUNCH_TRY(launch_fill_rows(nE, nC, batch, p->d_lb.d(), f->uerr.d(), s));
  }
  const int* usel = (const int*)f->usel.p;  // selects the old / new buffer of the ERROR-POINT samples only
  int nactive = batch;
  f->last_path = 1;
  // ---- fused update, several windows of the batch on their own streams -----------------------------------------
  // The two marching kernels of a sweep (forward, costate) are latency-bound and leave most of the GPU idle; the
  // streaming kernels (pchip, control update) are HBM-bound.  Independent windows of the batch, each running its own
  // sweep loop on its own stream and started one forward pass apart, let one window's marching kernels run under
  // another's streaming kernels.  Each window stops on its own count of active instances.
  int nwin = 1;
  if (fusedup) {
    // Automatic = 1 for now: measured at batch 16384, 4 windows are host-bound (8 runtime calls per window and sweep,
    // ~190 us, against ~290 us of GPU time per round) and slower than one stream; a captured graph per window is
    // the missing piece.  At batch 65536 windows neither gain nor lose (the streaming kernels dominate).
    nwin = opt->nWINDOWS > 0 ? opt->nWINDOWS : 1;
    while (nwin > 1 && (batch + nwin - 1) / nwin < 64) --nwin;
  }
  if (fusedup && nwin > 1) {
    f->last_path = 3;
    const int W = (((batch + nwin - 1) / nwin + 63) / 64) * 64;  // whole tiles of the pipeline kernels
    nwin = (batch + W - 1) / W;
    const int nsw = opt->nSWEEPS;
    while ((int)f->wstreams.size() < nwin) {
      hipStream_t st;
      HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      f->wstreams.push_back(st);
      for (int e = 0; e < 2; ++e) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        f->wevents.push_back(ev);
      }
    }
    if (!f->fork) HIP_TRY(hipEventCreateWithFlags(&f->fork, hipEventDisableTiming));
    if (!f->stag) HIP_TRY(hipEventCreateWithFlags(&f->stag, hipEventDisableTiming));
    if (f->h_nact_cap < nwin * nsw) {
      if (f->h_nact) (void)hipHostFree(f->h_nact);
      f->h_nact = nullptr;
      HIP_TRY(hipHostMalloc((void**)&f->h_nact, sizeof(int) * (size_t)nwin * nsw));
      f->h_nact_cap = nwin * nsw;
    }
    OCS_TRY(f->nact_slots.ensure(sizeof(int) * (size_t)nwin * nsw));
    HIP_TRY(hipMemsetAsync(f->nact_slots.p, 0, sizeof(int) * (size_t)nwin * nsw, s));
    HIP_TRY(hipEventRecord(f->fork, s));
    std::vector<int> done(nwin, 0), last(nwin, 0), act(nwin, 0);  // done: finished; last: sweeps enqueued
    int* dslots = (int*)f->nact_slots.p;
    auto enqueue = [&](int j, int sweep) -> int {
      const int off = j * W, cnt = std::min(W, batch - off);
      hipStream_t st = f->wstreams[j];
      ProblemDesc pj = pd;
      if (pj.pb) pj.pb += off;
      FwdOpts fo;
      fo.frozen = status + off;
      fo.dump = f->dump.d() + off;
      fo.ld = batch;
      fo.no_cost_row = opt->cost_row == 0;
      LAUNCH_TRY(launch_forward(pj, gd, cnt, x0 + off, f->ugrid.d() + off, xaug + off, J + off, fo, st));
      if (sweep == 1 && j + 1 < nwin) HIP_TRY(hipEventRecord(f->stag, st));  // the next window starts one pass later
      LAUNCH_TRY(launch_pchip_mid(tb, nS, nAug, cnt, xaug + off, f->xmid.d() + off, st, batch));
      LAUNCH_TRY(launch_costate(pj, gd, cnt, xaug + off, nAug, f->xmid.d() + off, f->ugrid.d() + off, status + off,
                                f->dump.d() + off, lam + off, st, batch));
      LAUNCH_TRY(launch_control_grid(pj, gd, tb, cnt, xaug + off, nAug, f->xmid.d() + off, lam + off, f->ugrid.d() + off,
                                     status + off, f->metric.d() + off, opt->uRelTol, opt->uAbsTol, st, batch, nullptr, om));
      int* slot = dslots + (size_t)j * nsw + (sweep - 1);
      LAUNCH_TRY(launch_fbs_advance(cnt, sweep, nparts, f->metric.d() + off, (int*)f->anyvalid.p + off,
                                    (int*)f->usel.p + off, status + off, mc + off, slot, st, batch));
      HIP_TRY(hipMemcpyAsync(f->h_nact + (size_t)j * nsw + (sweep - 1), slot, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventRecord(f->wevents[2 * j + (sweep & 1)], st));
      return OCS_OK;
    };
    for (int j = 0; j < nwin; ++j) {  // sweep 1, staggered
      HIP_TRY(hipStreamWaitEvent(f->wstreams[j], f->fork, 0));
      if (j > 0) HIP_TRY(hipStreamWaitEvent(f->wstreams[j], f->stag, 0));
      OCS_TRY(enqueue(j, 1));
      last[j] = 1;
    }
    for (int live = nwin; live > 0;) {  // round robin: a window gets its next sweep when the previous one is back
      for (int j = 0; j < nwin; ++j) {
        if (done[j]) continue;
        HIP_TRY(hipEventSynchronize(f->wevents[2 * j + (last[j] & 1)]));
        act[j] = f->h_nact[(size_t)j * nsw + (last[j] - 1)];
        if (act[j] == 0 || last[j] >= nsw) {
          done[j] = 1;
          --live;
          continue;
        }
        ++last[j];
        OCS_TRY(enqueue(j, last[j]));
      }
    }
    nactive = 0;
    for (int j = 0; j < nwin; ++j) nactive += act[j];
  }
  // ---- fused update, one stream, sweeps enqueued one ahead --------------------------------------------------------
  // The host learns the number of active instances of sweep k only after a round trip; waiting for it before
  // enqueuing sweep k+1 left the GPU idle for ~25 us per sweep.  Here sweep k+1 is enqueued first and its kernels
  // start with a look at sweep k's device counter: if no instance was left, they return at once (a sweep over
  // converged instances would change nothing anyway -- they are frozen -- but would cost its full time).
  bool spec_done = false;
  // (every state pass of the sweep takes the gate -- the wave-specialised kernels, split passes, the lane kernel of any
  //  plugin -- so the loop is the same for every problem: what differs is whether the pchip midpoints of x are a kernel
  //  of their own)
  // ... and the same for error points OFF the grid nodes (the reference's default of 1001 points on any grid but N = 1000, a given
  // u0): the kernels of the kernel-by-kernel sequence below, gated and enqueued ahead (path 5).  That sequence waited for the
  // host once per sweep: 485 against ~250 us per sweep at 500 steps.
  const bool ahead1 = !fusedup && fuo == 0 && opt->nWINDOWS <= 1 && forward_gate_any(pd);
  if ((fusedup && nwin == 1 && (fold || forward_gate_any(pd))) || ahead1) {
    const int nsw = opt->nSWEEPS;
    const bool ownx = fuo == 0 && costate_forms_midpoints(pd, N, batch);   // midpoints inside the costate / control kernels
    f->last_path = fold ? 4 : (fusedup ? 2 : 5);
    if (f->h_nact_cap < nsw) {
      if (f->h_nact) (void)hipHostFree(f->h_nact);
      f->h_nact = nullptr;
      HIP_TRY(hipHostMalloc((void**)&f->h_nact, sizeof(int) * (size_t)nsw));
      f->h_nact_cap = nsw;
    }
    // How many sweeps are enqueued beyond the one the host waits for.  The gate of a sweep is read on the device in stream
    // order, so any depth is correct; a sweep enqueued after the last live one costs two kernels that return at once.  With
    // depth 1 the host had one sweep (~180 us) to wake up, read the count and enqueue the next: enough on an idle host, not on
    // a busy one (a late host leaves the GPU idle between dependent kernels); depth 2 gives it two.  OCS_FBS_DEPTH overrides.
    static const int depth_env = [] {
      const char* e = getenv("OCS_FBS_DEPTH");
      const int d = e ? atoi(e) : 0;
      return d >= 1 && d <= 8 ? d : 2;
    }();
    const int depth = std::min(depth_env, std::max(1, nsw - 1));
    const int nev = depth + 1;
    while ((int)f->wevents.size() < nev) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      f->wevents.push_back(ev);
    }
    OCS_TRY(f->nact_slots.ensure(sizeof(int) * (size_t)nsw));
    HIP_TRY(hipMemsetAsync(f->nact_slots.p, 0, sizeof(int) * (size_t)nsw, s));
    int* dslots = (int*)f->nact_slots.p;
    // Sweeps >= 2 with the control update folded into the state pass (ocs_fold_kernel.hpp): the control a sweep
    // integrates is ControlChar of the costate of the sweep before, formed inside the pass; the costate pass measures
    // the change of the control its new costate implies and takes the convergence decision.  Two kernels per sweep,
    // and the grid samples of u are neither written nor read.
    // The first sweep as well: both kernels take the default start u0 = lower bound (:23) in place of ControlChar of
    // the costate of the sweep before (which does not exist yet; what lam holds then is read and dropped).
    auto enqueue = [&](int sweep) -> int {
      const int* gate = sweep > 1 ? dslots + (sweep - 2) : nullptr;  // active instances after the sweep before
      if (fold) {
        LAUNCH_TRY(launch_forward_cc(pd, gd, batch, tb.PR, p->d_lb.d(), p->d_ub.d(), x0, lam, xaug, J, status,
                                     opt->cost_row == 0, gate, sweep == 1, s));
        LAUNCH_TRY(launch_costate_met(pd, gd, batch, xaug, nAug, tb.PR, p->d_lb.d(), p->d_ub.d(), opt->uRelTol,
                                      opt->uAbsTol, sweep, status, mc, dslots + (sweep - 1), lam, s, gate));
        HIP_TRY(hipMemcpyAsync(f->h_nact + (sweep - 1), dslots + (sweep - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(f->wevents[sweep % nev], s));
        return OCS_OK;
      }
      FwdOpts fo;
      fo.frozen = status;
      fo.dump = f->dump.d();
      fo.no_cost_row = opt->cost_row == 0;
      fo.gate = gate;
      LAUNCH_TRY(launch_forward(pd, gd, batch, x0, f->ugrid.d(), xaug, J, fo, s));
      const double* xmid = ownx ? nullptr : f->xmid.d();
      if (!ownx) LAUNCH_TRY(launch_pchip_mid(tb, nS, nAug, batch, xaug, f->xmid.d(), s, 0, gate));
      LAUNCH_TRY(launch_costate(pd, gd, batch, xaug, nAug, xmid, f->ugrid.d(), status, f->dump.d(), lam, s, 0, tb.PR,
                                gate));
      if (!fusedup) {   // error points off the nodes: uNew there with check_convergence (:96, :99-115), then u = uNew on the grid
        if (cps)
          LAUNCH_TRY(launch_control_pts_sorted(pd, tb, nE, (const int*)f->QSE.p, f->SE.d(), f->TUE.d(), batch, xaug, nAug, lam,
                                               f->uerr.d(), f->metric.d(), opt->uRelTol, opt->uAbsTol, s, om, gate));
        else
          LAUNCH_TRY(launch_control_pts(pd, tb, nE, (const int*)f->KE.p, f->SE.d(), f->TUE.d(), batch, xaug, nAug, lam,
                                        f->uerr.d(), usel, (long long)uerrN, f->metric.d(), (int*)f->anyvalid.p, opt->uRelTol,
                                        opt->uAbsTol, s, om, gate));
        LAUNCH_TRY(launch_fbs_advance(batch, sweep, nparts, f->metric.d(), (int*)f->anyvalid.p, (int*)f->usel.p, status,
                                      mc, dslots + (sweep - 1), s, 0, gate));
        // (only the instances that continue take uNew: status is the one k_fbs_advance just wrote, :85 / :82)
        LAUNCH_TRY(launch_control_grid(pd, gd, tb, batch, xaug, nAug, xmid, lam, f->ugrid.d(), status, nullptr, 0.0, 0.0, s, 0,
                                       gate, om));
        HIP_TRY(hipMemcpyAsync(f->h_nact + (sweep - 1), dslots + (sweep - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(f->wevents[sweep % nev], s));
        return OCS_OK;
      }
      LAUNCH_TRY(launch_control_grid(pd, gd, tb, batch, xaug, nAug, xmid, lam, f->ugrid.d(), status, f->metric.d(),
                                     opt->uRelTol, opt->uAbsTol, s, 0, gate, om));
      LAUNCH_TRY(launch_fbs_advance(batch, sweep, nparts, f->metric.d(), (int*)f->anyvalid.p, (int*)f->usel.p, status,
                                    mc, dslots + (sweep - 1), s, 0, gate));
      HIP_TRY(hipMemcpyAsync(f->h_nact + (sweep - 1), dslots + (sweep - 1), sizeof(int), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipEventRecord(f->wevents[sweep % nev], s));
      return OCS_OK;
    };
    for (int k = 1; k <= std::min(depth, nsw); ++k) OCS_TRY(enqueue(k));
    for (int sweep = 1; sweep <= nsw; ++sweep) {
      if (sweep + depth <= nsw) OCS_TRY(enqueue(sweep + depth));
      HIP_TRY(hipEventSynchronize(f->wevents[sweep % nev]));
      nactive = f->h_nact[sweep - 1];
      if (nactive == 0) break;  // the sweeps already enqueued find their gates closed
    }
    spec_done = true;
  }
  for (int sweep = 1; !spec_done && !(fusedup && nwin > 1) && sweep <= opt->nSWEEPS && nactive > 0; ++sweep) {  // :79
    // uNew = sweep(u): compute_x_lam (:95) ...
    // instances that converged in an earlier sweep are integrated along but store nothing: their x, lam, J stay
    // those of the sweep they converged in (final_sweep(u), :82)
    FwdOpts fo;
    fo.frozen = status;
    fo.dump = f->dump.d();
    fo.no_cost_row = opt->cost_row == 0;  // soln holds x, lam, u and the scalar J (fb_sweep.m:117-125)
    LAUNCH_TRY(launch_forward(pd, gd, batch, x0, f->ugrid.d(), xaug, J, fo, s));
    // pchip midpoints of x: inside the costate and control kernels where the wave-specialised costate kernel applies
    const bool ownx = fusedup && fuo == 0 && costate_forms_midpoints(pd, N, batch);
    const double* xmid = ownx ? nullptr : f->xmid.d();
    if (!ownx) LAUNCH_TRY(launch_pchip_mid(tb, nS, nAug, batch, xaug, f->xmid.d(), s));
    HIP_TRY(hipMemsetAsync(f->nactive.p, 0, sizeof(int), s));
    if (fusedup) {
      // costate (:95); uNew = ControlChar(t, x(t), lam(t)) on the grid, in place (:96, :85), with the weighted change at
      // the nodes (:107) folded in.  A just-converged instance takes uNew as well, but it is frozen from now on: its
      // x, lam, J are the ones computed above from the old control, which is what final_sweep(u) returns (:82).
      LAUNCH_TRY(launch_costate(pd, gd, batch, xaug, nAug, xmid, f->ugrid.d(), status, f->dump.d(), lam, s, 0, tb.PR));
      LAUNCH_TRY(launch_control_grid(pd, gd, tb, batch, xaug, nAug, xmid, lam, f->ugrid.d(), status, f->metric.d(),
                                     opt->uRelTol, opt->uAbsTol, s, 0, nullptr, om));
      LAUNCH_TRY(launch_fbs_advance(batch, sweep, nparts, f->metric.d(), (int*)f->anyvalid.p, (int*)f->usel.p, status,
                                    mc, (int*)f->nactive.p, s));
      HIP_TRY(hipMemcpyAsync(&nactive, f->nactive.p, sizeof(int), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      continue;
    }
    LAUNCH_TRY(launch_costate(pd, gd, batch, xaug, nAug, f->xmid.d(), f->ugrid.d(), status, f->dump.d(), lam, s));
    // ... uNew = ControlChar(t, x(t), lam(t)) (:96) on the error points, with check_convergence(uNew, u)
    // (:81, :99-115) folded in
    if (cps)
      LAUNCH_TRY(launch_control_pts_sorted(pd, tb, nE, (const int*)f->QSE.p, f->SE.d(), f->TUE.d(), batch, xaug, nAug, lam,
                                           f->uerr.d(), f->metric.d(), opt->uRelTol, opt->uAbsTol, s, om));
    else
      LAUNCH_TRY(launch_control_pts(pd, tb, nE, (const int*)f->KE.p, f->SE.d(), f->TUE.d(), batch, xaug, nAug, lam,
                                    f->uerr.d(), usel, (long long)uerrN, f->metric.d(),
                                    (int*)f->anyvalid.p, opt->uRelTol, opt->uAbsTol, s, om));
    LAUNCH_TRY(launch_fbs_advance(batch, sweep, nparts, f->metric.d(), (int*)f->anyvalid.p,
                                  (int*)f->usel.p, status, mc, (int*)f->nactive.p, s));
    // u = uNew (:85) on the integrator grid, only for the instances that continue: a converged instance
    // keeps its OLD control, which is what final_sweep(u) integrates (:82)
    // (lam's pchip midpoints are formed inside the kernel)
    LAUNCH_TRY(launch_control_grid(pd, gd, tb, batch, xaug, nAug, f->xmid.d(), lam, f->ugrid.d(), status, nullptr, 0.0, 0.0,
                                   s, 0, nullptr, om));
    HIP_TRY(hipMemcpyAsync(&nactive, f->nactive.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  // final_sweep(u) (:82, :117-125): x, lam, J of a converged instance already belong to its OLD control
  // (a frozen instance is re-integrated with the same u each later sweep, bit for bit); what is left is
  // uOpt = ControlChar(interpPts, xOpt(interpPts), lamOpt(interpPts))  :123
  LAUNCH_TRY(launch_control_pts(pd, tb, nI, (const int*)f->KI.p, f->SI.d(), f->TUI.d(), batch, xaug, nAug, lam,
                                uInterp, nullptr, 0, nullptr, nullptr, 0.0, 0.0, s));
  return nactive > 0 ? OCS_NUM_NOT_CONVERGED : OCS_OK;
}

int ocs_fb_sweep_path(ocs_integrator g) { return (g && g->fbs) ? g->fbs->last_path : 0; }

// host: x0 nS x batch; u0grid nC x (2N+1) x batch, u0err nC x nERR x batch (or both NULL);
// x nS x (N+1) x batch, lam nS x (N+1) x batch, uInterp nC x nINTERP x batch, J batch, sweeps batch,
// maxChange nSWEEPS x batch (or NULL).
int ocs_fb_sweep(ocs_integrator g, ocs_problem p, int batch, const double* x0, const ocs_fbs_options* opt,
                 const double* u0grid, const double* u0err, double* x, double* lam, double* uInterp, double* J,
                 int* sweeps, double* maxChange) {
  OCS_TRACE("ocs_fb_sweep");
  if (!g || !p || !x0 || !opt || !x || !lam || !uInterp || !J || !sweeps || batch < 1)
    return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(upload_grid(g));
  OCS_TRY(ensure_tables(g));
  ocs_fbs_state* f = g->fbs;
  const int N = g->N, nS = p->nS, nC = p->nC, nAug = nS + 1, nT = 2 * N + 1;
  const int nE = opt->nERROR_PTS, nI = opt->nINTERP_PTS;
  const size_t B = (size_t)batch;
  hipStream_t s = g->stream;
  auto in = [&](const double* host, DevBuf& dst, int per) -> int {
    const size_t bytes = sizeof(double) * (size_t)per * B;
    HIP_TRY(hipStreamSynchronize(s));
    OCS_TRY(f->stage.ensure(bytes));
    OCS_TRY(dst.ensure(bytes));
    HIP_TRY(hipMemcpyAsync(f->stage.p, host, bytes, hipMemcpyHostToDevice, s));
    LAUNCH_TRY(launch_to_batch_minor(f->stage.d(), dst.d(), per, batch, s));
    return OCS_OK;
  };
  auto out = [&](const double* src, double* host, int per) -> int {
    const size_t bytes = sizeof(double) * (size_t)per * B;
    OCS_TRY(f->stage.ensure(bytes));
    LAUNCH_TRY(launch_to_traj_major(src, f->stage.d(), per, batch, s));
    HIP_TRY(hipMemcpyAsync(host, f->stage.p, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return OCS_OK;
  };
  OCS_TRY(in(x0, f->x0, nS));
  DevBuf dug, due;
  int rc = OCS_OK;
  auto body = [&]() -> int {
    if (u0grid) {
      OCS_TRY(in(u0grid, dug, nC * nT));
      OCS_TRY(in(u0err, due, nC * nE));
    }
    OCS_TRY(f->xaug.ensure(sizeof(double) * (size_t)(N + 1) * nAug * B));
    OCS_TRY(f->lam.ensure(sizeof(double) * (size_t)(N + 1) * nS * B));
    OCS_TRY(f->uint_.ensure(sizeof(double) * (size_t)nI * nC * B));
    OCS_TRY(f->J.ensure(sizeof(double) * B));
    OCS_TRY(f->status.ensure(sizeof(int) * B));
    OCS_TRY(f->maxchange.ensure(sizeof(double) * (size_t)opt->nSWEEPS * B));
    const int st = ocs_fb_sweep_dev(g, p, batch, f->x0.d(), opt, u0grid ? dug.d() : nullptr, u0grid ? due.d() : nullptr,
                                    f->xaug.d(), f->lam.d(), f->uint_.d(), f->J.d(), (int*)f->status.p,
                                    f->maxchange.d(), s);
    if (st < 0) return st;
    // drop the running-objective row: soln.x has the nS state rows
    std::vector<double> tmp((size_t)(N + 1) * nAug * B);
    OCS_TRY(out(f->xaug.d(), tmp.data(), (N + 1) * nAug));
    for (size_t b = 0; b < B; ++b)
      for (int i = 0; i <= N; ++i)
        for (int k = 0; k < nS; ++k)
          x[(b * (N + 1) + i) * nS + k] = tmp[(b * (N + 1) + i) * nAug + k];
    OCS_TRY(out(f->lam.d(), lam, (N + 1) * nS));
    OCS_TRY(out(f->uint_.d(), uInterp, nI * nC));
    HIP_TRY(hipMemcpy(J, f->J.p, sizeof(double) * B, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sweeps, f->status.p, sizeof(int) * B, hipMemcpyDeviceToHost));
    if (maxChange) {  // [nSWEEPS][B] on the device -> nSWEEPS x batch column-major on the host
      OCS_TRY(out(f->maxchange.d(), maxChange, opt->nSWEEPS));
    }
    return st;
  };
  rc = body();
  dug.release();
  due.release();
  return rc;
}

// host compute_x_lam(_J): x0 nS x batch, ugrid nC x (2N+1) x batch -> x nS x (N+1) x batch, lam same, J (or NULL)
int ocs_compute_x_lam(ocs_integrator g, ocs_problem p, int batch, const double* x0, const double* ugrid, double* x,
                      double* lam, double* J) {
  OCS_TRACE("ocs_compute_x_lam");
  if (!g || !p || !x0 || !ugrid || !x || !lam || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(upload_grid(g));
  OCS_TRY(ensure_tables(g));
  ocs_fbs_state* f = g->fbs;
  const int N = g->N, nS = p->nS, nC = p->nC, nAug = nS + 1, nT = 2 * N + 1;
  const size_t B = (size_t)batch;
  hipStream_t s = g->stream;
  DevBuf dug;
  auto body = [&]() -> int {
    const size_t bx = sizeof(double) * (size_t)nS * B, bu = sizeof(double) * (size_t)nC * nT * B;
    OCS_TRY(f->stage.ensure(std::max(bx, bu)));
    OCS_TRY(f->x0.ensure(bx));
    OCS_TRY(dug.ensure(bu));
    HIP_TRY(hipMemcpy(f->stage.p, x0, bx, hipMemcpyHostToDevice));
    LAUNCH_TRY(launch_to_batch_minor(f->stage.d(), f->x0.d(), nS, batch, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(f->stage.p, ugrid, bu, hipMemcpyHostToDevice));
    LAUNCH_TRY(launch_to_batch_minor(f->stage.d(), dug.d(), nC * nT, batch, s));
    OCS_TRY(f->xaug.ensure(sizeof(double) * (size_t)(N + 1) * nAug * B));
    OCS_TRY(f->lam.ensure(sizeof(double) * (size_t)(N + 1) * nS * B));
    OCS_TRY(f->J.ensure(sizeof(double) * B));
    OCS_TRY(ocs_compute_x_lam_dev(g, p, batch, f->x0.d(), dug.d(), f->xaug.d(), f->lam.d(), f->J.d(), s));
    std::vector<double> tmp((size_t)(N + 1) * nAug * B);
    OCS_TRY(f->stage.ensure(sizeof(double) * tmp.size()));
    LAUNCH_TRY(launch_to_traj_major(f->xaug.d(), f->stage.d(), (N + 1) * nAug, batch, s));
    HIP_TRY(hipMemcpyAsync(tmp.data(), f->stage.p, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t b = 0; b < B; ++b)
      for (int i = 0; i <= N; ++i)
        for (int k = 0; k < nS; ++k)
          x[(b * (N + 1) + i) * nS + k] = tmp[(b * (N + 1) + i) * nAug + k];
    LAUNCH_TRY(launch_to_traj_major(f->lam.d(), f->stage.d(), (N + 1) * nS, batch, s));
    HIP_TRY(hipMemcpyAsync(lam, f->stage.p, sizeof(double) * (size_t)(N + 1) * nS * B, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (J) HIP_TRY(hipMemcpy(J, f->J.p, sizeof(double) * B, hipMemcpyDeviceToHost));
    return OCS_OK;
  };
  const int rc = body();
  dug.release();
  return rc;
}

}  // extern "C"
