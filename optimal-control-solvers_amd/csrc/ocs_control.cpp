// ocs_control.cpp -- Control parametrisations (Control/Control.m and its three subclasses),
// vectorInterpolant sampling, and the single_shooting objective (single_shooting.m:137-150).
// The basis matrix is built on the host exactly once per handle, like the reference's
// constructors do; all per-trajectory arithmetic runs in the kernels.
#include "ocs_trace.hpp"
#include "ocs_handles.hpp"

#include <algorithm>

using namespace ocs;

struct ocs_control_s {
  int kind = 0, nBasis = 0, nC = 0, nT = 0;
  std::vector<double> t, pts, B;  // B: nBasis x nT column-major (property B)
  double t0 = 0, t1 = 0;
  // device copies: CSC (for u = v*B) and CSR (for dJdv = dJdu*B')
  DevBuf d_colptr, d_row, d_cval, d_rowptr, d_col, d_rval, d_BT, d_BT16, d_CT;
  bool banded = false;  // every column has at most two consecutive non-zeros and the band moves up by at most one row
  int band_r0 = 0;      // per column (PWLinear, PWConstant): column table d_CT for the fused banded kernels
  int fuse_mode = 0;  // ocs_control_set_fusion: 0 automatic, 1 never, 2 whenever the fused kernels support the case,
                      // (the wave-specialised ones where they apply, at any batch), 3 as 2 but on the lane kernels
                      // (k_forward_fc / k_backward_fc) only
  bool dense = false;  // more than half of B is non-zero and nBasis <= 32: register-resident dense kernels
  bool uploaded = false;
  int device = -1;   // HIP device that owns the handle's memory (set at the first upload)
  hipStream_t stream = nullptr;
  DevBuf d_v, d_u, d_dJdu, d_dJdv, d_stage, d_x0, d_J, d_idx;
};

// ---- host helpers (MATLAB semantics) -------------------------------------------------------
// linspace(d1, d2, n):  d1 + (0:n1).*(d2-d1)./n1 with the end points pinned
static void matlab_linspace(double a, double b, int n, double* out) {
  if (n <= 0) return;
  if (n == 1) {
    out[0] = b;
    return;
  }
  const int n1 = n - 1;
  for (int k = 0; k <= n1; ++k) out[k] = a + ((double)k * (b - a)) / (double)n1;
  out[0] = a;
  out[n1] = b;
}
static int interval_of(int n, const double* x, double q) {
  if (q <= x[0]) return 0;
  if (q >= x[n - 1]) return n - 2;
  int lo = 0, hi = n - 1;
  while (hi - lo > 1) {
    const int mid = (lo + hi) / 2;
    if (x[mid] <= q)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}
// griddedInterpolant(x, v, 'linear', 'nearest')(q)   (PWLinearControl.m:35)
static double tent_eval(int n, const double* x, const double* v, double q) {
  if (q < x[0]) return v[0];
  if (q > x[n - 1]) return v[n - 1];
  const int k = interval_of(n, x, q);
  return v[k] + (v[k + 1] - v[k]) * ((q - x[k]) / (x[k + 1] - x[k]));
}
static int sgn(double v) { return (v > 0) - (v < 0); }
// pchip slopes (Fritsch-Carlson as in MATLAB pchip / Moler's pchiptx)
static void pchip_slopes(int n, const double* x, const double* y, double* d) {
  if (n == 1) {
    d[0] = 0;
    return;
  }
  std::vector<double> h(n - 1), del(n - 1);
  for (int i = 0; i < n - 1; ++i) {
    h[i] = x[i + 1] - x[i];
    del[i] = (y[i + 1] - y[i]) / h[i];
  }
  if (n == 2) {
    d[0] = d[1] = del[0];
    return;
  }
  for (int k = 0; k < n - 2; ++k) {
    if (sgn(del[k]) * sgn(del[k + 1]) > 0) {
      const double hs = h[k] + h[k + 1];
      const double w1 = (h[k] + hs) / (3 * hs), w2 = (hs + h[k + 1]) / (3 * hs);
      const double a0 = std::fabs(del[k]), a1 = std::fabs(del[k + 1]);
      const double dmax = std::max(a0, a1), dmin = std::min(a0, a1);
      d[k + 1] = dmin / (w1 * (del[k] / dmax) + w2 * (del[k + 1] / dmax));
    } else {
      d[k + 1] = 0;
    }
  }
  d[0] = ((2 * h[0] + h[1]) * del[0] - h[0] * del[1]) / (h[0] + h[1]);
  if (sgn(d[0]) != sgn(del[0]))
    d[0] = 0;
  else if (sgn(del[0]) != sgn(del[1]) && std::fabs(d[0]) > std::fabs(3 * del[0]))
    d[0] = 3 * del[0];
  d[n - 1] = ((2 * h[n - 2] + h[n - 3]) * del[n - 2] - h[n - 2] * del[n - 3]) / (h[n - 2] + h[n - 3]);
  if (sgn(d[n - 1]) != sgn(del[n - 2]))
    d[n - 1] = 0;
  else if (sgn(del[n - 2]) != sgn(del[n - 3]) && std::fabs(d[n - 1]) > std::fabs(3 * del[n - 2]))
    d[n - 1] = 3 * del[n - 2];
}

static int upload_control(ocs_control_s* c) {
  if (c->uploaded) return OCS_OK;
  OCS_TRY(require_device());
  if (c->device < 0) c->device = current_device_or(-1);
  const int nB = c->nBasis, nT = c->nT;
  std::vector<int> colptr(nT + 1, 0), row, rowptr(nB + 1, 0), col;
  std::vector<double> cval, rval;
  for (int j = 0; j < nT; ++j) {
    for (int i = 0; i < nB; ++i) {
      const double b = c->B[i + (size_t)nB * j];
      if (b != 0.0) {
        row.push_back(i);
        cval.push_back(b);
      }
    }
    colptr[j + 1] = (int)row.size();
  }
  for (int i = 0; i < nB; ++i) {
    for (int j = 0; j < nT; ++j) {
      const double b = c->B[i + (size_t)nB * j];
      if (b != 0.0) {
        col.push_back(j);
        rval.push_back(b);
      }
    }
    rowptr[i + 1] = (int)col.size();
  }
  const size_t nnz = std::max<size_t>(row.size(), 1);
  row.resize(nnz);
  cval.resize(nnz);
  col.resize(nnz);
  rval.resize(nnz);
  OCS_TRY(c->d_colptr.ensure(sizeof(int) * colptr.size()));
  OCS_TRY(c->d_row.ensure(sizeof(int) * nnz));
  OCS_TRY(c->d_cval.ensure(sizeof(double) * nnz));
  OCS_TRY(c->d_rowptr.ensure(sizeof(int) * rowptr.size()));
  OCS_TRY(c->d_col.ensure(sizeof(int) * nnz));
  OCS_TRY(c->d_rval.ensure(sizeof(double) * nnz));
  HIP_TRY(hipMemcpy(c->d_colptr.p, colptr.data(), sizeof(int) * colptr.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_row.p, row.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_cval.p, cval.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_rowptr.p, rowptr.data(), sizeof(int) * rowptr.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_col.p, col.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_rval.p, rval.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
  size_t nz = 0;
  for (double b : c->B) nz += (b != 0.0);
  c->dense = basis_dense_supported(nB) && 2 * nz > c->B.size();
  if (c->dense) {
    std::vector<double> BT((size_t)nT * nB);
    for (int j = 0; j < nT; ++j)
      for (int i = 0; i < nB; ++i) BT[(size_t)j * nB + i] = c->B[i + (size_t)nB * j];
    OCS_TRY(c->d_BT.ensure(sizeof(double) * BT.size()));
    HIP_TRY(hipMemcpy(c->d_BT.p, BT.data(), sizeof(double) * BT.size(), hipMemcpyHostToDevice));
    // the same table padded to whole groups of 16 functions: layout of the fused-control kernels
    const int ld = nB <= 16 ? 16 : 32;
    std::vector<double> BT16((size_t)nT * ld, 0.0);
    for (int j = 0; j < nT; ++j)
      for (int i = 0; i < nB; ++i) BT16[(size_t)j * ld + i] = c->B[i + (size_t)nB * j];
    OCS_TRY(c->d_BT16.ensure(sizeof(double) * BT16.size()));
    HIP_TRY(hipMemcpy(c->d_BT16.p, BT16.data(), sizeof(double) * BT16.size(), hipMemcpyHostToDevice));
  }
  {  // column table of a banded basis: {w0, w1, adv, pad} with B(:,j) = w0 e_r + w1 e_{r+1}, r_j = r_{j-1} + adv
    const int R = fused_banded_rec();
    std::vector<double> ct((size_t)nT * R, 0.0);
    bool ok = nB >= 1;
    int rprev = 0;
    for (int j = 0; ok && j < nT; ++j) {
      int first = -1, last = -1, cnt = 0;
      for (int i = 0; i < nB; ++i)
        if (c->B[i + (size_t)nB * j] != 0.0) {
          if (first < 0) first = i;
          last = i;
          ++cnt;
        }
      if (cnt == 0 || cnt > 2 || last - first + 1 != cnt) {
        ok = false;
        break;
      }
      int r;
      if (cnt == 2) {
        r = first;
      } else if (j == 0) {
        r = first;
      } else if (first == rprev || first == rprev + 1) {
        r = rprev;          // the single entry sits inside the band of the column before
      } else {
        r = first - 1;      // ... or on the upper row of a band one row up
      }
      if (j > 0 && (r - rprev < 0 || r - rprev > 1)) {
        ok = false;
        break;
      }
      double* e = &ct[(size_t)j * R];
      e[0] = (r >= 0 && r < nB) ? c->B[r + (size_t)nB * j] : 0.0;
      e[1] = (r + 1 < nB) ? c->B[(r + 1) + (size_t)nB * j] : 0.0;
      e[2] = j == 0 ? 0.0 : (double)(r - rprev);
      e[3] = 0.0;
      if (j == 0) c->band_r0 = r;
      rprev = r;
    }
    c->banded = ok && !c->dense;
    if (c->banded) {
      ct[(size_t)(nT - 1) * R + 3] = (double)rprev;  // first row of the last column (start of the adjoint's cursors)
      OCS_TRY(c->d_CT.ensure(sizeof(double) * ct.size()));
      HIP_TRY(hipMemcpy(c->d_CT.p, ct.data(), sizeof(double) * ct.size(), hipMemcpyHostToDevice));
    }
  }
  if (!c->stream) HIP_TRY(hipStreamCreate(&c->stream));
  c->uploaded = true;
  return OCS_OK;
}

// host staging on the control's own stream
static int cstage_in(ocs_control_s* c, const double* host, DevBuf& dst, int per, int batch) {
  const size_t bytes = sizeof(double) * (size_t)per * batch;
  HIP_TRY(hipStreamSynchronize(c->stream));
  OCS_TRY(c->d_stage.ensure(bytes));
  OCS_TRY(dst.ensure(bytes));
  HIP_TRY(hipMemcpyAsync(c->d_stage.p, host, bytes, hipMemcpyHostToDevice, c->stream));
  LAUNCH_TRY(launch_to_batch_minor(c->d_stage.d(), dst.d(), per, batch, c->stream));
  return OCS_OK;
}
static int cstage_out(ocs_control_s* c, const double* src, double* host, int per, int batch) {
  const size_t bytes = sizeof(double) * (size_t)per * batch;
  OCS_TRY(c->d_stage.ensure(bytes));
  LAUNCH_TRY(launch_to_traj_major(src, c->d_stage.d(), per, batch, c->stream));
  HIP_TRY(hipMemcpyAsync(host, c->d_stage.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return OCS_OK;
}

// 'nearest' / 'next' of griddedInterpolant: the index of the sample a query takes (-1: NaN).  Shared with ocs_interp_dev.
namespace ocs {
int interp_sample_index(int method, int n, const double* x, double q) {
  if (q != q) return -1;
  if (method == OCS_INTERP_NEXT) {
    if (q > x[n - 1]) return -1;
    if (q <= x[0]) return 0;
    const int k = interval_of(n, x, q);   // x[k] <= q < x[k+1] inside, k = n - 2 at q == x[n-1]
    return q == x[k] ? k : k + 1;
  }
  const int k = interval_of(n, x, q);
  return (q - x[k] < x[k + 1] - q) ? k : k + 1;   // halfway: the later sample (interp1's rule); outside: the end sample
}
}  // namespace ocs

extern "C" {

// obj = PWLinearControl(t, nControlPts, nControls)        Control/PWLinearControl.m:13-18
// obj = PWConstantControl(t, nControlIntervals, nControls) Control/PWConstantControl.m:11-17
// obj = ChebyshevControl(t, nControlBasis, nControls)      Control/ChebyshevControl.m:13-18
int ocs_control_create(ocs_control* out, int kind, const double* t, int nt, int nBasis, int nControls) {
  if (!out || !t) return fail(OCS_ERR_INVALID, "null argument");
  *out = nullptr;
  if (nt < 2 || nBasis < 1 || nControls < 1) return fail(OCS_ERR_SHAPE, "need nt >= 2, nBasis >= 1, nControls >= 1");
  if (kind == OCS_CONTROL_PWLINEAR && nBasis < 2) return fail(OCS_ERR_SHAPE, "PWLinearControl needs >= 2 control points");
  if (kind < OCS_CONTROL_PWLINEAR || kind > OCS_CONTROL_CHEBYSHEV) return fail(OCS_ERR_UNSUPPORTED, "unknown control kind %d", kind);
  ocs_control_s* c = new ocs_control_s();
  c->kind = kind;
  c->nBasis = nBasis;
  c->nC = nControls;
  c->nT = nt;
  c->t.assign(t, t + nt);
  c->t0 = t[0];
  c->t1 = t[nt - 1];
  c->B.assign((size_t)nBasis * nt, 0.0);
  auto Bm = [&](int i, int j) -> double& { return c->B[i + (size_t)nBasis * j]; };
  if (kind == OCS_CONTROL_PWLINEAR) {
    c->pts.resize(nBasis);
    matlab_linspace(t[0], t[nt - 1], nBasis, c->pts.data());  // :16
    const double v10[2] = {1, 0}, v010[3] = {0, 1, 0}, v01[2] = {0, 1};
    for (int j = 0; j < nt; ++j) Bm(0, j) = tent_eval(2, c->pts.data(), v10, t[j]);  // :35-37
    for (int i = 1; i < nBasis - 1; ++i)                                              // :40-44
      for (int j = 0; j < nt; ++j) Bm(i, j) = tent_eval(3, c->pts.data() + i - 1, v010, t[j]);
    for (int j = 0; j < nt; ++j) Bm(nBasis - 1, j) = tent_eval(2, c->pts.data() + nBasis - 2, v01, t[j]);  // :47-49
  } else if (kind == OCS_CONTROL_PWCONSTANT) {
    std::vector<double> ls(nBasis + 1);
    matlab_linspace(t[0], t[nt - 1], nBasis + 1, ls.data());  // :14
    c->pts.assign(ls.begin(), ls.end() - 1);                  // :15
    for (int i = 0; i < nBasis - 1; ++i)                      // :43-47
      for (int j = 0; j < nt; ++j) Bm(i, j) = (t[j] >= c->pts[i] && t[j] < c->pts[i + 1]) ? 1.0 : 0.0;
    for (int j = 0; j < nt; ++j) Bm(nBasis - 1, j) = (t[j] >= c->pts[nBasis - 1]) ? 1.0 : 0.0;  // :49
  } else {
    c->pts.resize(nBasis);
    matlab_linspace(t[0], t[nt - 1], nBasis, c->pts.data());  // :16 (unused by the reference)
    for (int j = 0; j < nt; ++j) {
      const double tT = 2 * (t[j] - t[0]) / (t[nt - 1] - t[0]) - 1;  // :23
      Bm(0, j) = 1.0;
      if (nBasis > 1) Bm(1, j) = tT;
      for (int i = 2; i < nBasis; ++i) Bm(i, j) = 2 * tT * Bm(i - 1, j) - Bm(i - 2, j);  // :28-30
    }
  }
  *out = c;
  return OCS_OK;
}

int ocs_control_destroy(ocs_control c) {
  if (!c) return OCS_OK;
  if (c->stream) (void)hipStreamDestroy(c->stream);
  DevBuf* bufs[] = {&c->d_colptr, &c->d_row, &c->d_cval, &c->d_rowptr, &c->d_col, &c->d_rval, &c->d_BT, &c->d_BT16, &c->d_CT, &c->d_v,
                    &c->d_u, &c->d_dJdu, &c->d_dJdv, &c->d_stage, &c->d_x0, &c->d_J, &c->d_idx};
  for (DevBuf* b : bufs) b->release();
  delete c;
  return OCS_OK;
}
int ocs_control_dims(ocs_control c, int* nBasis, int* nControls, int* nt) {
  if (!c) return fail(OCS_ERR_INVALID, "null control");
  if (nBasis) *nBasis = c->nBasis;
  if (nControls) *nControls = c->nC;
  if (nt) *nt = c->nT;
  return OCS_OK;
}
int ocs_control_basis(ocs_control c, double* B) {
  if (!c || !B) return fail(OCS_ERR_INVALID, "null argument");
  memcpy(B, c->B.data(), sizeof(double) * c->B.size());
  return OCS_OK;
}
int ocs_control_points(ocs_control c, double* pts) {
  if (!c || !pts) return fail(OCS_ERR_INVALID, "null argument");
  memcpy(pts, c->pts.data(), sizeof(double) * c->pts.size());
  return OCS_OK;
}

// u = compute_u(obj, v)          PWLinearControl.m:59-62 (and twins)
int ocs_control_compute_u_dev(ocs_control c, int batch, const double* v, double* u, void* stream) {
  OCS_TRACE("ocs_control_compute_u_dev");
  if (!c || !v || !u || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(upload_control(c));
  if (c->dense) {  // (time-parallel since round 4: at every batch)
    LAUNCH_TRY(launch_basis_dense(true, c->nBasis, c->nT, c->nC, batch, c->d_BT.d(), v, u, (hipStream_t)stream));
    return OCS_OK;
  }
  LAUNCH_TRY(launch_basis_expand(c->nT, c->nC, batch, (const int*)c->d_colptr.p, (const int*)c->d_row.p,
                                 c->d_cval.d(), v, u, (hipStream_t)stream));
  return OCS_OK;
}
// dJdv = compute_dJdv(obj, dJdu) PWLinearControl.m:53-56 (and twins)
int ocs_control_compute_dJdv_dev(ocs_control c, int batch, const double* dJdu, double* dJdv, void* stream) {
  if (!c || !dJdu || !dJdv || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(upload_control(c));
  if (c->dense) {
    LAUNCH_TRY(launch_basis_dense(false, c->nBasis, c->nT, c->nC, batch, c->d_BT.d(), dJdu, dJdv, (hipStream_t)stream));
    return OCS_OK;
  }
  LAUNCH_TRY(launch_basis_contract(c->nBasis, c->nC, batch, (const int*)c->d_rowptr.p, (const int*)c->d_col.p,
                                   c->d_rval.d(), dJdu, dJdv, (hipStream_t)stream));
  return OCS_OK;
}
int ocs_control_compute_u(ocs_control c, int batch, const double* v, double* u) {
  if (!c || !v || !u || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(upload_control(c));
  const int nV = c->nC * c->nBasis, nU = c->nC * c->nT;
  OCS_TRY(cstage_in(c, v, c->d_v, nV, batch));
  OCS_TRY(c->d_u.ensure(sizeof(double) * (size_t)nU * batch));
  OCS_TRY(ocs_control_compute_u_dev(c, batch, c->d_v.d(), c->d_u.d(), c->stream));
  OCS_TRY(cstage_out(c, c->d_u.d(), u, nU, batch));
  return OCS_OK;
}
int ocs_control_compute_dJdv(ocs_control c, int batch, const double* dJdu, double* dJdv) {
  if (!c || !dJdu || !dJdv || batch < 1) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(upload_control(c));
  const int nV = c->nC * c->nBasis, nU = c->nC * c->nT;
  OCS_TRY(cstage_in(c, dJdu, c->d_dJdu, nU, batch));
  OCS_TRY(c->d_dJdv.ensure(sizeof(double) * (size_t)nV * batch));
  OCS_TRY(ocs_control_compute_dJdv_dev(c, batch, c->d_dJdu.d(), c->d_dJdv.d(), c->stream));
  OCS_TRY(cstage_out(c, c->d_dJdv.d(), dJdv, nV, batch));
  return OCS_OK;
}

// v = compute_initial_v(obj, u0): PWLinearControl.m:65-71, PWConstantControl.m:53-55,
// ChebyshevControl.m:46-48.  Build stance: an nC x 1 u0 is repmat'ed for the
// piecewise bases; any other length is an error instead of MATLAB's undefined `v`.
int ocs_control_compute_initial_v(ocs_control c, const double* u0, int len_u0, double* v) {
  if (!c || !u0 || !v) return fail(OCS_ERR_INVALID, "null argument");
  const int nC = c->nC, nB = c->nBasis;
  if (c->kind == OCS_CONTROL_CHEBYSHEV) {
    if (len_u0 != nC) return fail(OCS_ERR_SHAPE, "u0 must have nControls entries");
    for (int i = 0; i < nC * nB; ++i) v[i] = 0.0;
    for (int r = 0; r < nC; ++r) v[r] = u0[r];
    return OCS_OK;
  }
  if (len_u0 == nC) {
    for (int i = 0; i < nB; ++i)
      for (int r = 0; r < nC; ++r) v[r + (size_t)nC * i] = u0[r];
    return OCS_OK;
  }
  if (c->kind == OCS_CONTROL_PWLINEAR && len_u0 == nC * nB) {
    memcpy(v, u0, sizeof(double) * len_u0);
    return OCS_OK;
  }
  return fail(OCS_ERR_SHAPE, "compute_initial_v: u0 has %d entries (expected %d or %d)", len_u0, nC, nC * nB);
}
// [Lb, Ub] = compute_nlp_bounds(obj, controlBounds)   PWLinearControl.m:21-28, PWConstantControl.m:20-27
int ocs_control_compute_nlp_bounds(ocs_control c, const double* bounds, double* Lb, double* Ub) {
  if (!c || !bounds || !Lb || !Ub) return fail(OCS_ERR_INVALID, "null argument");
  if (c->kind == OCS_CONTROL_CHEBYSHEV)
    return fail(OCS_ERR_UNSUPPORTED, "ChebyshevControl has no compute_nlp_bounds (ChebyshevControl.m:51-53 is empty)");
  for (int i = 0; i < c->nBasis; ++i)
    for (int r = 0; r < c->nC; ++r) {
      Lb[r + (size_t)c->nC * i] = bounds[r] * 1.0;
      Ub[r + (size_t)c->nC * i] = bounds[c->nC + r] * 1.0;
    }
  return OCS_OK;
}

// vectorInterpolant(x, v, method)(tq)   functions/vectorInterpolant.m:1-12: samples of the
// griddedInterpolant the MATLAB shim wraps.  v is nComp x n, out nComp x nq.  Host-side glue.
int ocs_interp(int method, int nComp, int n, const double* x, const double* v, int nq, const double* tq, double* out) {
  if (!x || !v || !tq || !out || nComp < 1 || n < 2 || nq < 0) return fail(OCS_ERR_INVALID, "bad argument");
  if (method != OCS_INTERP_LINEAR && method != OCS_INTERP_PREVIOUS && method != OCS_INTERP_PCHIP &&
      method != OCS_INTERP_NEAREST && method != OCS_INTERP_NEXT)
    return fail(OCS_ERR_UNSUPPORTED, "unknown interpolation method %d", method);
  std::vector<double> row(n), d(n);
  for (int cidx = 0; cidx < nComp; ++cidx) {
    for (int i = 0; i < n; ++i) row[i] = v[cidx + (size_t)i * nComp];
    if (method == OCS_INTERP_PCHIP) pchip_slopes(n, x, row.data(), d.data());
    for (int j = 0; j < nq; ++j) {
      const double q = tq[j];
      double val;
      if (method == OCS_INTERP_PREVIOUS) {
        if (q < x[0])
          val = NAN;
        else
          val = row[q >= x[n - 1] ? n - 1 : interval_of(n, x, q)];
      } else if (method == OCS_INTERP_NEAREST || method == OCS_INTERP_NEXT) {
        const int idx = ocs::interp_sample_index(method, n, x, q);
        val = idx < 0 ? NAN : row[idx];
      } else {
        const int k = interval_of(n, x, q);
        if (method == OCS_INTERP_LINEAR) {
          val = row[k] + (row[k + 1] - row[k]) * ((q - x[k]) / (x[k + 1] - x[k]));
        } else {
          const double h = x[k + 1] - x[k], del = (row[k + 1] - row[k]) / h;
          const double dzzdx = (del - d[k]) / h, dzdxdx = (d[k + 1] - del) / h;
          const double c3 = (dzdxdx - dzzdx) / h, c2 = 2 * dzzdx - dzdxdx, s = q - x[k];
          val = row[k] + s * (d[k] + s * (c2 + s * c3));
        }
      }
      out[cidx + (size_t)j * nComp] = val;
    }
  }
  return OCS_OK;
}

// uFunc = compute_uFunc(obj, v); out = uFunc(tq).  PWLinearControl.m:74-77 ('linear' on controlPts),
// PWConstantControl.m:58-61 ('previous' on intervalStarts).  ChebyshevControl defines none: the
// build evaluates sum_k v_k T_k(tau) with the recurrence of ChebyshevControl.m:21-31.
int ocs_control_eval_uFunc(ocs_control c, const double* v, int nq, const double* tq, double* out) {
  if (!c || !v || !tq || !out) return fail(OCS_ERR_INVALID, "null argument");
  const int nC = c->nC, nB = c->nBasis;
  if (c->kind == OCS_CONTROL_PWLINEAR) return ocs_interp(OCS_INTERP_LINEAR, nC, nB, c->pts.data(), v, nq, tq, out);
  if (c->kind == OCS_CONTROL_PWCONSTANT) {
    if (nB == 1) {
      for (int j = 0; j < nq; ++j)
        for (int r = 0; r < nC; ++r) out[r + (size_t)nC * j] = tq[j] < c->pts[0] ? NAN : v[r];
      return OCS_OK;
    }
    return ocs_interp(OCS_INTERP_PREVIOUS, nC, nB, c->pts.data(), v, nq, tq, out);
  }
  for (int j = 0; j < nq; ++j) {
    const double tT = 2 * (tq[j] - c->t0) / (c->t1 - c->t0) - 1;
    for (int r = 0; r < nC; ++r) {
      double b0 = 1.0, b1 = tT, a = v[r] * 1.0;
      if (nB > 1) a += v[r + (size_t)nC] * tT;
      for (int i = 2; i < nB; ++i) {
        const double b2 = 2 * tT * b1 - b0;
        a += v[r + (size_t)nC * i] * b2;
        b0 = b1;
        b1 = b2;
      }
      out[r + (size_t)nC * j] = a;
    }
  }
  return OCS_OK;
}

// [J, dJdv] = nlpObjective(v)   functions/single_shooting.m:137-150
// device: x0 [nS][B] (overwritten at FreeInitStates), v [nV+nFree][B], J [B], dJdv [nV+nFree][B];
// FreeInitStates is a host array of 1-based state indices like MATLAB's.
int ocs_nlp_objective_dev(ocs_integrator g, ocs_problem p, ocs_control c, int batch, double* x0, const double* v,
                          int nFree, const int* FreeInitStates, double* J, double* dJdv, void* stream) {
  OCS_TRACE("ocs_nlp_objective_dev");
  if (!g || !p || !c || !x0 || !v || !J || !dJdv || batch < 1 || nFree < 0 || (nFree > 0 && !FreeInitStates))
    return fail(OCS_ERR_INVALID, "bad argument");
  if (c->nC != p->nC) return fail(OCS_ERR_SHAPE, "control has nC=%d, problem has nC=%d", c->nC, p->nC);
  if (c->nT != 2 * g->N + 1) return fail(OCS_ERR_SHAPE, "control was built on %d grid points, integrator has %d",
                                         c->nT, 2 * g->N + 1);
  hipStream_t s = (hipStream_t)stream;
  OCS_TRY(upload_control(c));
  const int nV = c->nC * c->nBasis, nU = c->nC * c->nT, nAug = p->nS + 1;
  const size_t B = (size_t)batch;
  // Dense basis with few functions on a plain RK4Integrator: u and dJdu never touch memory, the basis is
  // applied inside the RK4 kernels (ocs_fused_control_kernels.hip).  Measured on TestOCProblem + Chebyshev-16,
  // N = 1000 (ms per evaluation, unfused / fused, round 2): batch 64 0.80 / 0.43, 4096 0.90 / 0.44, 65536 2.54 / 0.61,
  // 262144 6.28 / 2.02.
  const bool fusable = c->dense && g->kind == 0 && fused_control_supported(p->functor, p->nS, p->nC, c->nBasis);
  // ... and where the shapes allow it (one state row, whole blocks and tiles) on the wave-specialised state pass and the
  // adjoint scan with the basis products on the matrix cores (ocs_fused_wave_kernels.hip)
  const bool wave_ok = fusable && c->fuse_mode != 1 && c->fuse_mode != 3 && (c->fuse_mode == 2 || batch <= 32768) &&
                       fused_wave_supported(p->functor, p->nS, p->nC, c->nBasis, g->N, batch);
  // Otherwise the fused LANE kernels against the unfused sequence (basis kernels + the automatic pass pair), re-measured in round 4
  // with the time-parallel dense basis kernels (Chebyshev-16, N = 1000, us per evaluation, unfused / fused lane):
  //   nS = 2: 8192 281 / 376, 16384 507 / 387;  nS = 3: 16384 712 / 752, 32768 1066 / 881;  nS = 4: 8192 411 / 868,
  //   16384 750 / 895, 32768 1260 / 1009, 65536 1946 / 1198   (profiles/r04_fusion_by_batch.log)
  const bool fused = fusable && c->fuse_mode != 1 &&
                     (c->fuse_mode >= 2 || wave_ok || batch >= (p->nS <= 2 ? 16384 : 32768));
  const bool fusedw = fused && wave_ok;
  // Banded basis (PWLinear, PWConstant): the same with two live coefficient rows per trajectory
  // (ocs_fused_banded_kernels.hip)
  // Measured (TestOCProblem, N = 500, PWLinear 101 points; ms per evaluation unfused / fused): batch 4096 0.16 / 0.31,
  // 65536 0.86 / 0.46 -- at small batch the unfused path has the wave-specialised forward kernel and time-parallel
  // sparse basis kernels, at large batch the traffic of u and dJdu decides.
  const bool fusedb = !fused && c->banded && g->kind == 0 && fused_banded_supported(p->functor, p->nS, p->nC) &&
                      c->fuse_mode != 1 && (c->fuse_mode == 2 || batch >= 32768);
  if (!fused && !fusedb) {
    OCS_TRY(c->d_u.ensure(sizeof(double) * (size_t)nU * B));
    OCS_TRY(c->d_dJdu.ensure(sizeof(double) * (size_t)nU * B));
    OCS_TRY(ocs_control_compute_u_dev(c, batch, v, c->d_u.d(), stream));                     // :139 / :145
  }
  if (nFree > 0) {
    std::vector<int> idx(nFree);
    for (int f = 0; f < nFree; ++f) {
      idx[f] = FreeInitStates[f] - 1;
      if (idx[f] < 0 || idx[f] >= p->nS) return fail(OCS_ERR_INVALID, "FreeInitStates entry %d out of range", FreeInitStates[f]);
    }
    // one buffer: [int idx[nFree] | pad to 8 | double lam0[nAug][B]]
    OCS_TRY(c->d_idx.ensure(((sizeof(int) * nFree + 7) / 8) * 8 + sizeof(double) * (size_t)nAug * B));
    HIP_TRY(hipMemcpyAsync(c->d_idx.p, idx.data(), sizeof(int) * nFree, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // idx is a stack vector
    LAUNCH_TRY(launch_scatter_rows(nFree, batch, (const int*)c->d_idx.p, v + (size_t)nV * B, x0, s));  // :146
  }
  double* lam0 = nullptr;
  if (nFree > 0)
    lam0 = reinterpret_cast<double*>(static_cast<char*>(c->d_idx.p) + ((sizeof(int) * nFree + 7) / 8) * 8);
  if (fusedb) {
    OCS_TRY(bind_problem(g, p, batch, s));
    OCS_TRY(g->d_ck.ensure(sizeof(double) * (size_t)nAug * (g->N + 1) * B));
    g->ck = nullptr;
    HIP_TRY(hipMemsetAsync(dJdv, 0, sizeof(double) * (size_t)nV * B, s));  // rows outside every band
    LAUNCH_TRY(launch_forward_fb(describe(p), describe(g), batch, c->nBasis, c->band_r0, c->d_CT.d(), v, x0, g->d_ck.d(),
                                 J, s));
    LAUNCH_TRY(launch_backward_fb(describe(p), describe(g), batch, c->nBasis, c->band_r0, c->d_CT.d(), v, g->d_ck.d(),
                                  dJdv, lam0, s));
    if (nFree > 0)
      LAUNCH_TRY(launch_gather_rows(nFree, batch, (const int*)c->d_idx.p, lam0, dJdv + (size_t)nV * B, s));
    return OCS_OK;
  }
  if (fused) {
    OCS_TRY(bind_problem(g, p, batch, s));
    OCS_TRY(g->d_ck.ensure(sizeof(double) * (size_t)nAug * (g->N + 1) * B));
    g->ck = nullptr;  // these checkpoints belong to no u in memory: a later compute_adjoints must not use them
    if (fusedw && describe(g).RECS) {
      const int ldbt = c->nBasis <= 16 ? 16 : 32;
      LAUNCH_TRY(launch_forward_fcw(describe(p), describe(g), batch, c->nBasis, ldbt, c->d_BT16.d(), v, x0, g->d_ck.d(), J, s));
      LAUNCH_TRY(launch_backward_fcs(describe(p), describe(g), batch, c->nBasis, ldbt, c->d_BT16.d(), v, g->d_ck.d(), dJdv,
                                     lam0, s));
      if (nFree > 0)
        LAUNCH_TRY(launch_gather_rows(nFree, batch, (const int*)c->d_idx.p, lam0, dJdv + (size_t)nV * B, s));
      return OCS_OK;
    }
    LAUNCH_TRY(launch_forward_fc(describe(p), describe(g), batch, c->nBasis, c->d_BT16.d(), v, x0, g->d_ck.d(), J, s));
    LAUNCH_TRY(launch_backward_fc(describe(p), describe(g), batch, c->nBasis, c->d_BT16.d(), v, g->d_ck.d(), dJdv,
                                  lam0, s));
    if (nFree > 0)
      LAUNCH_TRY(launch_gather_rows(nFree, batch, (const int*)c->d_idx.p, lam0, dJdv + (size_t)nV * B, s));
    return OCS_OK;
  }
  OCS_TRY(ocs_compute_states_dev(g, p, batch, x0, c->d_u.d(), nullptr, J, stream));         // :140 / :147
  g->want_lam0 = lam0;
  const int rc = ocs_compute_adjoints_dev(g, p, batch, c->d_u.d(), nullptr, nullptr, c->d_dJdu.d(), stream);  // :141 / :148
  g->want_lam0 = nullptr;
  if (rc < 0) return rc;
  OCS_TRY(ocs_control_compute_dJdv_dev(c, batch, c->d_dJdu.d(), dJdv, stream));              // :142 / :149
  if (nFree > 0)                                                                              // lam(FreeInitStates,1)
    LAUNCH_TRY(launch_gather_rows(nFree, batch, (const int*)c->d_idx.p, lam0, dJdv + (size_t)nV * B, s));
  return OCS_OK;
}

}  // extern "C"
// the objectives of the last host ocs_nlp_objective on this handle, on its device (ocs_multi.cpp's reductions)
const double* ocs_control_device_J(const ocs_control_s* c) { return c ? c->d_J.d() : nullptr; }
int ocs_control_device_id(const ocs_control_s* c) { return c ? c->device : -1; }
extern "C" {

int ocs_control_set_fusion(ocs_control c, int mode) {
  if (!c) return fail(OCS_ERR_INVALID, "null control");
  if (mode < 0 || mode > 3)
    return fail(OCS_ERR_INVALID, "fusion mode must be 0 (automatic), 1 (off), 2 (on) or 3 (on, lane kernels only)");
  c->fuse_mode = mode;
  return OCS_OK;
}

// host: x0 nS x batch (in/out), v (nV+nFree) x batch, J batch, dJdv (nV+nFree) x batch
int ocs_nlp_objective(ocs_integrator g, ocs_problem p, ocs_control c, int batch, double* x0, const double* v,
                      int nFree, const int* FreeInitStates, double* J, double* dJdv) {
  OCS_TRACE("ocs_nlp_objective");
  if (!g || !p || !c || !x0 || !v || !J || !dJdv || batch < 1 || nFree < 0) return fail(OCS_ERR_INVALID, "bad argument");
  OCS_TRY(upload_control(c));
  const int nV = c->nC * c->nBasis + nFree;
  if (nFree > 0) {  // reserve the index + lam0 scratch before any pointer into it is formed
    OCS_TRY(c->d_idx.ensure(((sizeof(int) * nFree + 7) / 8) * 8 + sizeof(double) * (size_t)(p->nS + 1) * batch));
  }
  OCS_TRY(cstage_in(c, v, c->d_v, nV, batch));
  OCS_TRY(cstage_in(c, x0, c->d_x0, p->nS, batch));
  OCS_TRY(c->d_J.ensure(sizeof(double) * batch));
  OCS_TRY(c->d_dJdv.ensure(sizeof(double) * (size_t)nV * batch));
  OCS_TRY(ocs_nlp_objective_dev(g, p, c, batch, c->d_x0.d(), c->d_v.d(), nFree, FreeInitStates, c->d_J.d(),
                                c->d_dJdv.d(), c->stream));
  HIP_TRY(hipMemcpyAsync(J, c->d_J.p, sizeof(double) * batch, hipMemcpyDeviceToHost, c->stream));
  OCS_TRY(cstage_out(c, c->d_dJdv.d(), dJdv, nV, batch));
  if (nFree > 0) OCS_TRY(cstage_out(c, c->d_x0.d(), x0, p->nS, batch));
  g->traj_status.assign(batch, 0);
  int rc = OCS_OK;
  for (int b = 0; b < batch; ++b)
    if (!std::isfinite(J[b])) {
      g->traj_status[b] = OCS_NUM_NONFINITE;
      rc = OCS_NUM_NONFINITE;
    }
  return rc;
}

}  // extern "C"
