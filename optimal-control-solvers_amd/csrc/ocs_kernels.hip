// ocs_kernels.hip -- gfx950 (MI355X) kernels for the batched RK4 state pass and its exact
// discrete adjoint.  fp64 throughout.
//
// Mapping ("L", lane-per-trajectory): one lane integrates one trajectory; a 64-lane wave
// (= one workgroup) owns 64 consecutive trajectories.  All per-step quantities that do not
// depend on the trajectory (step sizes, time coefficients) are wave-uniform and come from
// scalar loads.  Arrays are batch-minor ([time][row][batch]) so every vector load/store of a
// wave is one contiguous 512-B segment: coalescing needs no LDS staging in this mapping.
// The time recursion is serial, so at small batch the passes are bound by fp64 issue on one
// wave per SIMD, not by HBM; control samples and checkpoints are
// software-prefetched CH steps ahead in registers so that HBM latency stays off the chain.
//
// Reference semantics: Integrator/RK4Integrator.m:28-56 (compute_states), :59-94
// (compute_adjoints), :97-121 (compute_dJdu).  The reference caches all four stage states
// (xK); here only y_i is checkpointed and Y2..Y4 are recomputed in the adjoint pass (same
// arithmetic, deterministic => same values), which cuts checkpoint traffic 4x.
#include "ocs_device_common.hpp"
#include "ocs_internal.hpp"
#include "ocs_jit.hpp"
#include "ocs_problems.hpp"
#include "ocs_rk4_kernels.hpp"

namespace ocs {

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// ---------------------------------------------------------------------------------------
// layout helpers: 64x64 LDS-tiled transposes between [batch][per] and [per][batch]
// ---------------------------------------------------------------------------------------
// src is R x Cc row-major (R rows of Cc), dst is Cc x R row-major.
__global__ __launch_bounds__(256) void k_transpose(const double* __restrict__ src, double* __restrict__ dst,
                                                   int R, int Cc) {
  __shared__ double tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    if (r < R && c < Cc) tile[rr][tx] = src[(size_t)r * Cc + c];
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, r = r0 + tx;
    if (r < R && c < Cc) dst[(size_t)c * R + r] = tile[tx][cc];
  }
}

// straight 8-byte-per-lane copy: the known-byte-count reference for calibrating the HBM counters
__global__ __launch_bounds__(256) void k_copy8(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
int launch_copy8(const double* src, double* dst, size_t n, hipStream_t s) {
  k_copy8<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(src, dst, n);
  return hip_rc(hipGetLastError());
}

__global__ void k_count_nonfinite(const double* __restrict__ v, int n, int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool bad = (i < n) && !isfinite(v[i]);
  const unsigned long long m = __ballot(bad);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}

// ---------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------
#define OCS_DISPATCH_LOGISTIC(NSV, CALL) \
  switch (NSV) {                         \
    case 1: { using P = LogisticK<1>; CALL; } break; \
    case 2: { using P = LogisticK<2>; CALL; } break; \
    case 3: { using P = LogisticK<3>; CALL; } break; \
    case 4: { using P = LogisticK<4>; CALL; } break; \
    default: return -1;                  \
  }

bool functor_supported(Functor f, int nS, int nC) {
  if (f == Functor::User) return true;
  if (f == Functor::LQ) return lq_supported(nS, nC);
  if (f == Functor::Logistic) return nS >= 1 && nS <= 4 && nC == 1;
  return false;
}
int functor_ntc(Functor f, int nS) {
  (void)nS;
  if (f == Functor::User || f == Functor::LQ) return 1;
  if (f == Functor::Logistic) return LogisticK<1>::NTC;
  return 0;
}
int functor_ntu(Functor f, int nS) {
  (void)nS;
  if (f == Functor::User || f == Functor::LQ) return 1;
  if (f == Functor::Logistic) return LogisticK<1>::NTU;
  return 0;
}
unsigned functor_tc_param_mask(Functor f, int nS) {
  (void)nS;
  if (f == Functor::User) return 0u;
  if (f == Functor::LQ) return 1u;  // r feeds the e^{-rt} table
  if (f == Functor::Logistic) return LogisticK<1>::TC_PARAM_MASK;
  return 0;
}

template <class P>
static void run_tcoef(const ProblemDesc& p, const GridDesc& g, hipStream_t s) {
  const int nT = 2 * g.N + 1;
  k_tcoef<P><<<dim3((nT + 255) / 256), dim3(256), 0, s>>>(nT, g.T, p.ps, g.TC, g.TU);
  k_build_rec<P><<<dim3((g.N + 2 * kRecPad + 255) / 256), dim3(256), 0, s>>>(g.N, g.HT, g.TC, g.REC);
}
int launch_tcoef(const ProblemDesc& p, const GridDesc& g, hipStream_t s) {
  if (p.functor == Functor::User) {
    int nT = 2 * g.N + 1, N = g.N;
    const double *T = g.T, *ps = p.ps, *HT = g.HT;
    double *TC = g.TC, *TU = g.TU, *REC = g.REC;
    void* a1[] = {&nT, &T, &ps, &TC, &TU};
    int rc = jit_launch(p.user, UK_TCOEF, dim3((nT + 255) / 256), dim3(256), a1, s);
    if (rc) return rc;
    const double* TCc = g.TC;
    void* a2[] = {&N, &HT, &TCc, &REC};
    return jit_launch(p.user, UK_BUILD_REC, dim3((N + 2 * kRecPad + 255) / 256), dim3(256), a2, s);
  }
  if (p.functor == Functor::LQ) return launch_tcoef_lq(p, g, s);
  OCS_DISPATCH_LOGISTIC(p.nS, run_tcoef<P>(p, g, s));
  return hip_rc(hipGetLastError());
}
int rec_stride_host(int ntc) { return rec_stride(ntc); }
int rec_sc_offset_host(int ntc) { return rec_sc_offset(ntc); }
int rec_pad_host() { return kRecPad; }

constexpr int kChunk = 4;
// step records in flight: a divisor of 2*kChunk so the ring needs no register rotation at the loop
// back-edge; deeper for the small problems, whose register budget is loose and which run with few waves
template <class P>
constexpr int pf_of() { return P::NS <= 2 ? 4 : 3; }

// the lane adjoint kernel with checkpoint re-integration (k_backward<..., XRC>): where the launch fills the chip and the pass is
// HBM-bound (OCS_LANE_XRC_MIN: batch from which it is taken; 0 = never)
static int lane_xrc_min_batch() {
  static const int v = [] {
    const char* e = getenv("OCS_LANE_XRC_MIN");
    return e ? atoi(e) : 32768;
  }();
  return v;
}
template <class P>
static void run_forward(const FwdArgs& a, bool uconst, hipStream_t s) {
  const dim3 grid((a.batch + 63) / 64), block(64);
  if (uconst)
    k_forward<P, kChunk, pf_of<P>(), true, true><<<grid, block, 0, s>>>(a);
  else if (a.x && lane_xrc_min_batch() > 0 && a.batch >= lane_xrc_min_batch())
    k_forward<P, kChunk, pf_of<P>(), true, false, true><<<grid, block, 0, s>>>(a);   // (non-temporal stores of x)
  else if (a.x)
    k_forward<P, kChunk, pf_of<P>(), true, false><<<grid, block, 0, s>>>(a);
  else
    k_forward<P, kChunk, pf_of<P>(), false, false><<<grid, block, 0, s>>>(a);
}
// Mapping selection, measured on MI355X with the kernels of round 4 (LogisticK, N = 1000; us per pass; scripts/pair_loop.py,
// profiles/r04_pair_by_batch_mapping.log) by workgroups of the wave-specialised kernels (one per 64/nS trajectories):
//   workgroups      state pass: lane / pipeline            adjoint pass: lane / scan
//                   nS = 1      nS = 2      nS = 4          nS = 1        nS = 2       nS = 4
//      512         256 / 267   209 / 131   297 / 124       351 / 300     296 / 205    439 / 166
//     1024         434 / 545   285 / 267   342 / 241       647 / 630     429 / 416    466 / 370
//     2048         837 / 1026  499 / 553   452 / 500      1297 / 1191    791 / 825    639 / 737
//     4096        1762 / 2046 1004 / 1052  673 / 892      2736 / 2438   1598 / 1628  1172 / 1435
// The wave-specialised kernels win while the serial chain sets the time -- up to 1024 workgroups (512 with one state, where a
// lane kernel already has a wave per row) --; beyond that the lane kernels have the fewest instructions per trajectory and both
// passes are HBM-bound (only the one-state scan still matches its lane kernel).  Row-split: while the lane mapping would leave most
// SIMDs idle and the pipeline does not apply.
// The pipeline kernels take whole blocks of 8 steps.  Other step counts are split: the first 8*floor(N/8) steps
// go through the pipeline kernel, the remaining (< 8) through the lane kernel, which continues from / hands over
// the boundary column (running objective, lamT, the k1 half of the boundary column of dJdu), so the result is
// the one-kernel result bit for bit.  splits(): 0 = no pipeline, else the number of pipeline steps.
static int pipeline_steps(const ProblemDesc& p, int N, int batch, bool backward) {
  const int D = pipeline_block_steps(), N1 = (N / D) * D;
  // (user problems given as row functions have the state-pass kernel only: their adjoint pass is the scan)
  if (backward ? !pipeline_supported(p.functor, p.nS, p.nC) : !pipeline_problem_ok(p)) {
    // any other problem with nS <= 4, nC <= 2: the vector-lane state pass (whole tiles of 64 trajectories)
    if (!backward && vector_problem_ok(p) && N1 >= D && batch >= 64 && tile_ok(batch, 64)) return N1;
    return 0;
  }
  if (N1 < D || !pipeline_shape_ok(p.nS, N1, batch, backward)) return 0;
  return N1;
}
static bool forward_is_vector(const ProblemDesc& p) { return !pipeline_problem_ok(p) && vector_problem_ok(p); }
// boundary: the caller's output arrays can carry the hand-over column of a split pass (x forward, lam backward)
static int choose_mapping(const ProblemDesc& p, int N, int batch, int requested, bool plain, bool backward,
                          bool boundary) {
  if (requested != MAP_AUTO) return requested;
  const int N1 = plain ? pipeline_steps(p, N, batch, backward) : 0;
  const int tpw = (!backward && forward_is_vector(p)) ? 64 : 64 / p.nS;
  const int wgmax = (p.nS == 1 || tpw == 64 || backward) ? 512 : 1024;
  if (N1 > 0 && (N1 == N || boundary) && batch / tpw <= wgmax) return MAP_PIPELINE;
  if (plain && rowsplit_supported(p.functor, p.nS, p.nC) && batch <= 8192) return MAP_ROWSPLIT;
  return MAP_LANE;
}

// Scan against the serial mappings: the scan does ~2x the arithmetic of a serial adjoint step but has no
// dependent chain over time; it wins wherever the serial kernels are latency-bound (every batch measured so far).
// ... wherever the serial kernels are latency-bound: up to 1024 workgroups (table above); with one state at every batch.
// Beyond that the lane kernel re-integrates three of four checkpoints (k_backward<..., XRC>, from batch 32768) and wins with two
// and four states: adjoint pass at batch 65536, scan / lane / lane with re-integration: nS = 4 1435 / 1185 / 903 us, nS = 2
// 825 / 811 / 720; nS = 1 at 131072: 1191 / 1327 / 1167 (level: the scan stays).
static bool scan_pays(int nS, int N, int batch) {
  return N >= 8 && (nS == 1 || batch / (64 / nS) <= 1024);
}

// the wave-specialised state passes on whole horizons (the two-kernel sweep and the fused control update are built on them)
int fold_wg_limit() {
  static const int v = [] {
    const char* e = getenv("OCS_FOLD_MAX_WG");
    const int n = e ? atoi(e) : 0;
    return n > 0 ? n : (1 << 30);   // no limit: at 1024-4096 workgroups the folded sweep is 1.3-2.2x faster than the lane sequence
                                    // (profiles/r04_fbs_by_batch_fold_limit.log; until round 4 the limit was 512)
  }();
  return v;
}
bool forward_gate_supported(const ProblemDesc& p, const GridDesc& g, int batch) {
  if (p.functor == Functor::LQ) return false;
  const int tpw = forward_is_vector(p) ? 64 : 64 / p.nS;
  return pipeline_steps(p, g.N, batch, false) == g.N && g.N > 0 && batch / tpw <= fold_wg_limit();
}
bool tail_leg_wave_ok(const ProblemDesc& p, const GridDesc& g, int batch) {
  if (p.functor == Functor::LQ) return false;
  return choose_mapping(p, g.N, batch, MAP_AUTO, true, false, true) == MAP_PIPELINE;
}
// any state pass the sweep launches with `frozen` set (pipeline kernels, split passes, the lane kernel): FwdOpts::gate
bool forward_gate_any(const ProblemDesc& p) { return p.functor != Functor::LQ; }

int launch_forward(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                   double* x, double* J, const FwdOpts& o, hipStream_t s) {
  if (o.gate && (o.mapping != MAP_AUTO || o.uconst || o.Jadd || !forward_gate_any(p) || !o.frozen)) return -1;
  if (p.functor == Functor::LQ) return launch_forward_lq(p, g, batch, x0, u, x, J, o, s);
  const bool plain = !o.uconst && !o.Jadd;
  // (MAP_SCAN names the adjoint kernel; the state pass of such an integrator is chosen automatically)
  int map = choose_mapping(p, g.N, batch, o.mapping == MAP_SCAN ? MAP_AUTO : o.mapping, plain, false, x != nullptr);
  if (map == MAP_ROWSPLIT && (o.frozen || o.ld) && o.mapping == MAP_AUTO) map = MAP_LANE;  // not in that kernel
  if (map == MAP_PIPELINE && forward_is_vector(p) && o.ld && o.mapping == MAP_AUTO) map = MAP_LANE;
  if (map == MAP_PIPELINE) {
    const int N1 = plain ? pipeline_steps(p, g.N, batch, false) : 0;
    if (N1 == 0 || (N1 < g.N && !x)) return -1;  // the split needs the boundary column in memory
    GridDesc g1 = g;
    g1.N = N1;
    // (a split pass hands the running objective to the lane kernel through the boundary column: keep the row then)
    int rc;
    if (forward_is_vector(p)) {
      if (o.ld) return -1;
      rc = launch_forward_pv(p, g1, batch, x0, u, x, J, s, o.no_cost_row && N1 == g.N, o.gate, o.frozen);
    } else {
      rc = launch_forward_pl(p, g1, batch, x0, u, x, J, o.frozen, o.dump, o.ld, s, o.no_cost_row && N1 == g.N, o.gate);
    }
    if (rc || N1 == g.N) return rc;
    // remaining steps N1 .. N-1 on the lane kernel, continuing from column N1 (state rows and running objective)
    const size_t ldb = o.ld ? o.ld : batch;
    const size_t col = (size_t)(p.nS + 1) * ldb, ucol = (size_t)p.nC * ldb;
    double* xb = x + (size_t)N1 * col;
    FwdArgs a{g.N - N1, batch, g.REC + (size_t)N1 * rec_stride_host(functor_ntc(p.functor, p.nS)), p.ps, p.pb,
              p.pmask, xb, u + (size_t)(2 * N1) * ucol, xb, J, nullptr, o.frozen, o.dump, xb + (size_t)p.nS * ldb,
              o.ld};
    a.gate = o.gate;
    if (p.functor == Functor::User) {
      void* args[] = {(void*)&a};
      return jit_launch(p.user, UK_FWD_X, dim3((batch + 63) / 64), dim3(64), args, s);
    }
    OCS_DISPATCH_LOGISTIC(p.nS, run_forward<P>(a, false, s));
    return hip_rc(hipGetLastError());
  }
  if (map == MAP_ROWSPLIT) {
    if (!plain || o.frozen || o.ld || !rowsplit_supported(p.functor, p.nS, p.nC)) return -1;
    return launch_forward_rs(p, g, batch, x0, u, x, J, s);
  }
  if (o.uconst && !x) return -1;
  FwdArgs a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, x0, u, x, J, o.Jadd, o.frozen, o.dump, nullptr, o.ld};
  a.gate = o.gate;
  if (p.functor == Functor::User) {
    void* args[] = {(void*)&a};
    const int kid = o.uconst ? UK_FWD_UCONST : (x ? UK_FWD_X : UK_FWD_J);
    return jit_launch(p.user, kid, dim3((batch + 63) / 64), dim3(64), args, s);
  }
  OCS_DISPATCH_LOGISTIC(p.nS, run_forward<P>(a, o.uconst, s));
  return hip_rc(hipGetLastError());
}

template <class P>
static void run_backward(const BwdArgs& a, bool uconst, hipStream_t s) {
  const dim3 grid((a.batch + 63) / 64), block(64);
  const int xmin = lane_xrc_min_batch();
  if (!uconst && xmin > 0 && a.batch >= xmin && a.N >= kChunk) {
    if (a.lam && a.dJdu)
      k_backward<P, kChunk, kChunk, true, true, false, true><<<grid, block, 0, s>>>(a);
    else if (a.lam)
      k_backward<P, kChunk, kChunk, true, false, false, true><<<grid, block, 0, s>>>(a);
    else
      k_backward<P, kChunk, kChunk, false, true, false, true><<<grid, block, 0, s>>>(a);
    return;
  }
  if (uconst)
    k_backward<P, kChunk, pf_of<P>(), false, false, true><<<grid, block, 0, s>>>(a);
  else if (a.lam && a.dJdu)
    k_backward<P, kChunk, pf_of<P>(), true, true, false><<<grid, block, 0, s>>>(a);
  else if (a.lam)
    k_backward<P, kChunk, pf_of<P>(), true, false, false><<<grid, block, 0, s>>>(a);
  else
    k_backward<P, kChunk, pf_of<P>(), false, true, false><<<grid, block, 0, s>>>(a);
}
int launch_backward(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                    const double* lamT, double* lam, double* dJdu, const BwdOpts& o, hipStream_t s) {
  if (p.functor == Functor::LQ) return launch_backward_lq(p, g, batch, xck, u, lamT, lam, dJdu, o, s);
  const bool plain = !o.uconst;
  int map = choose_mapping(p, g.N, batch, o.mapping, plain, true, lam != nullptr);
  // the adjoint pass as a scan over time: whenever the problem is row-separable (any N, any batch); the serial
  // mappings remain selectable
  // The scan takes whole chunks of L steps; other step counts are split as for the pipeline mapping: steps N1..N-1
  // first on the lane kernel (which hands lam(:, N1) and the k1 half of column 2 N1 over through the output arrays),
  // so a split pass needs the lam array.
  const int Ls = scan_chunk_steps(), Ns = (g.N / Ls) * Ls;
  const bool scan_ok = plain && g.RECS && scan_problem_ok(p) && Ns >= Ls && (Ns == g.N || lam || o.split_scratch);
  if (o.mapping == MAP_AUTO && scan_ok && scan_pays(p.nS, g.N, batch)) map = MAP_SCAN;
  // any other problem with nS <= 4, nC <= 2: the scan with dense step maps, while the lane kernel would leave most of
  // the chip idle (it does (nS + 2) times the arithmetic of a serial step)
  const bool vscan = !scan_ok && plain && g.RECS && vector_problem_ok(p) && Ns >= Ls && (Ns == g.N || lam || o.split_scratch);
  // (measured again with round 4's kernels, us per adjoint pass, dense-map scan / lane kernel: nS = 1 at batch 16384 176 / 236,
  //  32768 365 / 365, 49152 556 / 510; nS = 2 at 16384 271 / 447, 24576 470 / 486, 32768 540 / 496; nS = 4 at 8192 381 / 422,
  //  16384 415 / 428, 24576 775 / 520 -- profiles/r04_user_vector_pair_by_batch_mapping.log)
  if (o.mapping == MAP_AUTO && vscan && g.N >= 8 && batch <= 16384 * 6 / (p.nS + 2)) map = MAP_SCAN;
  auto scan_launch = [&](const GridDesc& gg, const double* lT, const double* pend0) {
    return vscan ? launch_backward_vscan(p, gg, batch, xck, u, lT, lam, dJdu, o.lam0, pend0, s)
                 : launch_backward_scan(p, gg, batch, xck, u, lT, lam, dJdu, o.lam0, pend0, s);
  };
  if (map == MAP_SCAN) {
    if (!scan_ok && !vscan) return -1;
    if (Ns == g.N) return scan_launch(g, lamT, nullptr);
    const size_t col = (size_t)(p.nS + 1) * batch, ucol = (size_t)p.nC * batch;
    double* lamb = lam ? lam + (size_t)Ns * col : o.split_scratch;   // lam(:, Ns): in the output array, or as lam0
    double* db = dJdu ? dJdu + (size_t)(2 * Ns) * ucol : nullptr;
    const BwdArgs a{g.N - Ns, batch, g.REC + (size_t)Ns * rec_stride_host(functor_ntc(p.functor, p.nS)), p.ps, p.pb,
                    p.pmask, xck + (size_t)Ns * col, u + (size_t)(2 * Ns) * ucol, lamT, lam ? lamb : nullptr, db,
                    lam ? nullptr : lamb};
    int rc;
    if (p.functor == Functor::User) {
      void* args[] = {(void*)&a};
      rc = jit_launch(p.user, a.lam && a.dJdu ? UK_BWD_LAM_DJDU : (a.lam ? UK_BWD_LAM : UK_BWD_DJDU),
                      dim3((batch + 63) / 64), dim3(64), args, s);
    } else {
      OCS_DISPATCH_LOGISTIC(p.nS, run_backward<P>(a, false, s));
      rc = hip_rc(hipGetLastError());
    }
    if (rc) return rc;
    GridDesc g1 = g;
    g1.N = Ns;
    return scan_launch(g1, lamb, db);
  }
  if (map == MAP_PIPELINE) {
    const int N1 = plain ? pipeline_steps(p, g.N, batch, true) : 0;
    if (N1 == 0 || (N1 < g.N && !lam)) return -1;  // the split hands lam(:, N1) over through memory
    if (N1 == g.N) return launch_backward_pl(p, g, batch, xck, u, lamT, lam, dJdu, o.lam0, nullptr, s);
    // steps N-1 .. N1 first, on the lane kernel: lam columns N1..N, dJdu columns 2 N1 .. 2 N (column 2 N1 holds the
    // k1 half only, RK4Integrator.m:108-112) ...
    const size_t col = (size_t)(p.nS + 1) * batch, ucol = (size_t)p.nC * batch;
    double* lamb = lam + (size_t)N1 * col;
    double* db = dJdu ? dJdu + (size_t)(2 * N1) * ucol : nullptr;
    const BwdArgs a{g.N - N1, batch, g.REC + (size_t)N1 * rec_stride_host(functor_ntc(p.functor, p.nS)), p.ps, p.pb,
                    p.pmask, xck + (size_t)N1 * col, u + (size_t)(2 * N1) * ucol, lamT, lamb, db, nullptr};
    OCS_DISPATCH_LOGISTIC(p.nS, run_backward<P>(a, false, s));
    int rc = hip_rc(hipGetLastError());
    if (rc) return rc;
    // ... then the whole blocks below, started from lam(:, N1) and adding their k4 half to column 2 N1
    GridDesc g1 = g;
    g1.N = N1;
    return launch_backward_pl(p, g1, batch, xck, u, lamb, lam, dJdu, o.lam0, db, s);
  }
  if (map == MAP_ROWSPLIT) {
    if (!plain || !rowsplit_supported(p.functor, p.nS, p.nC)) return -1;
    return launch_backward_rs(p, g, batch, xck, u, lamT, lam, dJdu, o.lam0, s);
  }
  if (o.uconst ? (lam || dJdu || !o.lam0) : (!lam && !dJdu)) return -1;
  const BwdArgs a{g.N, batch, g.REC, p.ps, p.pb, p.pmask, xck, u, lamT, lam, dJdu, o.lam0};
  if (p.functor == Functor::User) {
    void* args[] = {(void*)&a};
    int kid = o.uconst ? UK_BWD_UCONST : (lam && dJdu ? UK_BWD_LAM_DJDU : (lam ? UK_BWD_LAM : UK_BWD_DJDU));
    const int xmin = lane_xrc_min_batch();   // (as for the registry problems: re-integration where the launch is HBM-bound)
    if (kid == UK_BWD_LAM_DJDU && xmin > 0 && batch >= xmin && g.N >= 4 && p.nS <= 4) kid = UK_BWD_LAM_DJDU_XRC;
    return jit_launch(p.user, kid, dim3((batch + 63) / 64), dim3(64), args, s);
  }
  OCS_DISPATCH_LOGISTIC(p.nS, run_backward<P>(a, o.uconst, s));
  return hip_rc(hipGetLastError());
}

template <class P>
static void run_eval(const ProblemDesc& p, int which, int k, const double* t, const double* y, const double* u,
                     const double* v, double* out, hipStream_t s) {
  k_eval<P><<<dim3((k + 127) / 128), dim3(128), 0, s>>>(which, k, t, y, u, v, p.ps, out, p.lb, p.ub);
}
int launch_eval(const ProblemDesc& p, int which, int k, const double* t, const double* y, const double* u,
                const double* v, double* out, hipStream_t s) {
  if (p.functor == Functor::User) {
    const double *ps = p.ps, *lb = p.lb, *ub = p.ub;
    void* args[] = {&which, &k, &t, &y, &u, &v, &ps, &out, &lb, &ub};
    return jit_launch(p.user, UK_EVAL, dim3((k + 127) / 128), dim3(128), args, s);
  }
  if (p.functor == Functor::LQ) return which == 3 ? -1 : launch_eval_lq(p, which, k, t, y, u, v, out, s);   // (ControlChar: its plugin twin)
  OCS_DISPATCH_LOGISTIC(p.nS, run_eval<P>(p, which, k, t, y, u, v, out, s));
  return hip_rc(hipGetLastError());
}

template <class P>
static void run_equilibrium(const EqArgs& a, hipStream_t s) {
  k_equilibrium<P><<<dim3((a.batch + 63) / 64), dim3(64), 0, s>>>(a);
}
int launch_equilibrium(const ProblemDesc& p, int batch, double r, const double* y0, const double* lb, const double* ub,
                       double* y, double* resnorm, double* residual, int* exitflag, int max_iter, double tol,
                       hipStream_t s) {
  const EqArgs a{batch, r, p.ps, p.pb, p.pmask, y0, lb, ub, y, resnorm, residual, exitflag, max_iter, tol};
  if (p.functor == Functor::User) {
    void* args[] = {(void*)&a};
    return jit_launch(p.user, UK_EQUILIBRIUM, dim3((batch + 63) / 64), dim3(64), args, s);
  }
  if (p.functor != Functor::Logistic) return -1;
  OCS_DISPATCH_LOGISTIC(p.nS, run_equilibrium<P>(a, s));
  return hip_rc(hipGetLastError());
}

// traj-major [batch][per]  ->  batch-minor [per][batch]
int launch_to_batch_minor(const double* src, double* dst, int per, int batch, hipStream_t s) {
  hipLaunchKernelGGL(k_transpose, dim3((per + 63) / 64, (batch + 63) / 64), dim3(256), 0, s, src, dst, batch,
                     per);
  return hip_rc(hipGetLastError());
}
// batch-minor [per][batch]  ->  traj-major [batch][per]
int launch_to_traj_major(const double* src, double* dst, int per, int batch, hipStream_t s) {
  hipLaunchKernelGGL(k_transpose, dim3((batch + 63) / 64, (per + 63) / 64), dim3(256), 0, s, src, dst, per,
                     batch);
  return hip_rc(hipGetLastError());
}

__global__ void k_traj_status(const double* __restrict__ J, int n, int* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) status[i] = isfinite(J[i]) ? 0 : 1;   // OCS_NUM_NONFINITE
}
int launch_traj_status(const double* J, int n, int* status, hipStream_t s) {
  hipLaunchKernelGGL(k_traj_status, dim3((n + 255) / 256), dim3(256), 0, s, J, n, status);
  return hip_rc(hipGetLastError());
}

// one workgroup: the post-reductions of a shard's objectives (ocs_multi.cpp)
// (mask: optional, entries <= 0 are left out -- the sweep index of fb_sweep, 0 = not converged)
__global__ __launch_bounds__(256) void k_objective_stats(const double* __restrict__ J, int n, int lo, double* __restrict__ out,
                                                         const int* __restrict__ mask) {
  __shared__ double ss[256], sc[256], sm[256], si[256];
  double sum = 0.0, cnt = 0.0, mn = INFINITY, am = -1.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double v = J[i];
    if (isfinite(v) && (!mask || mask[i] > 0)) {
      sum += v;
      cnt += 1.0;
      if (v < mn) {
        mn = v;
        am = (double)(lo + i);
      }
    }
  }
  ss[threadIdx.x] = sum; sc[threadIdx.x] = cnt; sm[threadIdx.x] = mn; si[threadIdx.x] = am;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      ss[threadIdx.x] += ss[threadIdx.x + w];
      sc[threadIdx.x] += sc[threadIdx.x + w];
      const double m2 = sm[threadIdx.x + w], i2 = si[threadIdx.x + w];
      // the smaller value; of equal values the smaller index (deterministic whatever the thread count)
      if (m2 < sm[threadIdx.x] || (m2 == sm[threadIdx.x] && i2 >= 0.0 && (si[threadIdx.x] < 0.0 || i2 < si[threadIdx.x]))) {
        sm[threadIdx.x] = m2;
        si[threadIdx.x] = i2;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = ss[0];
    out[1] = sc[0];
    out[2] = sm[0];
    out[3] = si[0];
  }
}
int launch_objective_stats(const double* J, int n, int lo, double* out, hipStream_t s, const int* mask) {
  hipLaunchKernelGGL(k_objective_stats, dim3(1), dim3(256), 0, s, J, n, lo, out, mask);
  return hip_rc(hipGetLastError());
}

int launch_count_nonfinite(const double* v, int n, int* count, hipStream_t s) {
  hipLaunchKernelGGL(k_count_nonfinite, dim3((n + 255) / 256), dim3(256), 0, s, v, n, count);
  return hip_rc(hipGetLastError());
}

}  // namespace ocs
