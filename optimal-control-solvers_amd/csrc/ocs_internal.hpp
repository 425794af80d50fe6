// ocs_internal.hpp -- launcher interface between the C-ABI layer (ocs_api.cpp) and the
// gfx950 kernels (ocs_kernels.hip).  Not part of the public boundary.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ocs {

// Which device functor a problem handle resolves to.
enum class Functor : int { Logistic = 1, LQ = 3, User = 100 };
struct UserModule;  // hipRTC-compiled user problem (ocs_jit.hpp)

struct LqWorkspace;   // chunk matrices and scratch of the time-parallel LQ passes (ocs_lq_kernels.hip), owned by an integrator
void lq_workspace_free(LqWorkspace* w);

struct ProblemDesc {
  Functor functor;
  int nS, nC;
  int npar;
  const double* ps;   // device: shared parameter block [npar]
  const double* pb;   // device: per-trajectory overrides [npar][batch] or nullptr
  unsigned pmask;     // bit k set -> parameter k is read from pb
  const double* lb;   // device: control lower bounds [nC]
  const double* ub;   // device: control upper bounds [nC]
  const UserModule* user = nullptr;  // set for Functor::User
  unsigned long long version = 0;    // of the handle's device-visible parameters (cache key of tables derived from them)
};

struct GridDesc {
  int N;              // nSTEPS
  const double* HT;   // device: [N][4] = {h, h/2, h/6, h/3} (host IEEE divisions, RK4Integrator.m:40,50,73,77)
  const double* T;    // device: [2N+1] grid times (obj.t)
  double* TC;         // device: [2N+1][NTC] time coefficients of the bound problem (F side)
  double* TU;         // device: [2N+1][NTU] time coefficients of its ControlChar
  double* REC;        // device: [N][rec_stride(NTC)] per-step records {h,h/2,h/6,h/3,tcA,tcM,tcB}
  double* RECS = nullptr;  // device: compact zero-padded records of the scan kernels (record of step 0), or nullptr
  bool uniform = false;    // every step has the same size (bitwise): kernels may keep h, h/2, h/6, h/3 in registers
  LqWorkspace** lqws = nullptr;  // where the integrator keeps the workspace of the time-parallel LQ passes (created on first use)
};

// Device-native layouts (batch-minor): x0 [nS][B], u [2N+1][nC][B], x [N+1][nAug][B],
// lam [N+1][nAug][B], dJdu [2N+1][nC][B], J [B], lamT [nAug][B].
int launch_tcoef(const ProblemDesc& p, const GridDesc& g, hipStream_t s);
// mapping of the serial RK4 kernels: lane-per-trajectory (ocs_kernels.hip) or row-split
// (ocs_rowsplit_kernels.hip, row-separable problems only)
enum Mapping : int { MAP_AUTO = 0, MAP_LANE = 1, MAP_ROWSPLIT = 2, MAP_PIPELINE = 3, MAP_SCAN = 4 };
// adjoint pass as a scan over time (ocs_scan_kernels.hip): row-separable problems, any N, any batch
bool scan_supported(Functor f, int nS, int nC);
// the same for a bound problem: registry problems as above, user problems if given as row functions (hipRTC instances)
bool user_rowsep(const UserModule* m);
bool scan_problem_ok(const ProblemDesc& p);
bool pipeline_problem_ok(const ProblemDesc& p);
// any OCProblem with nS <= 4, nC <= 2 (coupled rows, several controls): the vector-lane state pass
// (ocs_pipelinev_kernel.hpp; whole blocks of 8 steps, whole tiles of 64 trajectories) and the scan adjoint pass with
// dense step maps (ocs_vscan_kernel.hpp; N a multiple of scan_chunk_steps())
bool user_vector(const UserModule* m);
// a user problem given as row functions whose ocs_ControlChar reads the costate alone and whose ocs_row_dFdy does not
// read u (flag bit 2): the folded sweep of fb_sweep is instantiated for it
bool user_fold(const UserModule* m);
bool vector_problem_ok(const ProblemDesc& p);
int launch_forward_pv(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u, double* x,
                      double* J, hipStream_t s, bool no_cost_row, const int* gate, const int* frozen = nullptr);
int launch_backward_vscan(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                          const double* lamT, double* lam, double* dJdu, double* lam0, const double* pend0,
                          hipStream_t s);
// N a multiple of scan_chunk_steps(); pend0 as for launch_backward_pl
int scan_chunk_steps();
int launch_backward_scan(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                         const double* lamT, double* lam, double* dJdu, double* lam0, const double* pend0,
                         hipStream_t s);
// the scan kernels' record table: scan_recs_doubles(N) doubles at RECS_base, built from REC (stride rs, step
// constants at sco); GridDesc::RECS = RECS_base + scan_recs_front()
size_t scan_recs_doubles(int N);
size_t scan_recs_front();
int launch_build_recs(int N, int rs, int sco, const double* REC, double* RECS_base, hipStream_t s);
bool pipeline_supported(Functor f, int nS, int nC);
// tiles of `tile` trajectories: whole, or with a ragged last tile taken by a workgroup that overlaps its neighbour (tile_base,
// ocs_device_common.hpp): more than one tile and an even batch
// (odd batches were tried: with four states the overlapped instances of the sweep kernels then differ at 1e-8 between the two
//  workgroups that compute them -- rows start at odd multiples of 8 bytes and the kernels pair trajectories in 16-byte accesses;
//  one and two states were clean.  Even batches only.)
inline bool tile_ok(int batch, int tile) { return batch % tile == 0 || (batch > tile && batch % 2 == 0); }
inline int tile_count(int batch, int tile) { return (batch + tile - 1) / tile; }
bool pipeline_shape_ok(int nS, int N, int batch, bool backward);  // nSTEPS multiple of the block, batch of the tile
int pipeline_block_steps();  // the pipeline kernels take whole blocks of this many steps
// wave-specialised costate pass of the sweep (registry problems whose adjoint right-hand side does not read u)
bool costate_pl_ok(Functor f, int nS, int nC, int N, int batch);
// the same forming the pchip midpoints of x itself (no xmid array); PR: [N][costate_prec()] interval records
int costate_prec();
int launch_costate_plx(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                       const int* frozen, double* dump, double* lam, int ld, hipStream_t s, const int* gate = nullptr);
// the same pass as a scan over time (ocs_costate_scan_kernel.hpp); _met: with the convergence test of the folded sweep
bool costate_scan_ok(const ProblemDesc& p, const GridDesc& g, int batch);
bool costate_scan_u_ok(const ProblemDesc& p, const GridDesc& g, int batch);
// any user problem with nS <= 4, nC <= 2 given as full-vector methods: the costate pass as a scan with dense step maps
bool costate_vscan_ok(const ProblemDesc& p, const GridDesc& g, int batch);
int launch_costate_vscan(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                         const double* u, const int* frozen, double* lam, hipStream_t s, const int* gate);
int launch_costate_scan_u(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                          const double* u, const int* frozen, double* lam, hipStream_t s, const int* gate);
int launch_costate_scan(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                        const int* frozen, double* lam, hipStream_t s, const int* gate);
int launch_costate_scan_met(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                            const double* lb, const double* ub, double relTol, double absTol, int sweep, int* status,
                            double* maxChange, int* nactive, double* lam, hipStream_t s, const int* gate);
int launch_costate_pl(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* xmid,
                      const int* frozen, double* dump, double* lam, int ld, hipStream_t s);
// pend0: optional [B], the k1 half of column 2N of dJdu when the steps above N were integrated by another kernel
int launch_backward_pl(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                       const double* lamT, double* lam, double* dJdu, double* lam0, const double* pend0,
                       hipStream_t s);
// the same pass with a minimal recursion wave (ocs_pipeline2_kernels.hip)
int launch_forward_p2(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                      double* x, double* J, const int* frozen, int ld, hipStream_t s, bool no_cost_row,
                      const int* gate);
int launch_forward_pl(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                      double* x, double* J, const int* frozen, double* dump, int ld, hipStream_t s,
                      bool no_cost_row = false, const int* gate = nullptr);
// whether launch_forward with these shapes runs the one kernel that honours FwdOpts::gate
bool forward_gate_supported(const ProblemDesc& p, const GridDesc& g, int batch);
// workgroups (of 64/nS trajectories) up to which fb_sweep takes its wave-specialised kernels (the folded two-kernel sweep, the
// costate kernels that form the midpoints of x); OCS_FOLD_MAX_WG overrides (tuning)
int fold_wg_limit();
// ... any state pass the sweep launches (with FwdOpts::frozen set) honours the gate, LQ excepted
bool forward_gate_any(const ProblemDesc& p);
bool rowsplit_supported(Functor f, int nS, int nC);
int launch_forward_rs(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                      double* x, double* J, hipStream_t s);
int launch_backward_rs(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                       const double* lamT, double* lam, double* dJdu, double* lam0, hipStream_t s);

struct FwdOpts;
struct BwdOpts;
// matrix-core kernels of the shared-Jacobian linear-quadratic problem (ocs_lq_kernels.hip)
bool lq_supported(int nS, int nC);
int launch_tcoef_lq(const ProblemDesc& p, const GridDesc& g, hipStream_t s);
int launch_forward_lq(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                      double* x, double* J, const FwdOpts& o, hipStream_t s);
int launch_backward_lq(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                       const double* lamT, double* lam, double* dJdu, const BwdOpts& o, hipStream_t s);
// 'nearest' / 'next' of griddedInterpolant: index of the sample a query point takes, -1 = NaN (ocs_control.cpp)
int interp_sample_index(int method, int n, const double* x, double q);
int launch_eval_lq(const ProblemDesc& p, int which, int k, const double* t, const double* y, const double* u,
                   const double* v, double* out, hipStream_t s);

struct FwdOpts {
  int mapping = MAP_AUTO;
  bool uconst = false;          // u is a device [nC] vector shared by all grid points and trajectories
  const double* Jadd = nullptr; // J = Jadd + x(end,end)
  const int* frozen = nullptr;  // [B]: trajectories with frozen[b] != 0 store nothing (fb_sweep: converged instances)
  double* dump = nullptr;       // [B] scratch the stores of frozen trajectories go to
  int ld = 0;                   // row distance of the batch-minor arrays when the call covers a window of a larger
                                // batch (pointers offset by the caller, `batch` = size of the window); 0 = batch
  const int* gate = nullptr;    // device flag: the launch does nothing if *gate == 0 (only where
                                // forward_gate_supported says so; otherwise the call fails)
  bool no_cost_row = false;     // the running-objective row of x may be left unwritten (only J is wanted); honoured
                                // where it saves traffic (the pipeline kernel), ignored elsewhere
};
struct BwdOpts {
  int mapping = MAP_AUTO;
  bool uconst = false;
  double* lam0 = nullptr;       // [nAug][B]: lam(:,1)
  double* split_scratch = nullptr;  // [nAug][B]: hand-over column of a split pass when no lam array is requested
};
int launch_forward(const ProblemDesc& p, const GridDesc& g, int batch, const double* x0, const double* u,
                   double* x, double* J, const FwdOpts& o, hipStream_t s);
int launch_backward(const ProblemDesc& p, const GridDesc& g, int batch, const double* xck, const double* u,
                    const double* lamT, double* lam, double* dJdu, const BwdOpts& o, hipStream_t s);
// which: 0 F, 1 dFdx_times_vec, 2 dFdu_times_vec; column-major device arrays with k columns.
int launch_eval(const ProblemDesc& p, int which, int k, const double* t, const double* y, const double* u,
                const double* v, double* out, hipStream_t s);
// batched compute_equilibrium (k_equilibrium, ocs_rk4_kernels.hpp): y0 / y / residual [2 nS + nC][B], lb / ub [2 nS + nC]
int launch_equilibrium(const ProblemDesc& p, int batch, double r, const double* y0, const double* lb, const double* ub,
                       double* y, double* resnorm, double* residual, int* exitflag, int max_iter, double tol,
                       hipStream_t s);
// layout helpers; per = doubles per trajectory
int launch_to_batch_minor(const double* src, double* dst, int per, int batch, hipStream_t s);
int launch_to_traj_major(const double* src, double* dst, int per, int batch, hipStream_t s);
int launch_copy8(const double* src, double* dst, size_t n, hipStream_t s);
// number of non-finite entries of v[0..n) is added to *count (device int)
int launch_count_nonfinite(const double* v, int n, int* count, hipStream_t s);
// out[0..3] = {sum of the finite J[i], their number, min J, lo + index of the minimum (-1 if none is finite)}
int launch_objective_stats(const double* J, int n, int lo, double* out, hipStream_t s, const int* mask = nullptr);
// status[i] = OCS_NUM_NONFINITE if J[i] is NaN/Inf, else 0
int launch_traj_status(const double* J, int n, int* status, hipStream_t s);

// control bases (ocs_control_kernels.hip); v [nBasis][nC][B], u / dJdu [nT][nC][B]
int launch_basis_expand(int nT, int nC, int batch, const int* colptr, const int* row, const double* val,
                        const double* v, double* u, hipStream_t s);
int launch_basis_contract(int nBasis, int nC, int batch, const int* rowptr, const int* col, const double* val,
                          const double* dJdu, double* dJdv, hipStream_t s);
bool basis_dense_supported(int nBasis);
int launch_basis_dense(bool expand, int nBasis, int nT, int nC, int batch, const double* BT, const double* in,
                       double* out, hipStream_t s);
int launch_fill_rows(int ncols, int nC, int batch, const double* val, double* out, hipStream_t s);
// out[i] = a[i] + b[i]
int launch_add_vec(int n, const double* a, const double* b, double* out, hipStream_t s);
// RK4InfiniteIntegrator's tail leg (constant control): true where the wave-specialised state pass would be chosen for this grid
// and batch -- the tail then runs on samples of the constant control instead of the lane kernels (ocs_api.cpp)
bool tail_leg_wave_ok(const ProblemDesc& p, const GridDesc& g, int batch);
int launch_gather_rows(int nrows, int batch, const int* idx, const double* src, double* dst, hipStream_t s);
int launch_scatter_rows(int nrows, int batch, const int* idx, const double* src, double* dst, hipStream_t s);

// shooting objective + gradient with the (dense, <= 32 functions) control basis fused into the RK4 kernels
// (ocs_fused_control_kernels.hip).  BT16: transposed basis [2N+1][16 or 32], zero-padded (16 when nBasis <= 16)
bool fused_control_supported(Functor f, int nS, int nC, int nBasis);
int launch_forward_fc(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, const double* BT16,
                      const double* v, const double* x0, double* ck, double* J, hipStream_t s);
int launch_backward_fc(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, const double* BT16,
                       const double* v, const double* ck, double* dJdv, double* lam0, hipStream_t s);

// the same on the wave-specialised state pass and the adjoint scan, both basis products on the matrix cores
// (ocs_fused_wave_kernels.hip); BT: [2N+1][ldbt] as BT16
bool fused_wave_supported(Functor f, int nS, int nC, int nBasis, int N, int batch);
int launch_forward_fcw(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, int ldbt, const double* BT,
                       const double* v, const double* x0, double* ck, double* J, hipStream_t s);
int launch_backward_fcs(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, int ldbt, const double* BT,
                        const double* v, const double* ck, double* dJdv, double* lam0, hipStream_t s);

// the same for a banded basis (PWLinear, PWConstant; ocs_fused_banded_kernels.hip).  CT: column table
// [2N+1][fused_banded_rec()] = {w0, w1, adv, pad}, r0 the first row of column 0; dJdv must be zero-filled
bool fused_banded_supported(Functor f, int nS, int nC);
int fused_banded_rec();
int launch_forward_fb(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, int r0, const double* CT,
                      const double* v, const double* x0, double* ck, double* J, hipStream_t s);
int launch_backward_fb(const ProblemDesc& p, const GridDesc& g, int batch, int nBasis, int r0, const double* CT,
                       const double* v, const double* ck, double* dJdv, double* lam0, hipStream_t s);

// forward-backward sweep (ocs_fbs_kernels.hip)
struct FbsTables {   // pchip node tables of an integrator grid, device pointers
  int n;             // number of nodes N+1
  const double* TN;  // [n] node times
  const double* HN;  // [n-1] spacings
  const double* W1;  // [n] interior slope weights (entries 1..n-2 used)
  const double* W2;
  const double* TM;  // [n-1] interval midpoints
  const double* IH;  // [n-1] reciprocal spacings
  const double* PR;  // [n-1][costate_prec()] per-interval pchip records (launch_costate_plx)
};
// ldb: row distance of V / out when the call covers a window of a larger batch (0 = batch)
int launch_pchip_mid(const FbsTables& t, int nrows, int ld, int batch, const double* V, double* out, hipStream_t s,
                     int ldb = 0, const int* gate = nullptr);
bool costate_forms_midpoints(const ProblemDesc& p, int N, int batch);
// fb_sweep with the control update folded into the state pass (ocs_fold_kernel.hpp) and the convergence test into the
// costate pass (k_costate_plx, MET): sweeps >= 2 are two kernels
bool fold_supported(const ProblemDesc& p, const GridDesc& g, int batch);
int launch_forward_cc(const ProblemDesc& p, const GridDesc& g, int batch, const double* PR, const double* lb,
                      const double* ub, const double* x0, const double* lam, double* x, double* J, const int* frozen,
                      bool no_cost_row, const int* gate, bool first_sweep, hipStream_t s);
int launch_costate_met(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* PR,
                       const double* lb, const double* ub, double relTol, double absTol, int sweep, int* status,
                       double* maxChange, int* nactive, double* lam, hipStream_t s, const int* gate);
int launch_costate(const ProblemDesc& p, const GridDesc& g, int batch, const double* x, int ldx, const double* xmid,
                   const double* u, const int* frozen, double* dump, double* lam, hipStream_t s, int ldb = 0,
                   const double* PR = nullptr,   // xmid == NULL with PR: the kernel forms the midpoints (see below)
                   const int* gate = nullptr);
// metric (optional): [control_grid_parts(N)][B] partial maxima of the weighted change at the grid nodes (error points
// == nodes), for launch_fbs_advance
// batched single shooting (ocs_shooting.cpp): per-instance state of the spectral projected gradient iteration,
// every array [rows][B]
struct SpgArgs {
  int batch, nV, memory;
  double TolFun, TolX;
  const double *lb, *ub;      // [nV]
  double *v, *g, *d, *vt, *gt;  // [nV][B]: iterate, its gradient, direction, trial point, its gradient
  double *J, *Jt, *alpha, *lam, *gtd, *fmax;  // [B]
  double* hist;               // [memory][B] last objective values (non-monotone line search)
  int *active, *accepted, *iters;  // [B]
  int* counter;               // instances still active (which = 1) / rejected by the line search (which = 2)
};
// which: 0 init, 1 stopping test + direction + first trial, 2 Armijo test + next trial, 3 take the step, 4 verdict
int launch_spg(int which, const SpgArgs& a, int it, double* pgnorm, int* converged, hipStream_t s);
int control_grid_parts(int N);
// batched vectorInterpolant: V [n][nComp][B] -> out [nq][nComp][B]; method as OCS_INTERP_*; t: tables of the sample grid
int launch_interp(int method, const FbsTables& t, int nComp, int nq, const int* KQ, const double* SQ, int batch,
                  const double* V, double* out, hipStream_t s);
// 'pchip' with the query points sorted by interval (QS [n], QI [nq], SS [nq]: ocs_fbs_device.hpp k_interp_pchip_sorted)
int launch_interp_pchip_sorted(const FbsTables& t, int nComp, const int* QS, const int* QI, const double* SS, int batch,
                               const double* V, double* out, hipStream_t s);
int launch_control_grid(const ProblemDesc& p, const GridDesc& g, const FbsTables& t, int batch, const double* x, int ldx,
                        const double* xmid, const double* lam, double* u, const int* status, double* metric,
                        double relTol, double absTol, hipStream_t s, int ldb = 0, const int* gate = nullptr,
                        double relax = 1.0);
int launch_control_pts(const ProblemDesc& p, const FbsTables& t, int nq, const int* KQ, const double* SQ,
                       const double* TUQ, int batch, const double* x, int ldx, const double* lam, double* out,
                       const int* usel, long long odelta, double* metric, int* anyvalid, double relTol,
                       double absTol, hipStream_t s, double relax = 1.0, const int* gate = nullptr);
// error-point mode on points sorted by interval (QS [n]: offsets per interval): registry problems; parts = control_pts_run_parts(N)
int control_pts_run_parts(int N);
bool control_pts_sorted_ok(const ProblemDesc& p);
int launch_control_pts_sorted(const ProblemDesc& p, const FbsTables& t, int nq, const int* QS, const double* SQ,
                              const double* TUQ, int batch, const double* x, int ldx, const double* lam, double* out,
                              double* metric, double relTol, double absTol, hipStream_t s, double relax = 1.0,
                              const int* gate = nullptr);
int launch_tu_at(const ProblemDesc& p, int nq, const double* tq, double* TUQ, hipStream_t s);
int control_pts_parts(int nq);  // rows of the partial-maximum array `metric` [parts][B] that launch_control_pts fills
int launch_fbs_init(int batch, int nsweeps, int* usel, int* status, double* maxChange, hipStream_t s);
int launch_fbs_advance(int batch, int sweep, int nparts, const double* metric, int* anyvalid, int* usel, int* status,
                       double* maxChange, int* nactive, hipStream_t s, int ldb = 0, const int* gate = nullptr);

// registry queries (host)
bool functor_supported(Functor f, int nS, int nC);
int functor_ntc(Functor f, int nS);
int functor_ntu(Functor f, int nS);
int rec_stride_host(int ntc);
int rec_sc_offset_host(int ntc);
int rec_pad_host();
unsigned functor_tc_param_mask(Functor f, int nS);

}  // namespace ocs
