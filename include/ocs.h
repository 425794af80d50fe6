/*
 * ocs.h -- plain-C ABI of libocs.so: MI355X (gfx950) batched RK4 state / discrete-adjoint /
 * forward-backward-sweep kernels behind the plugin surface of
 * DrDanRyan/Optimal-Control-Solvers (OCProblem / Integrator / Control + the
 * single_shooting objective and fb_sweep drivers).
 *
 * The reference is MATLAB and has no FFI; each entry point below names the MATLAB
 * method or function (file:line under the reference root) it stands in for, and
 * INTEGRATION.md shows the loadlibrary/calllib shim classes that bind them.
 * The header is C89-clean (no C++, no torch types) so MATLAB's loadlibrary parses it.
 *
 * Conventions
 *  - every export returns an int32 status: 0 ok, <0 usage/runtime error (ocs_last_error()
 *    has the text), >0 numerical condition (see OCS_NUM_*).  Nothing throws.
 *  - "host" entry points take host pointers in MATLAB shapes, column-major, with the
 *    batch as an extra trailing dimension (batch = 1 reproduces the reference's shapes
 *    exactly), and are synchronous.
 *  - "_dev" entry points take device pointers in the device-native batch-minor layout
 *    (trajectory index fastest) and are asynchronous on `stream`
 *    (a hipStream_t passed as void*; NULL = the null stream).
 *  - handles are not thread-safe; an Integrator handle is stateful exactly like the
 *    MATLAB handle class: compute_adjoints is only valid after compute_states on the
 *    same handle with the same u (RK4Integrator.m:10,32,59-61).
 *  - nAug = nS + 1 (running cost appended as last row, RK4Integrator.m:29,33);
 *    N = nSTEPS; the control grid has 2N+1 points (nodes + midpoints, :21-24).
 */
#ifndef OCS_H
#define OCS_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---- */
#define OCS_OK 0
#define OCS_ERR_INVALID (-1)     /* bad argument / null handle                    */
#define OCS_ERR_SHAPE (-2)       /* dimension mismatch                            */
#define OCS_ERR_ORDER (-3)       /* compute_adjoints before compute_states        */
#define OCS_ERR_NO_DEVICE (-4)   /* no usable MI355X / HIP runtime failure at init */
#define OCS_ERR_HIP (-5)         /* a HIP call failed                             */
#define OCS_ERR_UNSUPPORTED (-6) /* problem/shape not in the kernel registry       */
#define OCS_NUM_NONFINITE 1      /* a returned J is NaN/Inf                        */
#define OCS_NUM_NOT_CONVERGED 2  /* fb_sweep: at least one instance hit nSWEEPS    */

/* ---- problem registry (device functors; see csrc/ocs_problems.hpp) ---- */
#define OCS_PROBLEM_TEST 1     /* tests/TestOCProblem.m:22-38        params [c m r], nS=1, nC=1 */
#define OCS_PROBLEM_LOGISTIC 2 /* LogisticK       params [c r m_1..m_nS], nC=1 */
#define OCS_PROBLEM_LQ 3       /* linear-quadratic, shared Jacobian  params [r | A | Bu | q | rdiag], nS<=32, nC<=4 */
#define OCS_PROBLEM_USER 100   /* plugin methods given as device source (ocs_problem_create_from_source) */

/* ---- control parametrisations ---- */
#define OCS_CONTROL_PWLINEAR 1   /* Control/PWLinearControl.m   */
#define OCS_CONTROL_PWCONSTANT 2 /* Control/PWConstantControl.m */
#define OCS_CONTROL_CHEBYSHEV 3  /* Control/ChebyshevControl.m  */

/* ---- interpolation methods of ocs_interp (griddedInterpolant / vectorInterpolant) ---- */
#define OCS_INTERP_LINEAR 0
#define OCS_INTERP_NEAREST 1  /* the closer sample; halfway: the later one; outside the grid: the end sample */
#define OCS_INTERP_PREVIOUS 2 /* NaN before the first sample */
#define OCS_INTERP_PCHIP 3
#define OCS_INTERP_NEXT 4     /* NaN after the last sample */

typedef struct ocs_problem_s *ocs_problem;
typedef struct ocs_integrator_s *ocs_integrator;
typedef struct ocs_control_s *ocs_control;

/* ---- library ---- */
const char *ocs_version(void);
const char *ocs_last_error(void);
int ocs_device_count(int *count);
int ocs_set_device(int device);
int ocs_synchronize(void);

/* ---- OCProblem (OCProblem/OCProblem.m:3-21, tests/TestOCProblem.m:16-20) ----
 * control_bounds: nC x 2 column-major [lb ub] = prob.ControlBounds.
 * A device kernel cannot call back into a MATLAB method, so `prob` crosses the boundary
 * as a registry id + parameter block (the one unavoidable change of shape). */
int ocs_problem_create(ocs_problem *out, int problem_id, int nS, int nC, const double *params,
                       int nparams, const double *control_bounds);
/* An OCProblem whose three plugin methods (OCProblem/OCProblem.m:8-21) are given as device C++ source and
 * compiled at run time with hipRTC for gfx950; contract of the source: csrc/ocs_user_functor.hpp (functions
 * ocs_F, ocs_dFdx_times_vec, ocs_dFdu_times_vec and, if has_control_char, ocs_ControlChar for fb_sweep).
 * Runs on the lane-per-trajectory kernels; for nS <= 4, nC <= 2 also on the vector-lane state pass and the scan adjoint
 * with dense step maps (any coupled F).  Optional hooks in the source (csrc/ocs_user_functor.hpp): ocs_tcoef / ocs_cc_tcoef
 * (time coefficients tabulated once per grid point).  has_control_char is a flag word: bit 0 (value 1) -- the source defines
 * ocs_ControlChar; bit 1 (value 2) -- the problem is ROW-SEPARABLE and the source defines row functions (ocs_row_F,
 * ocs_row_q, ocs_row_dFdy, ocs_row_dFdu: row r of F reads y_r, u and t only, the integrand is a sum of per-row shares;
 * nC = 1, nS in {1, 2, 4}, at most 16 parameters) from which the three methods are derived: such a problem also runs on
 * the wave-specialised state pass and the scan adjoint pass, the mappings of the registry problems;
 * bit 2 (value 4, with bits 0 and 1) -- the problem declares that ocs_ControlChar does not read x and ocs_row_dFdy does
 * not read u (the control of the minimum principle is a function of the costate alone, functions/fb_sweep.m:94-97 with
 * the Gen-1 ControlChar of make_from_symbolic.m:33-38 for a Hamiltonian separable in (x, u)): fb_sweep then runs its
 * two-kernel sweep for the problem (ocs_fb_sweep_path == 4); ocs_ControlChar receives x = zeros, ocs_row_dFdy u = 0 there.
 * A compile error returns OCS_ERR_INVALID with the compiler log in ocs_last_error().  ocs_problem_check_source only
 * compiles (no GPU needed).  Compiled code is kept for the life of the process and, across processes, in a directory
 * (OCS_JIT_CACHE_DIR, else $XDG_CACHE_HOME/ocs_amd, else $HOME/.cache/ocs_amd; OCS_JIT_CACHE=0: off), keyed by the generated
 * source, the kernel headers of this build and the compiler version: creating a known plugin costs a file read, not 5-6 s. */
int ocs_problem_create_from_source(ocs_problem *out, const char *source, int nS, int nC, const double *params,
                                   int nparams, const double *control_bounds, int has_control_char);
int ocs_problem_check_source(const char *source, int nS, int nC, int nparams, int has_control_char);
int ocs_problem_destroy(ocs_problem p);
int ocs_problem_dims(ocs_problem p, int *nS, int *nC);
/* Per-trajectory overrides of scalar parameters (batch extension; the reference has one
 * parameter set per call).  values: nidx x batch column-major; param_index: 0-based into params. */
int ocs_problem_set_batch_params(ocs_problem p, int batch, const int *param_index, int nidx,
                                 const double *values);
/* value = F / dFdx_times_vec / dFdu_times_vec (OCProblem.m:12,16,19; TestOCProblem.m:22-38)
 * evaluated on the device for k columns; host pointers, shapes as in MATLAB. */
int ocs_problem_F(ocs_problem p, int k, const double *t, const double *y, const double *u, double *out);
int ocs_problem_dFdx_times_vec(ocs_problem p, int k, const double *t, const double *y, const double *u,
                               const double *v, double *out);
int ocs_problem_dFdu_times_vec(ocs_problem p, int k, const double *t, const double *y, const double *u,
                               const double *v, double *out);
/* value = ControlChar(t, x, lam): the Gen-1 problem method (functions/make_from_symbolic.m:33-38) with its clamp to the
 * control bounds (:111), as fb_sweep.m:96,123 calls it; k columns on the device; host pointers: t [k], x and lam nS x k,
 * out nC x k.  Registry problems and plugins whose source defines ocs_ControlChar. */
int ocs_problem_ControlChar(ocs_problem p, int k, const double *t, const double *x, const double *lam, double *out);

/* [xStar, lamStar, uStar, resnorm, residual, exitflag] = compute_equilibrium(prob, xGuess, lamGuess, uGuess, lb, ub, r)
 *                                                                   functions/compute_equilibrium.m:1-34
 * Steady state of the optimality system, one instance per trajectory of the batch (per-trajectory parameters of `p`
 * apply): y = [x; lam; u] with n = 2 nS + nC entries.  Device arrays, batch-minor: yGuess, y, residual [n][batch];
 * lb, ub [n] (shared, as in the reference's call); resnorm [batch] (squared 2-norm of the residual, lsqnonlin's
 * resnorm); exitflag [batch] (1 converged, 0 iteration limit, -1 the residual is not finite at the clamped guess or
 * became so: lsqnonlin raises an error there, nothing is reported as a solution).  residual may be NULL.  The host
 * variant takes MATLAB-shaped arrays (n x batch, column-major), batch = 1 reproduces the reference call, and it
 * returns OCS_NUM_NONFINITE when some instance has exitflag -1 (the device variant leaves that to the caller: it does
 * not synchronise).  A variable with lb == ub stays fixed.
 * lsqnonlin is a MATLAB toolbox: the iteration here is a projected Levenberg-Marquardt with the reference's residual
 * (:13-21); the root found from the same guess is the same.  Registry problems with nS <= 4 and user problems. */
int ocs_compute_equilibrium_dev(ocs_problem p, int batch, double r, const double *yGuess, const double *lb,
                                const double *ub, double *y, double *resnorm, double *residual, int *exitflag,
                                void *stream);
int ocs_compute_equilibrium(ocs_problem p, int batch, double r, const double *yGuess, const double *lb,
                            const double *ub, double *y, double *resnorm, double *residual, int *exitflag);

/* ---- Integrator (Integrator/Integrator.m:6-15) ---- */
/* obj = RK4Integrator(tspan)                          Integrator/RK4Integrator.m:16-25 */
int ocs_rk4_create(ocs_integrator *out, const double *tspan, int npts);
int ocs_integrator_destroy(ocs_integrator g);
int ocs_integrator_nsteps(ocs_integrator g, int *nsteps); /* obj.nSTEPS */
/* Thread mapping of the RK4 kernels (no counterpart in the reference): 0 automatic,
 * 1 lane-per-trajectory, 2 row-split (one state row per lane), 3 wave-specialised pipeline,
 * 4 adjoint pass as a scan over time (compute_states then runs the pipeline mapping).  2-4 are for
 * row-separable problems only (OCS_ERR_UNSUPPORTED at the next compute call otherwise).
 * Results agree to fp64 round-off. */
#define OCS_MAPPING_AUTO 0
#define OCS_MAPPING_LANE 1
#define OCS_MAPPING_ROWSPLIT 2
#define OCS_MAPPING_PIPELINE 3
#define OCS_MAPPING_SCAN 4
int ocs_integrator_set_mapping(ocs_integrator g, int mapping);
int ocs_integrator_t(ocs_integrator g, double *t);        /* obj.t, 2N+1 values */
int ocs_integrator_h(ocs_integrator g, double *h);        /* obj.h, N values    */

/* [x, J] = compute_states(obj, prob, x0, u)           RK4Integrator.m:28-56
 * host:  x0 nS x batch, u nC x (2N+1) x batch, x nAug x (N+1) x batch (may be NULL), J batch.
 * Returns OCS_NUM_NONFINITE if any J is not finite (results are still written). */
int ocs_compute_states(ocs_integrator g, ocs_problem p, int batch, const double *x0, const double *u,
                       double *x, double *J);
/* Per-trajectory status (the reference has no counterpart: it integrates one trajectory per call and never checks
 * for NaN/Inf): status[b] = OCS_NUM_NONFINITE if the objective of trajectory b is NaN/Inf -- any non-finite stage
 * state makes it so -- else 0.  Device variant: from a device array J [batch], asynchronous on `stream`; host variant:
 * the flags of the handle's last ocs_compute_states / ocs_nlp_objective call. */
int ocs_trajectory_status_dev(int batch, const double *J, int *status, void *stream);
int ocs_integrator_trajectory_status(ocs_integrator g, int batch, int *status);
/* 1 if a roctx marker library was found at run time (every compute entry point then opens a named range) */
int ocs_tracing_enabled(void);

/* [lam, dJdu] = compute_adjoints(obj, prob, u, lamT)  RK4Integrator.m:59-121
 * lamT nAug x batch or NULL (default e_last, :63-66); dJdu may be NULL (nargout == 1). */
int ocs_compute_adjoints(ocs_integrator g, ocs_problem p, int batch, const double *u, const double *lamT,
                         double *lam, double *dJdu);
/* device, batch-minor: x0 [nS][batch], u [2N+1][nC][batch], x [N+1][nAug][batch], J [batch],
 * lamT [nAug][batch], lam [N+1][nAug][batch], dJdu [2N+1][nC][batch].
 * If x is non-NULL it doubles as the checkpoint store the adjoint pass re-reads: keep it
 * alive and unmodified until compute_adjoints_dev has run (the xK contract of the reference).
 * lam may be NULL when only dJdu is wanted.
 * Streams: a handle (integrator, control, problem) owns scratch buffers and cached tables; use it from ONE stream at
 * a time.  Switching to another stream between calls is safe: the tables built on the earlier stream are ordered in
 * front of the new stream's kernels by an event; work still in flight on the earlier stream that uses the handle's
 * scratch is not -- synchronise it first. */
int ocs_compute_states_dev(ocs_integrator g, ocs_problem p, int batch, const double *x0, const double *u,
                           double *x, double *J, void *stream);
int ocs_compute_adjoints_dev(ocs_integrator g, ocs_problem p, int batch, const double *u,
                             const double *lamT, double *lam, double *dJdu, void *stream);

/* obj = RK4InfiniteIntegrator(tspan, tspanExtra, uStar)   Integrator/RK4InfiniteIntegrator.m:12-17
 * compute_states/adjoints on this handle follow :20-30 (J = J1 + J2; terminal adjoint of leg 1 =
 * lam2(:,1) of the tail leg under the constant control uStar); lamT must be NULL. */
int ocs_rk4inf_create(ocs_integrator *out, const double *tspan, int npts, const double *tspanExtra,
                      int nptsExtra, const double *uStar, int nC);

/* ---- Control (Control/Control.m:4-14 and PWLinearControl.m / PWConstantControl.m / ChebyshevControl.m) ----
 * v is a column with the control index fastest: v(c + nC*(i-1)), i = basis function. */
int ocs_control_create(ocs_control *out, int kind, const double *t, int nt, int nBasis, int nControls);
int ocs_control_destroy(ocs_control c);
int ocs_control_dims(ocs_control c, int *nBasis, int *nControls, int *nt);
int ocs_control_basis(ocs_control c, double *B);    /* property B, nBasis x nt          */
int ocs_control_points(ocs_control c, double *pts); /* controlPts / intervalStarts      */
/* u = compute_u(obj, v)            host: v (nC*nBasis) x batch -> u nC x nt x batch      */
int ocs_control_compute_u(ocs_control c, int batch, const double *v, double *u);
/* dJdv = compute_dJdv(obj, dJdu)   host: dJdu nC x nt x batch -> dJdv (nC*nBasis) x batch */
int ocs_control_compute_dJdv(ocs_control c, int batch, const double *dJdu, double *dJdv);
/* device, batch-minor: v / dJdv [nBasis][nC][batch], u / dJdu [nt][nC][batch] */
int ocs_control_compute_u_dev(ocs_control c, int batch, const double *v, double *u, void *stream);
int ocs_control_compute_dJdv_dev(ocs_control c, int batch, const double *dJdu, double *dJdv, void *stream);
/* v = compute_initial_v(obj, u0)   PWLinearControl.m:65-71, PWConstantControl.m:53-55, ChebyshevControl.m:46-48 */
int ocs_control_compute_initial_v(ocs_control c, const double *u0, int len_u0, double *v);
/* [Lb, Ub] = compute_nlp_bounds(obj, controlBounds)   PWLinearControl.m:21-28 */
int ocs_control_compute_nlp_bounds(ocs_control c, const double *bounds, double *Lb, double *Ub);
/* uFunc = compute_uFunc(obj, v); out = uFunc(tq)      PWLinearControl.m:74-77, PWConstantControl.m:58-61 */
int ocs_control_eval_uFunc(ocs_control c, const double *v, int nq, const double *tq, double *out);

/* vectorInterpolant(x, v, method)(tq)   functions/vectorInterpolant.m:1-12; v nComp x n, out nComp x nq.
 * This is the "griddedInterpolant-compatible" output: solvers return sample arrays, the caller wraps them. */
int ocs_interp(int method, int nComp, int n, const double *x, const double *v, int nq, const double *tq,
               double *out);

/* the same for a batch of sample sets resident on the device (the resampling step right after the solvers,
 * single_shooting.m:128-130, fb_sweep.m:123): x, tq host; v [n][nComp][batch] -> out [nq][nComp][batch], batch-minor */
int ocs_interp_dev(int method, int nComp, int n, const double *x, const double *v, int nq, const double *tq,
                   double *out, int batch, void *stream);

/* [J, dJdv] = nlpObjective(v)   functions/single_shooting.m:137-150
 * v holds nC*nBasis control coefficients followed by nFree free initial states; FreeInitStates are
 * 1-based state indices (host array).  x0 is overwritten at FreeInitStates (:146).
 * host: x0 nS x batch, v (nV+nFree) x batch, J batch, dJdv (nV+nFree) x batch.
 * device: the same arrays batch-minor ([rows][batch]). */
int ocs_nlp_objective(ocs_integrator g, ocs_problem p, ocs_control c, int batch, double *x0, const double *v,
                      int nFree, const int *FreeInitStates, double *J, double *dJdv);
int ocs_nlp_objective_dev(ocs_integrator g, ocs_problem p, ocs_control c, int batch, double *x0,
                          const double *v, int nFree, const int *FreeInitStates, double *J, double *dJdv,
                          void *stream);
/* soln = single_shooting(...) for a batch of independent problems (one per column of x0 / per parameter set),
 * functions/single_shooting.m:69-115 with the toolbox optimiser (fmincon 'sqp', :114) replaced by a batched spectral
 * projected gradient: Barzilai-Borwein step, projection on [Lb, Ub] (compute_nlp_bounds), non-monotone Armijo
 * back-tracking; every instance has its own step length, line search and stopping test
 * ||P(v - g) - v||_inf <= TolFun or ||step||_inf <= TolX; every evaluation is one nlpObjective over the batch.
 * Iterates differ from fmincon's, the KKT point does not. */
typedef struct {
  double TolX;        /* single_shooting.m:20 */
  double TolFun;      /* :21 */
  int MaxIter;        /* outer iterations per instance */
  int memory;         /* objective values the non-monotone line search looks back on */
  int maxBacktracks;  /* step halvings before an instance gives up */
} ocs_ss_options;
int ocs_ss_default_options(ocs_ss_options *o);
/* device arrays, batch-minor: x0 nS x batch (overwritten at FreeInitStates), v (nV+nFree) x batch: start -> solution;
 * Lb, Ub host, nV+nFree values or NULL; J batch; iterations, converged (int), pgnorm batch or NULL.
 * Returns OCS_NUM_NOT_CONVERGED if an instance ends with a projected gradient above 10 TolFun. */
int ocs_single_shooting_batch_dev(ocs_integrator g, ocs_problem p, ocs_control c, int batch, double *x0, double *v,
                                  int nFree, const int *FreeInitStates, const double *Lb, const double *Ub,
                                  const ocs_ss_options *opt, double *J, int *iterations, int *converged,
                                  double *pgnorm, void *stream);

/* For a dense basis with at most 32 functions (ChebyshevControl) on an RK4Integrator, nlpObjective can apply
 * the basis inside the RK4 kernels (u and dJdu are never materialised).  mode: 0 automatic (large batches),
 * 1 never, 2 whenever supported.  Results are the same up to the summation order of dJdu*B'. */
int ocs_control_set_fusion(ocs_control c, int mode);

/* ---- forward-backward sweep (functions/fb_sweep.m, compute_x_lam.m, compute_x_lam_J.m) ----
 * Gen-1 drivers run on a Gen-2 OCProblem through the adapter stateRHS = F(1:nS), objective = F(end),
 * adjointRHS = -dFdx_times_vec(t,[x;0],u,[lam;1])(1:nS), ControlChar = clamp(argzero dFdu_times_vec(...))
 * (make_from_symbolic.m:11-23,111).  odevr7 is replaced by RK4 on the grid of `g` (an RK4Integrator),
 * x(t)/lam(t) are pchip interpolants of the node values as in compute_x_lam.m:9,14. */
/* Fill the struct with ocs_fbs_default_options before setting fields: it has grown at its end between builds (uRelax
 * is the latest member) and may again; a caller compiled against an older header passes a shorter struct. */
typedef struct ocs_fbs_options {
  double uRelTol;  /* fb_sweep.m:16 */
  double uAbsTol;  /* :17 */
  int nSWEEPS;     /* :20 */
  int nERROR_PTS;  /* :21 */
  int nINTERP_PTS; /* :22 */
  int fused_update_off; /* build option, default 0: with the error points on the grid nodes the convergence metric is
                           taken inside the control update, and the pchip midpoints of x are formed inside the costate
                           and control kernels, and from the second sweep on the control update runs inside the state
                           pass (which reads the costate of the sweep before) and the convergence test inside the costate
                           pass; 1 keeps all of them separate kernels, 2 only the midpoints, 3 only the control update
                           of sweeps >= 2 (same results to round-off) */
  int nWINDOWS;         /* build option, default 0 = automatic (currently 1): with the fused update the batch can be cut
                           into this many windows that run their sweep loops on separate streams (marching kernels of
                           one window under the streaming kernels of another); results do not depend on it */
  int cost_row;         /* default 0: the running-objective row of xaug (row nS) is left unspecified -- soln of the
                           reference holds x, lam, u and the scalar J only (fb_sweep.m:117-125); 1 writes it */
  double uRelax;        /* extension, default 0 = off: the reference has no damping (u = uNew, :85) and its sweep may
                           oscillate or diverge (-> empty struct, :77).  0 < uRelax < 1 replaces :85 by
                           u = u + uRelax (uNew - u), on the grid and at the error points, after the unchanged test of
                           :107-110; the solution returned on convergence is that of the reference's final_sweep(u) */
} ocs_fbs_options;
int ocs_fbs_default_options(ocs_fbs_options *o);
/* [x, lam(, J)] = compute_x_lam(_J)(prob, x0, tspan, u, RelTol, AbsTol)   compute_x_lam.m:1-19, compute_x_lam_J.m:1-21
 * host: x0 nS x batch, ugrid nC x (2N+1) x batch (u sampled on the grid) -> x, lam nS x (N+1) x batch, J batch or NULL */
int ocs_compute_x_lam(ocs_integrator g, ocs_problem p, int batch, const double *x0, const double *ugrid,
                      double *x, double *lam, double *J);
/* device: xaug [N+1][nAug][batch] (states + running objective), lam [N+1][nS][batch] */
int ocs_compute_x_lam_dev(ocs_integrator g, ocs_problem p, int batch, const double *x0, const double *ugrid,
                          double *xaug, double *lam, double *J, void *stream);
/* soln = fb_sweep(prob, x0, tspan, options)   fb_sweep.m:1-126
 * u0grid / u0err: the initial control sampled on the 2N+1 grid and on linspace(T0,TF,nERROR_PTS), or both
 * NULL for the default lower bound (:23).  sweeps[b] = sweep index at which instance b converged, 0 if it
 * never did (the reference then returns an empty struct, :77).  maxChange (nSWEEPS x batch, may be NULL)
 * holds the "Normalized change in u" of :109 per sweep (NaN where not run).  uInterp = soln.u sampled on
 * linspace(T0,TF,nINTERP_PTS) (:123).  Returns OCS_NUM_NOT_CONVERGED if any instance did not converge.
 * _dev: xaug is [N+1][nS+1][batch]; its last row (the running objective) is written only with opt->cost_row. */
int ocs_fb_sweep(ocs_integrator g, ocs_problem p, int batch, const double *x0, const ocs_fbs_options *opt,
                 const double *u0grid, const double *u0err, double *x, double *lam, double *uInterp, double *J,
                 int *sweeps, double *maxChange);
int ocs_fb_sweep_dev(ocs_integrator g, ocs_problem p, int batch, const double *x0, const ocs_fbs_options *opt,
                     const double *u0grid, const double *u0err, double *xaug, double *lam, double *uInterp,
                     double *J, int *sweeps, double *maxChange, void *stream);
/* Diagnostic: the sweep loop the last ocs_fb_sweep(_dev) on this integrator ran --
 * 0 none yet; 1 the reference's sequence kernel by kernel (state pass, pchip midpoints, costate pass, ControlChar on the
 * error points / the grid, convergence bookkeeping, one host round trip per sweep); 2 the same with the control update
 * fused into one kernel and sweeps enqueued one ahead; 3 windows of the batch on their own streams; 4 the two-kernel
 * sweep (state pass with the control update folded in + costate pass with the convergence test, csrc/ocs_fold_kernel.hpp):
 * registry problems of the logistic family and hipRTC problems created with flag bit 2 (row functions, ocs_ControlChar of
 * the costate alone); 5 the sequence of 1 with the error points off the grid nodes (or a given u0), its kernels gated and
 * enqueued ahead like 2. */
int ocs_fb_sweep_path(ocs_integrator g);

/* ---- the batch axis over the GPUs of one node (the reference has no batch axis and no parallelism:
 * every entry point integrates one trajectory, tests/solve_test_problem.m:37) ----
 * ocs_multi_create: devices[n] (NULL: 0 .. n-1), one stream and one persistent host thread per device and ONE RCCL
 * communicator over them (ncclCommInitAll; librccl.so is loaded at run time -- without it, or if the communicator cannot
 * be created, the same reductions run on the host over the per-device partial results).  The caller's current HIP
 * device is the same before and after every ocs_multi_* call.  A call below cuts the batch into contiguous blocks (block k = ocs_multi_shard(m, batch, k) goes to
 * device k; sizes differ by at most one), runs the one-device host entry point of the same name on every block
 * concurrently -- no data-path exchange -- and ends with the O(1)-size reductions on the devices: all-reduce(SUM) of
 * [sum J, count] and all-gather of (min J, argmin) over RCCL.  Arrays are the MATLAB-shaped host arrays of the
 * one-device entry points for the WHOLE batch (trajectory index slowest, so a block is a contiguous range of each
 * array).  Handles own device memory: g[k], p[k], c[k] are created by the caller with device ocs_multi_device(m, k)
 * current (ocs_set_device) and describe the same problem / grid / basis on every device.
 * stats (may be NULL), 4 doubles: {sum of the finite J, their number, min J, index of the minimum in the whole batch};
 * ocs_multi_fb_sweep counts the converged instances only.  Return values as the one-device entry points (the largest
 * numerical status over the devices; the first error). */
typedef struct ocs_multi_s *ocs_multi;
int ocs_multi_create(ocs_multi *out, const int *devices, int n);
int ocs_multi_destroy(ocs_multi m);
int ocs_multi_size(ocs_multi m);
int ocs_multi_device(ocs_multi m, int k);                              /* device id of block k, or < 0 */
int ocs_multi_shard(ocs_multi m, int batch, int k, int *lo, int *hi);  /* block k = trajectories [lo, hi) */
/* [x, J] = compute_states(obj, prob, x0, u)   Integrator/RK4Integrator.m:28-56; x may be NULL */
int ocs_multi_compute_states(ocs_multi m, const ocs_integrator *g, const ocs_problem *p, int batch, const double *x0,
                             const double *u, double *x, double *J, double *stats);
/* [lam, dJdu] = compute_adjoints(obj, prob, u, lamT)   RK4Integrator.m:59-121, after ocs_multi_compute_states on the
 * same handles and batch; lamT, dJdu may be NULL */
int ocs_multi_compute_adjoints(ocs_multi m, const ocs_integrator *g, const ocs_problem *p, int batch, const double *u,
                               const double *lamT, double *lam, double *dJdu);
/* [J, dJdv] = nlpObjective(v)   functions/single_shooting.m:137-150 for a batch of candidates */
int ocs_multi_nlp_objective(ocs_multi m, const ocs_integrator *g, const ocs_problem *p, const ocs_control *c, int batch,
                            double *x0, const double *v, int nFree, const int *FreeInitStates, double *J, double *dJdv,
                            double *stats);
/* soln = fb_sweep(prob, x0, tspan, options)   functions/fb_sweep.m:1-126 for a batch of instances */
int ocs_multi_fb_sweep(ocs_multi m, const ocs_integrator *g, const ocs_problem *p, int batch, const double *x0,
                       const ocs_fbs_options *opt, const double *u0grid, const double *u0err, double *x, double *lam,
                       double *uInterp, double *J, int *sweeps, double *maxChange, double *stats);

/* 1 if the handle holds an RCCL communicator, 0 if its reductions run on the host (librccl.so missing, ncclCommInitAll
 * failed, OCS_MULTI_NO_RCCL=1; ocs_last_error() then says which) */
int ocs_multi_has_communicator(ocs_multi m);
/* Device-resident blocks (the iteration loop of functions/single_shooting.m:114,137-150 keeps its iterates on the devices):
 * every array argument is an array of n DEVICE pointers, entry k on device ocs_multi_device(m, k), batch-minor exactly as
 * the one-device _dev entry point of the same name takes it, for a block of batch[k] trajectories (blocks need not be
 * equal; block k holds the global indices [sum_{j<k} batch[j], ...)).  Nothing is copied or transposed.  The kernels are
 * enqueued on the per-device streams (ocs_multi_stream) and the call returns without waiting for them; with reduce != 0
 * the reductions -- all-reduce(SUM) of [sum J, count], all-gather of (min J, argmin) over RCCL -- are enqueued behind the
 * kernels on the same streams, and ocs_multi_stats waits for them and returns {sum of the finite J, their number, min J,
 * global index of the minimum} (ocs_multi_fb_sweep_dev: over the converged instances).  ocs_multi_synchronize waits for
 * all streams.  ocs_multi_fb_sweep_dev returns when every device's sweep loop has ended (the loop reads the count of
 * active instances after every sweep); u0 is the default lower bound (fb_sweep.m:23). */
int ocs_multi_stream(ocs_multi m, int k, void **stream);   /* hipStream_t of device k */
int ocs_multi_synchronize(ocs_multi m);
int ocs_multi_stats(ocs_multi m, double *stats);
int ocs_multi_compute_states_dev(ocs_multi m, const ocs_integrator *g, const ocs_problem *p, const int *batch,
                                 const double *const *x0, const double *const *u, double *const *x, double *const *J,
                                 int reduce);
int ocs_multi_compute_adjoints_dev(ocs_multi m, const ocs_integrator *g, const ocs_problem *p, const int *batch,
                                   const double *const *u, const double *const *lamT, double *const *lam,
                                   double *const *dJdu);
int ocs_multi_nlp_objective_dev(ocs_multi m, const ocs_integrator *g, const ocs_problem *p, const ocs_control *c,
                                const int *batch, double *const *x0, const double *const *v, int nFree,
                                const int *FreeInitStates, double *const *J, double *const *dJdv, int reduce);
int ocs_multi_fb_sweep_dev(ocs_multi m, const ocs_integrator *g, const ocs_problem *p, const int *batch,
                           const double *const *x0, const ocs_fbs_options *opt, double *const *xaug, double *const *lam,
                           double *const *uInterp, double *const *J, int *const *sweeps, double *const *maxChange,
                           int reduce);

/* device buffers for hosts without a GPU array type of their own (MATLAB through loadlibrary; torch / C callers bring
 * their own pointers): allocation on the CURRENT device (ocs_set_device), plain copies.  Copies are asynchronous on
 * `stream` when one is given and the call returns after the copy when it is NULL. */
int ocs_device_malloc(void **ptr, unsigned long bytes);
int ocs_device_free(void *ptr);
int ocs_device_upload(void *dst_device, const void *src_host, unsigned long bytes, void *stream);
int ocs_device_download(void *dst_host, const void *src_device, unsigned long bytes, void *stream);

/* layout helpers: MATLAB (trajectory-major, [batch][cols][rows]) <-> batch-minor ([cols][rows][batch]),
 * device pointers, rows*cols doubles per trajectory. */
int ocs_to_batch_minor_dev(const double *src, double *dst, int per_traj, int batch, void *stream);
int ocs_to_traj_major_dev(const double *src, double *dst, int per_traj, int batch, void *stream);
/* dst[0..n) = src[0..n) with this path's access width (8 B per lane): the known-byte-count launch used
 * to calibrate the rocprofv3 HBM counters (scripts/calibrate_traffic.py). */
int ocs_copy_dev(const double *src, double *dst, long n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OCS_H */
