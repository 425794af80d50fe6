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

/* ---- control parametrisations ---- */
#define OCS_CONTROL_PWLINEAR 1   /* Control/PWLinearControl.m   */
#define OCS_CONTROL_PWCONSTANT 2 /* Control/PWConstantControl.m */
#define OCS_CONTROL_CHEBYSHEV 3  /* Control/ChebyshevControl.m  */

/* ---- interpolation methods of ocs_interp (griddedInterpolant / vectorInterpolant) ---- */
#define OCS_INTERP_LINEAR 0
#define OCS_INTERP_PREVIOUS 2
#define OCS_INTERP_PCHIP 3

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

/* ---- Integrator (Integrator/Integrator.m:6-15) ---- */
/* obj = RK4Integrator(tspan)                          Integrator/RK4Integrator.m:16-25 */
int ocs_rk4_create(ocs_integrator *out, const double *tspan, int npts);
int ocs_integrator_destroy(ocs_integrator g);
int ocs_integrator_nsteps(ocs_integrator g, int *nsteps); /* obj.nSTEPS */
int ocs_integrator_t(ocs_integrator g, double *t);        /* obj.t, 2N+1 values */
int ocs_integrator_h(ocs_integrator g, double *h);        /* obj.h, N values    */

/* [x, J] = compute_states(obj, prob, x0, u)           RK4Integrator.m:28-56
 * host:  x0 nS x batch, u nC x (2N+1) x batch, x nAug x (N+1) x batch (may be NULL), J batch.
 * Returns OCS_NUM_NONFINITE if any J is not finite (results are still written). */
int ocs_compute_states(ocs_integrator g, ocs_problem p, int batch, const double *x0, const double *u,
                       double *x, double *J);
/* [lam, dJdu] = compute_adjoints(obj, prob, u, lamT)  RK4Integrator.m:59-121
 * lamT nAug x batch or NULL (default e_last, :63-66); dJdu may be NULL (nargout == 1). */
int ocs_compute_adjoints(ocs_integrator g, ocs_problem p, int batch, const double *u, const double *lamT,
                         double *lam, double *dJdu);
/* device, batch-minor: x0 [nS][batch], u [2N+1][nC][batch], x [N+1][nAug][batch], J [batch],
 * lamT [nAug][batch], lam [N+1][nAug][batch], dJdu [2N+1][nC][batch].
 * If x is non-NULL it doubles as the checkpoint store the adjoint pass re-reads: keep it
 * alive and unmodified until compute_adjoints_dev has run (the xK contract of the reference).
 * lam may be NULL when only dJdu is wanted. */
int ocs_compute_states_dev(ocs_integrator g, ocs_problem p, int batch, const double *x0, const double *u,
                           double *x, double *J, void *stream);
int ocs_compute_adjoints_dev(ocs_integrator g, ocs_problem p, int batch, const double *u,
                             const double *lamT, double *lam, double *dJdu, void *stream);

/* layout helpers: MATLAB (trajectory-major, [batch][cols][rows]) <-> batch-minor ([cols][rows][batch]),
 * device pointers, rows*cols doubles per trajectory. */
int ocs_to_batch_minor_dev(const double *src, double *dst, int per_traj, int batch, void *stream);
int ocs_to_traj_major_dev(const double *src, double *dst, int per_traj, int batch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OCS_H */
