/* ocs_cpu_fast.c -- a TUNED CPU implementation of the bench workload, test infrastructure like the rest of oracle/
 * (only bench.py's cpu_baseline leg and tests/ use it; nothing in the product path does).
 *
 * ocs_oracle.c restates Integrator/RK4Integrator.m line by line -- one trajectory per call, all four stage states
 * cached (xK, :10,32), NaN fill, a function call per stage -- which is the right CHECKER and a poor BASELINE: it runs the
 * reference's data flow, not what a CPU can do.  This file is the same arithmetic (RK4Integrator.m:28-121 on the LogisticK
 * problem, TestOCProblem.m:22-38 per row) written for the host's vector units: batch-minor arrays exactly as the GPU path
 * gets them, blocks of 64 trajectories per thread, every inner loop a unit-stride loop over the block (AVX-512 / AVX2 via
 * omp simd), exp(-r t) tabulated once per grid point, stage states recomputed in the adjoint pass instead of cached
 * (as the GPU kernels do), FMA contraction on.  Results agree with the restatement to round-off (tests/test_oracle_kat.py).
 *
 * Layouts (batch-minor): x0 [nS][B], u [2N+1][B], x, lam [N+1][nS+1][B], dJdu [2N+1][B], J [B].   nC = 1. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLK 64
#define MAXS 8

/* F row r of LogisticK: x (m_r - x) - u ; integrand e^{-rt} (sum x^2 + c u^2) */
void ocs_fast_logistic_pair(int nS, int N, long B, const double *tspan, const double *m, double c, double rr,
                            const double *x0, const double *u, double *x, double *J, double *lam, double *dJdu,
                            int nthreads) {
  const int nA = nS + 1;
  const long nT = 2L * N + 1;
  double *E = (double *)malloc(sizeof(double) * (size_t)nT);   /* e^{-r t} on the grid (nodes + midpoints, :21-24) */
  double *H = (double *)malloc(sizeof(double) * (size_t)N);
  for (int i = 0; i < N; ++i) {
    H[i] = tspan[i + 1] - tspan[i];
    E[2 * i] = exp(-rr * tspan[i]);
    E[2 * i + 1] = exp(-rr * ((tspan[i] + tspan[i + 1]) / 2));
  }
  E[2 * N] = exp(-rr * tspan[N]);
  if (nthreads < 1) nthreads = 1;
  const long nblk = (B + BLK - 1) / BLK;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
  for (long blk = 0; blk < nblk; ++blk) {
    const long b0 = blk * BLK;
    const int nb = (int)((B - b0) < BLK ? (B - b0) : BLK);
    double y[MAXS][BLK], yc[BLK];
    /* ---------------- state pass :28-56 ---------------- */
    for (int r = 0; r < nS; ++r)
      for (int j = 0; j < nb; ++j) y[r][j] = x0[(size_t)r * B + b0 + j];
    for (int j = 0; j < nb; ++j) yc[j] = 0.0;
    for (int r = 0; r < nS; ++r) memcpy(x + (size_t)r * B + b0, y[r], sizeof(double) * nb);
    memcpy(x + (size_t)nS * B + b0, yc, sizeof(double) * nb);
    for (int i = 0; i < N; ++i) {
      const double h = H[i], hh = h / 2, h6 = h / 6, eA = E[2 * i], eM = E[2 * i + 1], eB = E[2 * i + 2];
      const double *uA = u + (size_t)(2 * i) * B + b0, *uM = uA + B, *uB = uM + B;
      double q1[BLK], q2[BLK], q3[BLK], q4[BLK];
#pragma omp simd
      for (int j = 0; j < nb; ++j) {
        q1[j] = c * uA[j] * uA[j];
        q2[j] = q3[j] = c * uM[j] * uM[j];
        q4[j] = c * uB[j] * uB[j];
      }
      for (int r = 0; r < nS; ++r) {
        const double mr = m[r];
        double *yr = y[r];
#pragma omp simd
        for (int j = 0; j < nb; ++j) {
          const double yi = yr[j];
          const double F1 = yi * (mr - yi) - uA[j];
          const double Y2 = yi + hh * F1;
          const double F2 = Y2 * (mr - Y2) - uM[j];
          const double Y3 = yi + hh * F2;
          const double F3 = Y3 * (mr - Y3) - uM[j];
          const double Y4 = yi + h * F3;
          const double F4 = Y4 * (mr - Y4) - uB[j];
          q1[j] += yi * yi;
          q2[j] += Y2 * Y2;
          q3[j] += Y3 * Y3;
          q4[j] += Y4 * Y4;
          yr[j] = yi + h6 * (F1 + 2 * F2 + 2 * F3 + F4);
        }
      }
#pragma omp simd
      for (int j = 0; j < nb; ++j) yc[j] += h6 * (eA * q1[j] + 2 * eM * q2[j] + 2 * eM * q3[j] + eB * q4[j]);
      double *xo = x + (size_t)(i + 1) * nA * B + b0;
      for (int r = 0; r < nS; ++r) memcpy(xo + (size_t)r * B, y[r], sizeof(double) * nb);
      memcpy(xo + (size_t)nS * B, yc, sizeof(double) * nb);
    }
    memcpy(J + b0, yc, sizeof(double) * nb);
    /* ---------------- adjoint pass :59-121, stage states recomputed ---------------- */
    double l[MAXS][BLK], pend[BLK];
    for (int r = 0; r < nS; ++r)
      for (int j = 0; j < nb; ++j) l[r][j] = 0.0;
    for (int j = 0; j < nb; ++j) pend[j] = 0.0;
    {
      double *lo = lam + (size_t)N * nA * B + b0;
      for (int r = 0; r < nS; ++r) memcpy(lo + (size_t)r * B, l[r], sizeof(double) * nb);
      for (int j = 0; j < nb; ++j) lo[(size_t)nS * B + j] = 1.0;
    }
    for (int i = N - 1; i >= 0; --i) {
      const double h = H[i], hh = h / 2, h6 = h / 6, h3 = h / 3, eA = E[2 * i], eM = E[2 * i + 1], eB = E[2 * i + 2];
      const double *uA = u + (size_t)(2 * i) * B + b0, *uM = uA + B, *uB = uM + B;
      const double *xi = x + (size_t)i * nA * B + b0;
      double dmid[BLK], dnode[BLK], pnew[BLK];
#pragma omp simd
      for (int j = 0; j < nb; ++j) {   /* cost-row shares of dFdu_times_vec: 2 c e u k_last, k_last = {h6, h3, h3, h6} */
        dmid[j] = 2 * c * eM * uM[j] * (h3 + h3);
        dnode[j] = pend[j] + 2 * c * eB * uB[j] * h6;
        pnew[j] = 2 * c * eA * uA[j] * h6;
      }
      for (int r = 0; r < nS; ++r) {
        const double mr = m[r];
        const double *xr = xi + (size_t)r * B;
        double *lr = l[r];
#pragma omp simd
        for (int j = 0; j < nb; ++j) {
          const double yi = xr[j], la = lr[j];
          const double F1 = yi * (mr - yi) - uA[j];
          const double Y2 = yi + hh * F1;
          const double F2 = Y2 * (mr - Y2) - uM[j];
          const double Y3 = yi + hh * F2;
          const double F3 = Y3 * (mr - Y3) - uM[j];
          const double Y4 = yi + h * F3;
          const double k4 = h6 * la;                                             /* :73 */
          const double g3 = (mr - 2 * Y4) * k4 + 2 * eB * Y4 * h6;               /* :74-75 */
          const double k3 = h3 * la + h * g3;                                    /* :77 */
          const double g2 = (mr - 2 * Y3) * k3 + 2 * eM * Y3 * h3;
          const double k2 = h3 * la + hh * g2;                                   /* :81 */
          const double g1 = (mr - 2 * Y2) * k2 + 2 * eM * Y2 * h3;
          const double k1 = h6 * la + hh * g1;                                   /* :85 */
          const double g0 = (mr - 2 * yi) * k1 + 2 * eA * yi * h6;
          lr[j] = la + g1 + g2 + g3 + g0;                                        /* :86-88 */
          dmid[j] -= k2 + k3;                                                    /* dF_r/du = -1 */
          dnode[j] -= k4;
          pnew[j] -= k1;
        }
      }
      double *lo = lam + (size_t)i * nA * B + b0;
      for (int r = 0; r < nS; ++r) memcpy(lo + (size_t)r * B, l[r], sizeof(double) * nb);
      for (int j = 0; j < nb; ++j) lo[(size_t)nS * B + j] = 1.0;
      memcpy(dJdu + (size_t)(2 * i + 1) * B + b0, dmid, sizeof(double) * nb);
      memcpy(dJdu + (size_t)(2 * i + 2) * B + b0, dnode, sizeof(double) * nb);
      memcpy(pend, pnew, sizeof(double) * nb);
    }
    memcpy(dJdu + b0, pend, sizeof(double) * nb);
  }
  free(E);
  free(H);
}
