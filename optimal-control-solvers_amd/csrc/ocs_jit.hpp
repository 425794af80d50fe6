// ocs_jit.hpp -- private interface of the hipRTC path for user-supplied problems.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace ocs {

// Launch shapes shared by the registry launchers and the hipRTC instances of the same kernel templates (the name
// strings of ocs_jit.cpp and the launch dimensions are derived from these; the launchers assert that they are the
// kernel headers' own numbers)
constexpr int kScanW = 16, kScanL = 4;   // k_backward_scan: waves per workgroup, steps per chunk
constexpr int p2_waves(int nS) { return 4 + (nS == 4 ? 2 : 4); }   // k_forward_p2 without in-kernel control expansion
constexpr int kPvWaves = 7;                                          // k_forward_pv: M, S, C x 4, J
constexpr int vscan_waves(int nS) { return nS == 1 ? 16 : (nS * nS + nS <= 12 ? 8 : 4); }   // k_backward_vscan
constexpr int kVScanL = 4;
constexpr bool vector_shape_ok(int nS, int nC) { return nS >= 1 && nS <= 4 && nC >= 1 && nC <= 2; }

enum UserKernel : int {
  UK_TCOEF = 0, UK_BUILD_REC, UK_FWD_X, UK_FWD_J, UK_FWD_UCONST, UK_BWD_LAM_DJDU, UK_BWD_LAM, UK_BWD_DJDU,
  UK_BWD_UCONST, UK_EVAL, UK_COSTATE, UK_CONTROL_GRID, UK_CONTROL_PTS, UK_TU_AT, UK_EQUILIBRIUM,
  // row-separable user problems only (OCS_USER_ROWSEP): the wave-specialised state pass and the scan adjoint pass
  UK_FWD_P2_X, UK_FWD_P2_J, UK_SCAN_LAM_DJDU, UK_SCAN_LAM, UK_SCAN_DJDU, UK_SCAN_LAM_DJDU_LT, UK_SCAN_LAM_LT, UK_SCAN_DJDU_LT,
  // problems given as the three full-vector methods (coupled rows, several controls), nS <= 4, nC <= 2: the vector-lane
  // state pass (ocs_pipelinev_kernel.hpp) and the scan adjoint pass with dense step maps (ocs_vscan_kernel.hpp)
  UK_FWD_PV_X, UK_FWD_PV_J, UK_VSCAN_LAM_DJDU, UK_VSCAN_LAM, UK_VSCAN_DJDU, UK_VSCAN_LAM_DJDU_LT, UK_VSCAN_LAM_LT, UK_VSCAN_DJDU_LT,
  // row functions + ocs_ControlChar of the costate alone (OCS_USER_CC_NOX): the two kernels of fb_sweep's folded sweep
  // (ocs_fold_kernel.hpp on a uniform / any grid, ocs_costate_scan_kernel.hpp with the convergence test)
  UK_FWD_CC_UNI, UK_FWD_CC, UK_COSTATE_SCAN_MET,
  // any problem given as row functions: the costate pass of the sweep as a scan that reads the control samples
  UK_COSTATE_SCAN_U,
  // full-vector problems with nS <= 4, nC <= 2: the costate pass of the sweep as a scan with dense step maps
  UK_COSTATE_VSCAN,
  // nS <= 4: the lane adjoint kernel with checkpoint re-integration and non-temporal stores (HBM-bound launches, full output)
  UK_BWD_LAM_DJDU_XRC,
  // nS <= 4 with ocs_ControlChar: the error-point mode of fb_sweep by runs of intervals (k_control_pts_sorted)
  UK_CONTROL_PTS_SORTED,
  UK_COUNT
};

struct UserModule {
  int nS = 0, nC = 0, npar = 0, chunk = 4;
  bool has_cc = false, loaded = false, rowsep = false, vector = false, fold = false;
  bool tcoef_hooks = false;   // the source defines OCS_USER_TCOEF / OCS_USER_CC_TCOEF (coefficients tabulated from the shared parameters)
  std::vector<char> code;
  hipModule_t mod = nullptr;
  hipFunction_t fn[UK_COUNT] = {};
};

int jit_build(const char* user_src, int nS, int nC, int npar, bool has_cc, bool load, UserModule** out,
              std::string& log, bool rowsep = false, bool ccnox = false);
void jit_free(UserModule* m);
int jit_launch(const UserModule* m, int kid, dim3 grid, dim3 block, void** params, hipStream_t s, unsigned shmem = 0);

}  // namespace ocs
