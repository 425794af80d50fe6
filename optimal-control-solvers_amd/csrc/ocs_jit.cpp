// ocs_jit.cpp -- user-supplied OCProblem plugins: the three plugin methods arrive as device C++ source,
// are compiled together with the kernel templates by hipRTC for gfx950 and loaded as a code object.
// This restores the open plugin surface of OCProblem/OCProblem.m that the built-in registry narrows
// (SURVEY 8(f) rank 4).  hipRTC is loaded lazily with dlopen so that libocs.so itself has no link-time
// dependency on it.
#include "ocs_handles.hpp"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <memory>
#include <mutex>
#include <unordered_map>

#include "ocs_jit.hpp"
#include <cstdlib>

#include "_obj/ocs_jit_sources.inc"  // generated: the kernel headers as string literals

namespace ocs {

typedef struct _hiprtcProgram* hiprtcProgram;
typedef int hiprtcResult;
struct Rtc {
  void* lib = nullptr;
  hiprtcResult (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**);
  hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char**);
  hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*);
  hiprtcResult (*GetCode)(hiprtcProgram, char*);
  hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t*);
  hiprtcResult (*GetProgramLog)(hiprtcProgram, char*);
  hiprtcResult (*DestroyProgram)(hiprtcProgram*);
  hiprtcResult (*AddNameExpression)(hiprtcProgram, const char*);
  hiprtcResult (*GetLoweredName)(hiprtcProgram, const char*, const char**);
  const char* (*GetErrorString)(hiprtcResult);
  hiprtcResult (*Version)(int*, int*);
};

static Rtc* rtc() {
  static Rtc r;
  static int state = 0;
  if (state == 0) {
    const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    state = -1;
    if (r.lib) {
#define L(field, sym) *(void**)(&r.field) = dlsym(r.lib, sym)
      L(CreateProgram, "hiprtcCreateProgram");
      L(CompileProgram, "hiprtcCompileProgram");
      L(GetCodeSize, "hiprtcGetCodeSize");
      L(GetCode, "hiprtcGetCode");
      L(GetProgramLogSize, "hiprtcGetProgramLogSize");
      L(GetProgramLog, "hiprtcGetProgramLog");
      L(DestroyProgram, "hiprtcDestroyProgram");
      L(AddNameExpression, "hiprtcAddNameExpression");
      L(GetLoweredName, "hiprtcGetLoweredName");
      L(GetErrorString, "hiprtcGetErrorString");
      L(Version, "hiprtcVersion");
#undef L
      if (r.CreateProgram && r.CompileProgram && r.GetCodeSize && r.GetCode && r.GetProgramLogSize &&
          r.GetProgramLog && r.DestroyProgram && r.AddNameExpression && r.GetLoweredName)
        state = 1;
    }
  }
  return state == 1 ? &r : nullptr;
}

int user_chunk(int nS) { return nS <= 4 ? 4 : 1; }

static std::vector<std::string> kernel_names(int nS, int nC, bool rowsep, bool fold) {
  const std::string ch = std::to_string(user_chunk(nS));
  std::vector<std::string> n(UK_COUNT);
  n[UK_TCOEF] = "ocs::k_tcoef<ocs::UserP>";
  n[UK_BUILD_REC] = "ocs::k_build_rec<ocs::UserP>";
  n[UK_FWD_X] = "ocs::k_forward<ocs::UserP, " + ch + ", 4, true, false>";
  n[UK_FWD_J] = "ocs::k_forward<ocs::UserP, " + ch + ", 4, false, false>";
  n[UK_FWD_UCONST] = "ocs::k_forward<ocs::UserP, " + ch + ", 4, true, true>";
  n[UK_BWD_LAM_DJDU] = "ocs::k_backward<ocs::UserP, " + ch + ", 4, true, true, false>";
  n[UK_BWD_LAM] = "ocs::k_backward<ocs::UserP, " + ch + ", 4, true, false, false>";
  if (user_chunk(nS) == 4) n[UK_BWD_LAM_DJDU_XRC] = "ocs::k_backward<ocs::UserP, 4, 4, true, true, false, true>";
  n[UK_BWD_DJDU] = "ocs::k_backward<ocs::UserP, " + ch + ", 4, false, true, false>";
  n[UK_BWD_UCONST] = "ocs::k_backward<ocs::UserP, " + ch + ", 4, false, false, true>";
  n[UK_EVAL] = "ocs::k_eval<ocs::UserP>";
  n[UK_COSTATE] = "ocs::k_costate<ocs::UserP, 4>";
  n[UK_CONTROL_GRID] = "ocs::k_control_grid<ocs::UserP, false>";
  n[UK_CONTROL_PTS] = "ocs::k_control_pts<ocs::UserP>";
  if (nS <= 4) n[UK_CONTROL_PTS_SORTED] = "ocs::k_control_pts_sorted<ocs::UserP>";
  n[UK_TU_AT] = "ocs::k_tu_at<ocs::UserP>";
  n[UK_EQUILIBRIUM] = "ocs::k_equilibrium<ocs::UserP>";
  const std::string wl = std::to_string(kScanW) + ", " + std::to_string(kScanL);
  if (rowsep) {   // (entries left empty are not compiled)
    n[UK_FWD_P2_X] = "ocs::k_forward_p2<ocs::UserP, true, true, false, 0>";
    n[UK_FWD_P2_J] = "ocs::k_forward_p2<ocs::UserP, false, true, false, 0>";
    n[UK_SCAN_LAM_DJDU] = "ocs::k_backward_scan<ocs::UserP, " + wl + ", true, true, false, 0>";
    n[UK_SCAN_LAM] = "ocs::k_backward_scan<ocs::UserP, " + wl + ", true, false, false, 0>";
    n[UK_SCAN_DJDU] = "ocs::k_backward_scan<ocs::UserP, " + wl + ", false, true, false, 0>";
    n[UK_SCAN_LAM_DJDU_LT] = "ocs::k_backward_scan<ocs::UserP, " + wl + ", true, true, true, 0>";
    n[UK_SCAN_LAM_LT] = "ocs::k_backward_scan<ocs::UserP, " + wl + ", true, false, true, 0>";
    n[UK_SCAN_DJDU_LT] = "ocs::k_backward_scan<ocs::UserP, " + wl + ", false, true, true, 0>";
  }
  if (rowsep) n[UK_COSTATE_SCAN_U] = "ocs::k_costate_scan<ocs::UserP, " + wl + ", false, true>";
  if (fold) {
    n[UK_FWD_CC_UNI] = "ocs::k_forward_cc<ocs::UserP, true>";
    n[UK_FWD_CC] = "ocs::k_forward_cc<ocs::UserP, false>";
    n[UK_COSTATE_SCAN_MET] = "ocs::k_costate_scan<ocs::UserP, " + wl + ", true>";
  }
  if (!rowsep && vector_shape_ok(nS, nC)) {
    const std::string vw = std::to_string(vscan_waves(nS)) + ", " + std::to_string(kVScanL);
    n[UK_FWD_PV_X] = "ocs::k_forward_pv<ocs::UserP, true>";
    n[UK_FWD_PV_J] = "ocs::k_forward_pv<ocs::UserP, false>";
    n[UK_VSCAN_LAM_DJDU] = "ocs::k_backward_vscan<ocs::UserP, " + vw + ", true, true, false>";
    n[UK_VSCAN_LAM] = "ocs::k_backward_vscan<ocs::UserP, " + vw + ", true, false, false>";
    n[UK_VSCAN_DJDU] = "ocs::k_backward_vscan<ocs::UserP, " + vw + ", false, true, false>";
    n[UK_VSCAN_LAM_DJDU_LT] = "ocs::k_backward_vscan<ocs::UserP, " + vw + ", true, true, true>";
    n[UK_VSCAN_LAM_LT] = "ocs::k_backward_vscan<ocs::UserP, " + vw + ", true, false, true>";
    n[UK_VSCAN_DJDU_LT] = "ocs::k_backward_vscan<ocs::UserP, " + vw + ", false, true, true>";
    n[UK_COSTATE_VSCAN] = "ocs::k_costate_vscan<ocs::UserP, " + vw + ">";
  }
  return n;
}

struct Compiled {
  std::vector<char> code;
  std::vector<std::string> lowered;
};

static unsigned long long fnv1a(const void* data, size_t n, unsigned long long h) {
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < n; ++i) {
    h ^= p[i];
    h *= 1099511628211ull;
  }
  return h;
}
static std::string disk_cache_path(const std::string& src, const char* const* hdr, int nhdr, const std::vector<std::string>& names) {
  const char* off = getenv("OCS_JIT_CACHE");
  if (off && off[0] == '0') return std::string();
  std::string dir;
  if (const char* d = getenv("OCS_JIT_CACHE_DIR")) dir = d;
  else if (const char* x = getenv("XDG_CACHE_HOME")) dir = std::string(x) + "/ocs_amd";
  else if (const char* h = getenv("HOME")) dir = std::string(h) + "/.cache/ocs_amd";
  if (dir.empty()) return std::string();
  unsigned long long h1 = 1469598103934665603ull, h2 = 0x9E3779B97F4A7C15ull;
  h1 = fnv1a(src.data(), src.size(), h1);
  h2 = fnv1a(src.data(), src.size(), h2);
  for (int k = 0; k < nhdr; ++k) {
    h1 = fnv1a(hdr[k], strlen(hdr[k]), h1);
    h2 = fnv1a(hdr[k], strlen(hdr[k]), h2);
  }
  for (const std::string& n : names) {
    h1 = fnv1a(n.data(), n.size() + 1, h1);
    h2 = fnv1a(n.data(), n.size() + 1, h2);
  }
  for (size_t at = 1; at <= dir.size(); ++at)   // mkdir -p (failures simply leave the cache off: fopen fails later)
    if (at == dir.size() || dir[at] == '/') (void)mkdir(dir.substr(0, at).c_str(), 0700);
  char name[64];
  snprintf(name, sizeof(name), "/%016llx%016llx.ocsjit", h1, h2);
  return dir + name;
}
// file: "OCSJIT01" | u64 n_names | (u64 len, bytes) x n | u64 code size | code | u64 fnv of everything before
static bool disk_cache_read(const std::string& path, Compiled& c, int nnames) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  std::vector<char> buf;
  char tmp[65536];
  size_t got;
  while ((got = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
  fclose(f);
  if (buf.size() < 8 + 8 + 8 + 8 || memcmp(buf.data(), "OCSJIT01", 8) != 0) return false;
  unsigned long long want;
  memcpy(&want, buf.data() + buf.size() - 8, 8);
  if (fnv1a(buf.data(), buf.size() - 8, 1469598103934665603ull) != want) return false;
  size_t at = 8;
  auto u64 = [&](unsigned long long& v) {
    if (at + 8 > buf.size() - 8) return false;
    memcpy(&v, buf.data() + at, 8);
    at += 8;
    return true;
  };
  unsigned long long n = 0;
  if (!u64(n) || (int)n != nnames) return false;
  c.lowered.resize(n);
  for (unsigned long long k = 0; k < n; ++k) {
    unsigned long long len = 0;
    if (!u64(len) || at + len > buf.size() - 8) return false;
    c.lowered[k].assign(buf.data() + at, buf.data() + at + len);
    at += len;
  }
  unsigned long long sz = 0;
  if (!u64(sz) || at + sz != buf.size() - 8) return false;
  c.code.assign(buf.data() + at, buf.data() + at + sz);
  return true;
}
static void disk_cache_write(const std::string& path, const Compiled& c) {
  std::vector<char> buf;
  auto put = [&](const void* p, size_t n) { buf.insert(buf.end(), (const char*)p, (const char*)p + n); };
  put("OCSJIT01", 8);
  unsigned long long n = c.lowered.size();
  put(&n, 8);
  for (const std::string& s : c.lowered) {
    unsigned long long len = s.size();
    put(&len, 8);
    put(s.data(), s.size());
  }
  unsigned long long sz = c.code.size();
  put(&sz, 8);
  put(c.code.data(), c.code.size());
  const unsigned long long h = fnv1a(buf.data(), buf.size(), 1469598103934665603ull);
  put(&h, 8);
  const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return;
  const bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
  fclose(f);
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());   // (atomic: readers see whole files only)
}

// Compiles the user's source.  `load` = false stops after compilation (usable without a GPU).
int jit_build(const char* user_src, int nS, int nC, int npar, bool has_cc, bool load, UserModule** out,
              std::string& log, bool rowsep, bool ccnox) {
  Rtc* r = rtc();
  if (!r) {
    log = "hipRTC (libhiprtc.so) could not be loaded";
    return OCS_ERR_UNSUPPORTED;
  }
  std::string src;
  src += "#include <hip/hip_runtime.h>\n";
  src += "#define OCS_USER_NS " + std::to_string(nS) + "\n#define OCS_USER_NC " + std::to_string(nC) +
         "\n#define OCS_USER_NPAR " + std::to_string(npar) + "\n";
  if (const char* defs = getenv("OCS_JIT_DEFINES")) {   // tuning: "NAME=VALUE,NAME=VALUE" ahead of the kernel headers (part of the cache key)
    std::string d = defs;
    size_t at = 0;
    while (at < d.size()) {
      size_t e = d.find(',', at);
      if (e == std::string::npos) e = d.size();
      std::string one = d.substr(at, e - at);
      const size_t eq = one.find('=');
      if (!one.empty()) src += "#define " + (eq == std::string::npos ? one : one.substr(0, eq) + " " + one.substr(eq + 1)) + "\n";
      at = e + 1;
    }
  }
  if (has_cc) src += "#define OCS_USER_HAS_CONTROLCHAR 1\n";
  if (rowsep) src += "#define OCS_USER_ROWSEP 1\n";
  const bool fold = rowsep && has_cc && ccnox;
  if (fold) src += "#define OCS_USER_CC_NOX 1\n";
  src += "#include \"ocs_device_common.hpp\"\n";
  src += "constexpr int NS = OCS_USER_NS, NC = OCS_USER_NC, NPAR = OCS_USER_NPAR;\n";
  src += (npar <= 16 && !rowsep) ? "typedef const double* OCS_PARAMS;\n" : "typedef ocs::uniform_ptr OCS_PARAMS;\n";
  src += "#line 1 \"user_problem\"\n";
  src += user_src;
  src += "\n#include \"ocs_user_functor.hpp\"\n#include \"ocs_rk4_kernels.hpp\"\n#include \"ocs_fbs_device.hpp\"\n";
  if (rowsep) src += "#include \"ocs_pipeline2_kernel.hpp\"\n#include \"ocs_scan_kernel.hpp\"\n";
  if (rowsep) src += "#include \"ocs_costate_scan_kernel.hpp\"\n";
  if (fold) src += "#include \"ocs_fold_kernel.hpp\"\n";
  const bool vec = !rowsep && vector_shape_ok(nS, nC);
  if (vec) src += "#include \"ocs_pipelinev_kernel.hpp\"\n#include \"ocs_vscan_kernel.hpp\"\n#include \"ocs_costate_vscan_kernel.hpp\"\n";

  const char* hdr_src[] = {src_ocs_device_common_hpp, src_ocs_user_functor_hpp, src_ocs_rk4_kernels_hpp,
                           src_ocs_fbs_device_hpp, src_ocs_pipeline2_kernel_hpp, src_ocs_scan_kernel_hpp,
                           src_ocs_pipelinev_kernel_hpp, src_ocs_vscan_kernel_hpp, src_ocs_fold_kernel_hpp,
                           src_ocs_costate_scan_kernel_hpp, src_ocs_costate_vscan_kernel_hpp};
  const char* hdr_name[] = {"ocs_device_common.hpp", "ocs_user_functor.hpp", "ocs_rk4_kernels.hpp",
                            "ocs_fbs_device.hpp", "ocs_pipeline2_kernel.hpp", "ocs_scan_kernel.hpp",
                            "ocs_pipelinev_kernel.hpp", "ocs_vscan_kernel.hpp", "ocs_fold_kernel.hpp",
                            "ocs_costate_scan_kernel.hpp", "ocs_costate_vscan_kernel.hpp"};
  const std::vector<std::string> names = kernel_names(nS, nC, rowsep, fold);
  // Compiled code is kept for the life of the process, keyed by the full generated source (the kernel headers are fixed
  // per build of the library): creating the same problem again -- parameter studies, a test suite -- costs a module load
  // instead of 5-6 s of hipRTC.
  // One slot per source; its mutex is held while the source compiles, so concurrent creators of the same problem (the
  // worker threads of ocs_multi_*, one per device) wait for ONE compilation instead of running it N times.
  struct Slot {
    std::mutex mu;
    std::shared_ptr<const Compiled> cc;
  };
  static std::mutex cache_mu;
  static std::unordered_map<std::string, std::shared_ptr<Slot>> cache;
  std::shared_ptr<Slot> slot;
  {
    std::lock_guard<std::mutex> lk(cache_mu);
    std::shared_ptr<Slot>& sl = cache[src];
    if (!sl) sl = std::make_shared<Slot>();
    slot = sl;
  }
  std::unique_lock<std::mutex> slot_lk(slot->mu);
  std::shared_ptr<const Compiled> cc = slot->cc;
  // ... and across processes in a directory (OCS_JIT_CACHE_DIR, else $XDG_CACHE_HOME/ocs_amd, else $HOME/.cache/ocs_amd;
  // OCS_JIT_CACHE=0 switches it off): one file per (generated source, kernel headers of this build, kernel names), so a script
  // that creates the same plugin every day pays the 5-6 s of hipRTC once.
  int rtc_major = 0, rtc_minor = 0;
  if (r->Version) (void)r->Version(&rtc_major, &rtc_minor);
  const std::string disk = disk_cache_path(src + "\n// hiprtc " + std::to_string(rtc_major) + "." + std::to_string(rtc_minor) +
                                               " gfx950 -O3 -ffp-contract=fast", hdr_src, 11, names);
  if (!cc && !disk.empty()) {
    auto loaded = std::make_shared<Compiled>();
    if (disk_cache_read(disk, *loaded, UK_COUNT)) {
      cc = loaded;
      slot->cc = cc;
    }
  }
  if (!cc) {
    hiprtcProgram prog = nullptr;
    if (r->CreateProgram(&prog, src.c_str(), "ocs_user_problem.hip", 11, hdr_src, hdr_name) != 0) {
      log = "hiprtcCreateProgram failed";
      return OCS_ERR_HIP;
    }
    for (const std::string& n : names)
      if (!n.empty()) r->AddNameExpression(prog, n.c_str());
    // The include directory of the ROCm installation explicitly: hipRTC normally serves <hip/hip_runtime.h> from a built-in
    // copy, but not in every process environment (under rocprofv3 started from another directory the compilation failed
    // with "'hip/hip_runtime.h' file not found").
    const char* rocm = getenv("ROCM_PATH");
    if (!rocm || !*rocm) rocm = getenv("HIP_PATH");
    const std::string inc = std::string("-I") + ((rocm && *rocm) ? rocm : "/opt/rocm") + "/include";
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", inc.c_str()};
    const hiprtcResult rc = r->CompileProgram(prog, 5, opts);
    size_t logsz = 0;
    r->GetProgramLogSize(prog, &logsz);
    if (logsz > 1) {
      log.resize(logsz);
      r->GetProgramLog(prog, &log[0]);
    }
    if (rc != 0) {
      r->DestroyProgram(&prog);
      if (log.empty()) log = "hiprtcCompileProgram failed";
      return OCS_ERR_INVALID;
    }
    auto fresh = std::make_shared<Compiled>();
    fresh->lowered.resize(UK_COUNT);
    for (int k = 0; k < UK_COUNT; ++k) {
      if (names[k].empty()) continue;
      const char* ln = nullptr;
      if (r->GetLoweredName(prog, names[k].c_str(), &ln) != 0 || !ln) {
        r->DestroyProgram(&prog);
        log = "no lowered name for " + names[k];
        return OCS_ERR_HIP;
      }
      fresh->lowered[k] = ln;
    }
    size_t sz = 0;
    r->GetCodeSize(prog, &sz);
    fresh->code.resize(sz);
    r->GetCode(prog, fresh->code.data());
    r->DestroyProgram(&prog);
    cc = fresh;
    slot->cc = cc;
    if (!disk.empty()) disk_cache_write(disk, *fresh);
  }
  slot_lk.unlock();
  UserModule* m = new UserModule();
  m->nS = nS;
  m->nC = nC;
  m->npar = npar;
  m->has_cc = has_cc;
  m->rowsep = rowsep;
  m->vector = vec;
  m->fold = fold;
  m->tcoef_hooks = strstr(user_src, "OCS_USER_TCOEF") != nullptr || strstr(user_src, "OCS_USER_CC_TCOEF") != nullptr;
  m->chunk = user_chunk(nS);
  m->code = cc->code;
  const std::vector<std::string>& lowered = cc->lowered;
  if (load) {
    if (require_device() != OCS_OK) {
      delete m;
      log = "no HIP device to load the compiled problem on";
      return OCS_ERR_NO_DEVICE;
    }
    if (hipModuleLoadData(&m->mod, m->code.data()) != hipSuccess) {
      delete m;
      log = "hipModuleLoadData failed";
      return OCS_ERR_HIP;
    }
    for (int k = 0; k < UK_COUNT; ++k)
      if (!names[k].empty() && hipModuleGetFunction(&m->fn[k], m->mod, lowered[k].c_str()) != hipSuccess) {
        log = "kernel " + names[k] + " missing from the compiled module";
        (void)hipModuleUnload(m->mod);
        delete m;
        return OCS_ERR_HIP;
      }
    m->loaded = true;
  }
  *out = m;
  return OCS_OK;
}

bool user_rowsep(const UserModule* m) { return m && m->rowsep; }
bool user_vector(const UserModule* m) { return m && m->vector; }
bool user_fold(const UserModule* m) { return m && m->fold; }

void jit_free(UserModule* m) {
  if (!m) return;
  if (m->loaded) (void)hipModuleUnload(m->mod);
  delete m;
}

int jit_launch(const UserModule* m, int kid, dim3 grid, dim3 block, void** params, hipStream_t s, unsigned shmem) {
  if (!m || !m->loaded || !m->fn[kid]) return -1;
  const hipError_t e = hipModuleLaunchKernel(m->fn[kid], grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem, s,
                                             params, nullptr);
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace ocs

using namespace ocs;

extern "C" {

// prob = <user class>(...)  for an OCProblem whose F / dFdx_times_vec / dFdu_times_vec are given as device
// source (OCProblem/OCProblem.m:8-21); see csrc/ocs_user_functor.hpp for the contract.
int ocs_problem_create_from_source(ocs_problem* out, const char* source, int nS, int nC, const double* params,
                                   int nparams, const double* control_bounds, int has_control_char) {
  if (!out || !source || !control_bounds || (nparams > 0 && !params)) return fail(OCS_ERR_INVALID, "null argument");
  *out = nullptr;
  if (nS < 1 || nS > 64 || nC < 1 || nC > 8 || nparams < 0) return fail(OCS_ERR_SHAPE, "need 1 <= nS <= 64, 1 <= nC <= 8");
  std::string log;
  UserModule* m = nullptr;
  // flag word: bit 0 ocs_ControlChar present, bit 1 row functions, bit 2 (with both) ocs_ControlChar does not read x
  // and ocs_row_dFdy does not read u
  const bool rowsep = (has_control_char & 2) != 0;
  if ((has_control_char & 4) && (has_control_char & 3) != 3)
    return fail(OCS_ERR_INVALID, "flag bit 2 (ControlChar of the costate alone) needs row functions and ocs_ControlChar (bits 0 and 1)");
  if (rowsep && (nC != 1 || nparams > 16 || !(nS == 1 || nS == 2 || nS == 4)))
    return fail(OCS_ERR_SHAPE, "a problem given as row functions needs nC = 1, nS in {1, 2, 4} and at most 16 parameters");
  const int rc = jit_build(source, nS, nC, nparams, (has_control_char & 1) != 0, true, &m, log, rowsep, (has_control_char & 4) != 0);
  if (rc != OCS_OK) return fail(rc, "user problem: %s", log.substr(0, 400).c_str());
  ocs_problem_s* p = new ocs_problem_s();
  p->id = OCS_PROBLEM_USER;
  p->nS = nS;
  p->nC = nC;
  p->functor = Functor::User;
  p->user = m;
  p->device = current_device_or(-1);   // the code object is loaded on the device current now
  p->par.assign(params, params + nparams);
  if (p->par.empty()) p->par.push_back(0.0);
  p->user2func.resize(nparams);
  for (int k = 0; k < nparams; ++k) p->user2func[k] = k;
  p->bounds.assign(control_bounds, control_bounds + 2 * nC);
  p->version = next_version();
  // Flag bit 2 is a declaration the fold kernels rely on (they pass u = 0 to ocs_row_dFdy and x = 0 to ocs_ControlChar).
  // Its dFdy half is probed here: (dF/dy)' v of the row functions at a few random points must not change with u.
  // Its ControlChar half the same way: the clamped ControlChar(t, x, lam) must not change with x.
  if (has_control_char & 4) {
    const int k = 4, nAug = nS + 1;
    std::vector<double> t(k), y((size_t)nAug * k), u1((size_t)nC * k), u2((size_t)nC * k), v((size_t)nAug * k), g1((size_t)nAug * k),
        g2((size_t)nAug * k);
    unsigned long long sd = 0x9E3779B97F4A7C15ull;
    auto rnd = [&sd]() {
      sd = sd * 6364136223846793005ull + 1442695040888963407ull;
      return (double)(sd >> 11) / 9007199254740992.0;
    };
    for (int j = 0; j < k; ++j) t[j] = 0.5 + j;
    for (double& q : y) q = 0.5 + rnd();
    for (double& q : v) q = 0.5 + rnd();
    for (int j = 0; j < nC * k; ++j) {
      const double lo = control_bounds[j % nC], hi = control_bounds[nC + j % nC];
      const bool fin = std::isfinite(lo) && std::isfinite(hi);
      u1[j] = fin ? lo + 0.25 * (hi - lo) : 0.3;
      u2[j] = fin ? lo + 0.75 * (hi - lo) : 0.9;
    }
    int rc2 = ocs_problem_dFdx_times_vec(p, k, t.data(), y.data(), u1.data(), v.data(), g1.data());
    if (rc2 == OCS_OK) rc2 = ocs_problem_dFdx_times_vec(p, k, t.data(), y.data(), u2.data(), v.data(), g2.data());
    if (rc2 != OCS_OK) {
      ocs_problem_destroy(p);
      return rc2;
    }
    for (size_t q = 0; q < g1.size(); ++q)
      if (g1[q] != g2[q] && !(std::isnan(g1[q]) && std::isnan(g2[q]))) {
        ocs_problem_destroy(p);
        return fail(OCS_ERR_INVALID, "flag bit 2 (control from the costate alone) declares that ocs_row_dFdy does not read u, but "
                                     "(dF/dy)' v changes with u (row %d): drop the declaration", (int)(q % nAug));
      }
    std::vector<double> x1((size_t)nS * k), x2((size_t)nS * k), lm((size_t)nS * k), c1((size_t)nC * k), c2((size_t)nC * k);
    for (double& q : x1) q = 0.5 + rnd();
    for (double& q : x2) q = 1.5 + rnd();
    for (double& q : lm) q = rnd() - 0.5;
    rc2 = ocs_problem_ControlChar(p, k, t.data(), x1.data(), lm.data(), c1.data());
    if (rc2 == OCS_OK) rc2 = ocs_problem_ControlChar(p, k, t.data(), x2.data(), lm.data(), c2.data());
    if (rc2 != OCS_OK) {
      ocs_problem_destroy(p);
      return rc2;
    }
    for (size_t q = 0; q < c1.size(); ++q)
      if (c1[q] != c2[q] && !(std::isnan(c1[q]) && std::isnan(c2[q]))) {
        ocs_problem_destroy(p);
        return fail(OCS_ERR_INVALID, "flag bit 2 (control from the costate alone) declares that ocs_ControlChar does not read x, but "
                                     "its value changes with x (control %d): drop the declaration", (int)(q % nC));
      }
  }
  *out = p;
  return OCS_OK;
}

// Compile-only check of a user problem (no GPU needed): 0 if the source builds for gfx950, else the
// compiler log is in ocs_last_error().
int ocs_problem_check_source(const char* source, int nS, int nC, int nparams, int has_control_char) {
  if (!source) return fail(OCS_ERR_INVALID, "null argument");
  std::string log;
  UserModule* m = nullptr;
  const int rc = jit_build(source, nS, nC, nparams, (has_control_char & 1) != 0, false, &m, log, (has_control_char & 2) != 0,
                           (has_control_char & 4) != 0);
  if (rc != OCS_OK) return fail(rc, "user problem: %s", log.substr(0, 400).c_str());
  jit_free(m);
  return OCS_OK;
}

}  // extern "C"
