// ocs_trace.hpp -- roctx ranges around the C-ABI entry points (tracing): every compute call shows up as
// a named range in rocprofv3 --marker-trace / roctx-aware tools.  The marker library is looked up at run time
// (librocprofiler-sdk-roctx.so, then libroctx64.so), so libocs.so has no link-time dependency on a profiler; without
// one, or with OCS_NO_ROCTX set, a range is two predictable branches.
#pragma once
#include <dlfcn.h>

#include <cstdlib>

namespace ocs {

struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    if (getenv("OCS_NO_ROCTX")) return;
    const char* libs[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
    for (const char* l : libs) {
      void* h = dlopen(l, RTLD_LAZY | RTLD_LOCAL);
      if (!h) continue;
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (push && pop) return;
      push = nullptr;
      pop = nullptr;
    }
  }
};
inline const RoctxApi& roctx_api() {
  static const RoctxApi api;
  return api;
}
struct TraceRange {
  bool on;
  explicit TraceRange(const char* name) : on(roctx_api().push != nullptr) {
    if (on) roctx_api().push(name);
  }
  ~TraceRange() {
    if (on) roctx_api().pop();
  }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};
#define OCS_TRACE(name) ::ocs::TraceRange ocs_trace_range_(name)
inline bool roctx_available() { return roctx_api().push != nullptr; }

}  // namespace ocs
