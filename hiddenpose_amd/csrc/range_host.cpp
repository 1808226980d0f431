// Stage ranges for rocprofv3's marker trace (hp_range_*, include/hiddenpose_hip.h; SURVEY section 5: "roctx ranges per stage").
// HIP-free.  The marker library is NOT a link dependency: it is looked up at run time the first time ranges are switched on,
// so the product library loads on a box without the profiler and a disabled range costs one relaxed atomic load.
#include <dlfcn.h>

#include <atomic>
#include <cstdint>
#include <mutex>

#include "hp_host.h"

namespace hp {
namespace {
using push_fn = int (*)(const char*);
using pop_fn = int (*)();
using start_fn = uint64_t (*)(const char*);
using stop_fn = void (*)(uint64_t);

std::atomic<int> g_range_on{0};  // 0 off, 1 on (library resolved)
std::mutex g_range_mu;
void* g_range_lib = nullptr;
push_fn g_push = nullptr;
pop_fn g_pop = nullptr;
start_fn g_start = nullptr;
stop_fn g_stop = nullptr;
thread_local int t_depth = 0;

// rocprofv3 (--marker-trace) intercepts the SDK's roctx; the older roctracer library is the fallback for rocprof v1/v2.
const char* const kLibs[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};

bool resolve_locked() {
  if (g_push && g_pop) return true;
  for (const char* name : kLibs) {
    void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h) continue;
    auto pu = reinterpret_cast<push_fn>(dlsym(h, "roctxRangePushA"));
    auto po = reinterpret_cast<pop_fn>(dlsym(h, "roctxRangePop"));
    auto sa = reinterpret_cast<start_fn>(dlsym(h, "roctxRangeStartA"));
    auto so = reinterpret_cast<stop_fn>(dlsym(h, "roctxRangeStop"));
    if (pu && po && sa && so) {
      g_range_lib = h;
      g_push = pu;
      g_pop = po;
      g_start = sa;
      g_stop = so;
      return true;
    }
    dlclose(h);
  }
  return false;
}
}  // namespace
}  // namespace hp

extern "C" int hp_range_enable(int on) {
  std::lock_guard<std::mutex> lk(hp::g_range_mu);
  if (!on) {
    hp::g_range_on.store(0);
    return 0;
  }
  if (!hp::resolve_locked()) {
    hp::g_range_on.store(0);
    return 0;  // no marker library on this box: ranges stay off, which is not an error
  }
  hp::g_range_on.store(1);
  return 1;
}

extern "C" int hp_range_push(const char* name) {
  if (!name) {
    hp::set_error("hp_range_push: null name");
    return HP_ERR_BAD_ARG;
  }
  if (!hp::g_range_on.load(std::memory_order_relaxed)) return 0;
  (void)hp::g_push(name);
  return ++hp::t_depth;
}

extern "C" int hp_range_pop(void) {
  if (hp::t_depth <= 0) return 0;  // also when ranges were switched off with one still open: it is closed, not leaked
  (void)hp::g_pop();
  return --hp::t_depth;
}

extern "C" int64_t hp_range_start(const char* name) {
  if (!name) {
    hp::set_error("hp_range_start: null name");
    return HP_ERR_BAD_ARG;
  }
  if (!hp::g_range_on.load(std::memory_order_relaxed)) return 0;
  // the marker library numbers its ranges from 0; ids of this ABI are > 0 so that 0 can mean "ranges are off"
  return (int64_t)hp::g_start(name) + 1;
}

extern "C" int hp_range_stop(int64_t id) {
  if (id <= 0 || !hp::g_stop) return 0;
  hp::g_stop((uint64_t)(id - 1));
  return 0;
}
