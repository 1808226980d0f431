// Per-kernel HIP-event timing of the C ABI (hp_profile_*, include/hiddenpose_hip.h); error string and version: hp_error.cpp.
#include "hp_internal.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

namespace hp {
namespace {
struct ProfEntry {
  std::string name;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  int64_t launches = 0;
  double total_ms = 0.0;
};
std::atomic<int> g_prof_on{0};
std::mutex g_prof_mu;
std::vector<ProfEntry> g_prof;

void drain_locked() {
  for (auto& e : g_prof) {
    for (auto& pr : e.pending) {
      float ms = 0.f;
      if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        e.total_ms += ms;
        e.launches += 1;
      }
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
    e.pending.clear();
  }
}
}  // namespace

ProfScope::ProfScope(const char* name, hipStream_t s) {
  const int mode = g_prof_on.load(std::memory_order_relaxed);
  if (!mode) return;
  // mode 2: the matrix-core convolution families only (an event pair costs a few microseconds of GPU time per launch;
  // the ~800 small launches of a training step would add ~4 ms to a timed region that only needs the GEMM timings)
  if (mode == 2 && (strncmp(name, "conv_", 5) != 0 || strncmp(name, "conv_pack", 9) == 0)) return;
  hipEvent_t a = nullptr, b = nullptr;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int idx = -1;
  for (size_t i = 0; i < g_prof.size(); ++i)
    if (g_prof[i].name == name) idx = (int)i;
  if (idx < 0) {
    g_prof.emplace_back();
    g_prof.back().name = name;
    idx = (int)g_prof.size() - 1;
  }
  (void)hipEventRecord(a, s);
  g_prof[idx].pending.emplace_back(a, b);
  slot = idx;
  st = s;
  stop = b;
}

ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(stop, st);
}
}  // namespace hp

extern "C" int hp_profile_enable(int on) {
  hp::g_prof_on.store(on == 2 ? 2 : on ? 1 : 0);
  return HP_OK;
}

extern "C" int hp_profile_reset(void) {
  std::lock_guard<std::mutex> lk(hp::g_prof_mu);
  hp::drain_locked();
  hp::g_prof.clear();
  return HP_OK;
}

extern "C" int hp_profile_count(void) {
  std::lock_guard<std::mutex> lk(hp::g_prof_mu);
  return (int)hp::g_prof.size();
}

extern "C" int hp_profile_get(int i, char* name, int name_cap, int64_t* launches, double* total_ms) {
  std::lock_guard<std::mutex> lk(hp::g_prof_mu);
  if (i < 0 || i >= (int)hp::g_prof.size() || !name || name_cap <= 0) {
    hp::set_error("hp_profile_get: bad index %d", i);
    return HP_ERR_BAD_ARG;
  }
  hp::drain_locked();
  const auto& e = hp::g_prof[i];
  snprintf(name, (size_t)name_cap, "%s", e.name.c_str());
  if (launches) *launches = e.launches;
  if (total_ms) *total_ms = e.total_ms;
  return HP_OK;
}

