// Internal helpers shared by the translation units of libhiddenpose_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "hiddenpose_hip.h"

namespace hp {

void set_error(const char* fmt, ...);

#define HP_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      ::hp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return HP_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

#define HP_REQUIRE(cond, ...)          \
  do {                                 \
    if (!(cond)) {                     \
      ::hp::set_error(__VA_ARGS__);    \
      return HP_ERR_BAD_ARG;           \
    }                                  \
  } while (0)

// Optional per-kernel timing with HIP events on the launch stream (hp_profile_* in the C ABI).
// Disabled by default: one relaxed atomic load per launch.
struct ProfScope {
  ProfScope(const char* name, hipStream_t st);
  ~ProfScope();
  int slot = -1;
  hipStream_t st = nullptr;
  hipEvent_t stop = nullptr;
};
#define HP_PROF(name, stream) ::hp::ProfScope _hp_prof_scope(name, stream)

inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace hp
