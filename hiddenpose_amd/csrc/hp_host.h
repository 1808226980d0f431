// HIP-free helpers of libhiddenpose_hip.so: the error string, argument checks, small integer helpers.  The host-only
// translation units (hp_error.cpp, lct_host.cpp, rgbe_host.cpp) include nothing else of the library, so that
// `python -m hiddenpose_amd.build --asan-host` can compile them with a plain host compiler under
// -fsanitize=address,undefined -- no device code, no HIP runtime.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <string>

#include "hiddenpose_hip.h"

namespace hp {

void set_error(const char* fmt, ...);

#define HP_REQUIRE(cond, ...)          \
  do {                                 \
    if (!(cond)) {                     \
      ::hp::set_error(__VA_ARGS__);    \
      return HP_ERR_BAD_ARG;           \
    }                                  \
  } while (0)

inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace hp
