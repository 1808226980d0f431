// Error string + version entry points of the C ABI (include/hiddenpose_hip.h).  HIP-free (see hp_host.h).
#include "hp_host.h"

namespace hp {
static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}
}  // namespace hp

extern "C" int hp_version(void) { return 100; }

extern "C" const char* hp_last_error_string(void) { return hp::g_last_error.c_str(); }
