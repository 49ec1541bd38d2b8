// libefg_hip.so: error reporting + version (host only).
#include "common.h"

namespace efg {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace efg

extern "C" const char* efg_last_error(void) { return efg::g_err; }
extern "C" const char* efg_version(void) { return "efg_hip 0.1 gfx950"; }
