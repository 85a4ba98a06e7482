// Error plumbing shared by every entry point of libsvk.so.
#include <stdarg.h>
#include <stdio.h>

#include "svk.h"

namespace svk {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace svk

extern "C" int svk_abi_version(void) { return SVK_ABI_VERSION; }
extern "C" const char* svk_last_error(void) { return svk::g_err; }
// Developer-build switches compiled into this library (bit 0: -DSVK_PA_TIMING, bit 1: -DSVK_QV_TIMING); the timing tools
// under tools/ refuse to run against a product build (0).
extern "C" int svk_build_flags(void) {
  int f = 0;
#ifdef SVK_PA_TIMING
  f |= 1;
#endif
#ifdef SVK_QV_TIMING
  f |= 2;
#endif
#ifdef SVK_KV_TIMING
  f |= 4;
#endif
#ifdef SVK_UR_TIMING
  f |= 8;
#endif
  return f;
}
