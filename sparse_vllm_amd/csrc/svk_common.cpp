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
