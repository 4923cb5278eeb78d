// capi.hip - error reporting and version of the C ABI (include/sfh_amd.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void sfh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sfh_last_error(void) { return g_err; }

extern "C" int sfh_version(void) { return 100; /* 0.1.0 */ }
