// Error plumbing + version of the C ABI (include/recoder_hip.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void rk_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *rk_last_error(void) { return g_err; }
extern "C" int rk_version(void) { return 100; }
