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

// replay context of the per-entry sequencing (include/recoder_hip.h rk_replay_t)
static thread_local rk_replay_t g_replay;
static thread_local bool g_replay_on = false;
extern "C" void rk_replay_set(const rk_replay_t *ctx) {
  g_replay_on = ctx != nullptr && ctx->cursor != nullptr;
  if (g_replay_on) g_replay = *ctx;
}
extern "C" void rk_replay_clear(void) { g_replay_on = false; }
const rk_replay_t *rk_replay_get(void) { return g_replay_on ? &g_replay : nullptr; }
