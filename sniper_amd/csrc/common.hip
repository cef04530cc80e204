// common.hip -- error reporting + version for libsniper_hip.so
#include "common.h"

#include <stdarg.h>

static thread_local char g_err[512] = "";

void sn_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

SN_EXPORT const char *sn_last_error(void) { return g_err; }
SN_EXPORT int sn_version(void) { return 100; }
