// Host-side glue: thread-local error string + ABI version.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/trl_hip.h"

static thread_local char g_err[512] = "";

void trl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* trl_last_error(void) { return g_err; }
extern "C" int trl_abi_version(void) { return 1; }
