// Thread-local error string + version for libdosx.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/dosx.h"

static thread_local char g_err[512] = "";

void dosx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dosx_last_error(void) { return g_err; }
extern "C" int dosx_version(void) { return 100; }
