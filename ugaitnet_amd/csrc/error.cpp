// Thread-local last-error message of libugaitnet_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/ugaitnet_hip.h"

static thread_local char g_err[512] = "";

void ugn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ugn_last_error(void) { return g_err; }
extern "C" int ugn_abi_version(void) { return UGN_ABI_VERSION; }
