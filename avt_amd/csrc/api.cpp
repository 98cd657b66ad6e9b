// Error channel + version of the AVT gfx950 C ABI (see include/avt_hip.h).
#include <cstdarg>
#include <cstdio>
#include "../../include/avt_hip.h"

static thread_local char g_err[512] = "";

void avt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* avt_last_error(void) { return g_err; }
extern "C" int avt_abi_version(void) { return AVT_ABI_VERSION; }
