// error.hip — last-error text and ABI version of libmappo_hip.so.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void mappo_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *mappo_last_error(void) { return g_err; }
extern "C" int mappo_abi_version(void) { return 1; }
