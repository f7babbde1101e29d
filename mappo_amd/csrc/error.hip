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
extern "C" int mappo_abi_version(void) { return 3; }   // 3: mappo_wide_layout + explicit layout argument of mappo_wide_l1_backward

ProfSlot g_prof[MAPPO_PROF_COUNT] = {};

extern "C" int mappo_profile_arm(int32_t kernel_id, void *ev_start, void *ev_stop) {
  MAPPO_REQUIRE(kernel_id >= 0 && kernel_id < MAPPO_PROF_COUNT, "profile_arm: kernel_id %d", kernel_id);
  g_prof[kernel_id].start = (hipEvent_t)ev_start;
  g_prof[kernel_id].stop = (hipEvent_t)ev_stop;
  return MAPPO_OK;
}
