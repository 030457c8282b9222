// Library-level entry points and the thread-local error slot.
#include "gv_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void gv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int gv_version(void) { return GV_ABI_VERSION; }
extern "C" const char* gv_last_error(void) { return g_err; }
extern "C" const char* gv_target(void) { return "gfx950"; }
#ifdef GV_ACT_F16
extern "C" int gv_act_format(void) { return GV_ACT_FORMAT_F16; }
#else
extern "C" int gv_act_format(void) { return GV_ACT_FORMAT_BF16; }
#endif
