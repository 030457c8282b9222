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
// gfx950 = MI355X: 256 CUs in 8 XCDs, workgroup ids dealt round-robin over the XCDs.  The full-row / wide / weight-gradient launch
// geometries (panel.hip's `id & 7` sibling groups, gemm_dw8.h's item order) assume exactly that for L2 SHARING ONLY: on another XCD
// count or dispatch order results are unchanged (the mappings are bijective, surplus workgroups exit), only sibling workgroups stop
// meeting in one L2.
extern "C" const char* gv_target(void) { return "gfx950"; }
#ifdef GV_ACT_F16
extern "C" int gv_act_format(void) { return GV_ACT_FORMAT_F16; }
#else
extern "C" int gv_act_format(void) { return GV_ACT_FORMAT_BF16; }
#endif
