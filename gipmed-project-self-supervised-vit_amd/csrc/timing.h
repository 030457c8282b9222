// Live per-kernel timing shared by the GEMM translation units (gv_linear_timing, include/gipvit.h): while enabled,
// every GEMM-class launch is bracketed by two HIP events on the launch stream, folded per kernel name on read.
#pragma once
#include "gv_common.h"

namespace gvtime {
bool enabled();
// records the start event on `s`; returns a handle for end() (or -1 when timing is off)
int begin(const char* kernel_name, double flops, double bytes, hipStream_t s);
void end(int handle, hipStream_t s);
}  // namespace gvtime

// panel.hip: gv_linear's wide bf16 products on the full-row kernel; -1 = not one of them
int gv_panel_wide(const gv_linear_args* a, hipStream_t s);
