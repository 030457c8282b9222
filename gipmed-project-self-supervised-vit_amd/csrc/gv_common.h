// Shared device/host helpers for libgipvit_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/gipvit.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define GV_LDS __attribute__((address_space(3)))
#define GV_GLOBAL __attribute__((address_space(1)))

// ---- error plumbing (never throws across the ABI) -------------------------
void gv_set_error(const char* fmt, ...);
#define GV_FAIL(code, ...) do { gv_set_error(__VA_ARGS__); return (code); } while (0)
#define GV_REQUIRE(cond, code, ...) do { if (!(cond)) GV_FAIL(code, __VA_ARGS__); } while (0)
#define GV_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) { gv_set_error("%s launch failed: %s", name, hipGetErrorString(e_)); return (int)e_; } } while (0)

static inline bool gv_aligned(const void* p, size_t a) { return ((uintptr_t)p & (a - 1)) == 0; }

// ---- wave-level reductions (64-lane wavefront) ------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact-erf GELU and its derivative (nn.GELU default)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}
