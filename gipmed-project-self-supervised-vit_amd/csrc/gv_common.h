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

// erf-GELU (nn.GELU default) and its derivative.  Phi(x) = 0.5 (1 + erf(x / sqrt 2)) through the
// 3-term Abramowitz-Stegun 7.1.25 form (|erf err| <= 2.5e-5, two orders below the bf16 rounding of
// every consumer; the exact pre-activation is what backward re-reads): with z = |x| / sqrt 2,
// t = 1 / (1 + 0.47047 z), h = 0.5 (a1 t + a2 t^2 + a3 t^3) exp(-z^2), Phi = x >= 0 ? 1 - h : h.
// One v_rcp_f32, one v_exp_f32 and ~10 plain VALU per element instead of libm erff's ~40 in a
// GEMM epilogue.  exp(-z^2) = exp(-x^2 / 2) is the Gaussian factor of gelu', so dgelu shares it.
__device__ __forceinline__ void phi_parts(float x, float& phi, float& gauss) {
    const float z = fabsf(x);
    const float t = __frcp_rn(fmaf(0.33267253f, z, 1.0f));            // 0.47047 / sqrt(2)
    gauss = exp2f(x * x * -0.72134752f);                              // exp(-x^2 / 2) = 2^(-x^2 log2(e) / 2)
    float p = fmaf(0.3739278f, t, -0.0479399f);                       // 0.5 * (a3 t + a2)
    p = fmaf(p, t, 0.1740121f);                                       // 0.5 * a1
    const float h = p * t * gauss;
    phi = x >= 0.f ? 1.0f - h : h;
}
__device__ __forceinline__ float gelu_f(float x) {
    float phi, g;
    phi_parts(x, phi, g);
    return x * phi;
}
__device__ __forceinline__ float dgelu_f(float x) {
    float phi, g;
    phi_parts(x, phi, g);
    return fmaf(x * 0.39894228040143268f, g, phi);
}
