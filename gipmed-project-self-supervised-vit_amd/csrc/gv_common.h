// Shared device/host helpers for libgipvit_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/gipvit.h"

// The library's 16-bit operand / activation format is a BUILD property (gv_act_format()): bfloat16 in libgipvit_hip.so, IEEE half
// in libgipvit_hip_f16.so (-DGV_ACT_F16: the reference's --amp --amp-dtype float16 arithmetic, train.py:452-465).  The kernels are
// written once against these names; "bf16" in identifiers and comments below reads "the 16-bit format of this build".
#ifdef GV_ACT_F16
#define GV_A16 _Float16
#define GV_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#else
#define GV_A16 __bf16
#define GV_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
typedef GV_A16 bf16;
typedef __attribute__((ext_vector_type(2))) GV_A16 bf16x2;
typedef __attribute__((ext_vector_type(4))) GV_A16 bf16x4;
typedef __attribute__((ext_vector_type(8))) GV_A16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define GV_LDS __attribute__((address_space(3)))
#define GV_GLOBAL __attribute__((address_space(1)))

// ds_read_b64_tr_b16: the transposing LDS read moves 16-bit patterns, whatever they encode
#ifdef GV_ACT_F16
typedef __attribute__((ext_vector_type(4))) short gv_i16x4;
#define GV_DS_READ_TR16(p) __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((GV_LDS gv_i16x4*)(p)))
#else
#define GV_DS_READ_TR16(p) __builtin_amdgcn_ds_read_tr16_b64_v4bf16((GV_LDS bf16x4*)(p))
#endif

// ---- error plumbing (never throws across the ABI) -------------------------
void gv_set_error(const char* fmt, ...);
#define GV_FAIL(code, ...) do { gv_set_error(__VA_ARGS__); return (code); } while (0)
#define GV_REQUIRE(cond, code, ...) do { if (!(cond)) GV_FAIL(code, __VA_ARGS__); } while (0)
#define GV_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) { gv_set_error("%s launch failed: %s", name, hipGetErrorString(e_)); return (int)e_; } } while (0)

static inline bool gv_aligned(const void* p, size_t a) { return ((uintptr_t)p & (a - 1)) == 0; }

// > 64 KiB of dynamic LDS needs an opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) that is a PER-DEVICE attribute of a
// kernel: one bit per device ordinal, set with an atomic OR (host threads may race to the same launch site; setting the
// attribute twice is harmless).  Devices past ordinal 63 set it on every launch.
struct GvLdsOptIn { unsigned long long done = 0; };
static inline int gv_lds_opt_in(GvLdsOptIn& st, const void* kern, int bytes, const char* name) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    if (bit && (__atomic_load_n(&st.done, __ATOMIC_ACQUIRE) & bit)) return GV_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { gv_set_error("%s: hipFuncSetAttribute(%d bytes of LDS): %s", name, bytes, hipGetErrorString(e)); return (int)e; }
    if (bit) __atomic_fetch_or(&st.done, bit, __ATOMIC_RELEASE);
    return GV_OK;
}

// Compute units the one-workgroup-per-CU launches (full-row / wide products, grouped weight gradients) are sized for: 256, or
// 256 - C when the process leaves C CUs to RCCL's channel workgroups (GIPVIT_CU_BUDGET, set by the drivers before the library is
// loaded when the data-parallel world has more than one rank -- every such launch holds a CU's whole register file and LDS, so a
// communication workgroup cannot co-reside: without the budget it would push a 251-workgroup launch into a second round).
#include <stdlib.h>
static inline int gv_cu_budget() {
    static const int b = [] { const char* e = getenv("GIPVIT_CU_BUDGET"); const int v = e ? atoi(e) : 256; return (v >= 64 && v <= 256) ? v : 256; }();
    return b;
}

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

// Wave sum without LDS traffic: four DPP row rotations leave every lane of a 16-lane row with the row's sum, four
// v_readlane pick one lane per row.  (__shfl_xor lowers to ds_bpermute_b32: six dependent LDS round trips per sum --
// measured 21 us of a 96-us fused GEMM + LayerNorm epilogue at two sums per token row.)  Same value in all lanes.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));   // row_ror:1
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// erf-GELU (nn.GELU default) and its derivative, transcendental-free: both are odd-polynomial
// fits around 1/2 evaluated at the CLAMPED argument (one v_med3_f32), ~11-13 plain VALU ops per
// element instead of libm erff's ~40 (a GEMM epilogue is VALU-bound on it):
//   Phi(x)   ~ 1/2 + xc Q(xc^2),  xc = clamp(x, -4, 4),  deg-7 Q: |err| <= 4.2e-5 (|gelu err| <= 1.7e-4)
//   gelu'(x) ~ 1/2 + xd R(xd^2),  xd = clamp(x, -5, 5),  deg-9 R: |err| <= 3.8e-4
// (Chebyshev-node least squares against erf / the exact derivative; both errors are far below the
// bf16 rounding, 2^-9 relative, of every consumer; backward re-reads the exact pre-activation.)
__device__ __forceinline__ float gelu_f(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
    const float s = xc * xc;
    float q = fmaf(-1.9031826664e-09f, s, 1.4105862408e-07f);
    q = fmaf(q, s, -4.5653132491e-06f);
    q = fmaf(q, s, 8.6345539916e-05f);
    q = fmaf(q, s, -1.085383136e-03f);
    q = fmaf(q, s, 9.7898487455e-03f);
    q = fmaf(q, s, -6.6360631345e-02f);
    q = fmaf(q, s, 3.9892696491e-01f);
    return x * fmaf(xc, q, 0.5f);
}
__device__ __forceinline__ float dgelu_f(float x) {
    const float xd = __builtin_amdgcn_fmed3f(x, -5.0f, 5.0f);
    const float s = xd * xd;
    float r = fmaf(-1.154122852e-11f, s, 1.5301791658e-09f);
    r = fmaf(r, s, -8.8285028494e-08f);
    r = fmaf(r, s, 2.9218555444e-06f);
    r = fmaf(r, s, -6.1681373513e-05f);
    r = fmaf(r, s, 8.7549978582e-04f);
    r = fmaf(r, s, -8.5804168959e-03f);
    r = fmaf(r, s, 5.8243992531e-02f);
    r = fmaf(r, s, -2.6485431579e-01f);
    r = fmaf(r, s, 7.9775551922e-01f);
    return fmaf(xd, r, 0.5f);
}
