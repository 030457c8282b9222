// LayerNorm forward / backward for D = 192 / 384 / 768 (ViT-T/S/B), eps 1e-6.
// Replaces nn.LayerNorm in Block.norm1/norm2 and the final norm
// (vit.pyc@L138,142,195,252).  HBM-bound: one wave per row, each lane owns the same
// 3*VEC columns of every row it visits, so the column reductions of the backward
// (dgamma, dbeta, bias gradient of the preceding Linear) stay in registers.
//   fwd bytes/row: D*4 in + D*2 out;   bwd bytes/row: D*(2 + 4 + 4) in, D*(4 + 2) out.
#include "gv_common.h"

namespace {

template <int VEC> struct VecF;
template <> struct VecF<1> { using T = float; };
template <> struct VecF<2> { using T = f32x2; };
template <> struct VecF<4> { using T = f32x4; };
template <int VEC> struct VecB;
template <> struct VecB<1> { using T = bf16; };
template <> struct VecB<2> { using T = bf16x2; };
template <> struct VecB<4> { using T = bf16x4; };

template <int VEC> __device__ __forceinline__ void ldf(const float* p, float* o) {
    typename VecF<VEC>::T v = *(const typename VecF<VEC>::T*)p;
    if constexpr (VEC == 1) o[0] = v; else { _Pragma("unroll") for (int i = 0; i < VEC; ++i) o[i] = v[i]; }
}
template <int VEC> __device__ __forceinline__ void stf(float* p, const float* o) {
    typename VecF<VEC>::T v;
    if constexpr (VEC == 1) v = o[0]; else { _Pragma("unroll") for (int i = 0; i < VEC; ++i) v[i] = o[i]; }
    *(typename VecF<VEC>::T*)p = v;
}
template <int VEC> __device__ __forceinline__ void ldb(const bf16* p, float* o) {
    typename VecB<VEC>::T v = *(const typename VecB<VEC>::T*)p;
    if constexpr (VEC == 1) o[0] = (float)v; else { _Pragma("unroll") for (int i = 0; i < VEC; ++i) o[i] = (float)v[i]; }
}
template <int VEC> __device__ __forceinline__ void stb(bf16* p, const float* o) {
    typename VecB<VEC>::T v;
    if constexpr (VEC == 1) v = (bf16)o[0]; else { _Pragma("unroll") for (int i = 0; i < VEC; ++i) v[i] = (bf16)o[i]; }
    *(typename VecB<VEC>::T*)p = v;
}

// grid-stride over rows, one wave per row; the next row's loads are issued before the current
// row's reductions so every wave keeps two rows of HBM traffic in flight
// F32IO: the fp32 parity mode (f32path.hip) keeps activations in f32 -- y / dy / gb are f32 rows instead of bf16
template <int VEC, bool F32IO = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(gv_layernorm_fwd_args a) {
    constexpr int D = 192 * VEC;
    const int lane = threadIdx.x & 63;
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    float gm[3][VEC], bt[3][VEC], v[3][VEC], nv[3][VEC];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = (i * 64 + lane) * VEC;
        ldf<VEC>(a.gamma + c, gm[i]);
        ldf<VEC>(a.beta + c, bt[i]);
        ldf<VEC>(a.x + (long)row * a.x_stride + c, v[i]);
    }
    while (true) {
        const int nrow = row + stride;
        const bool more = nrow < a.rows;
        if (more) {
#pragma unroll
            for (int i = 0; i < 3; ++i) ldf<VEC>(a.x + (long)nrow * a.x_stride + (i * 64 + lane) * VEC, nv[i]);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < VEC; ++j) s += v[i][j];
        const float mean = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const float d = v[i][j] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + a.eps);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float o[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = (v[i][j] - mean) * rstd * gm[i][j] + bt[i][j];
            if constexpr (F32IO) stf<VEC>((float*)a.y + (long)row * D + (i * 64 + lane) * VEC, o);
            else stb<VEC>((bf16*)a.y + (long)row * D + (i * 64 + lane) * VEC, o);
        }
        if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
        if (!more) break;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[i][j] = nv[i][j];
        row = nrow;
    }
}

template <int VEC, bool F32IO = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(gv_layernorm_bwd_args a) {
    constexpr int D = 192 * VEC;
    __shared__ float red[4][3][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gm[3][VEC], s_dg[3][VEC], s_db[3][VEC], s_g[3][VEC];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ldf<VEC>(a.gamma + (i * 64 + lane) * VEC, gm[i]);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { s_dg[i][j] = 0.f; s_db[i][j] = 0.f; s_g[i][j] = 0.f; }
    }
    for (int row = blockIdx.x * 4 + wave; row < a.rows; row += gridDim.x * 4) {
        const float* x = a.x + (long)row * a.x_stride;
        float* g = a.g + (long)row * a.g_stride;
        const float mean = a.mean[row], rstd = a.rstd[row];
        float xh[3][VEC], wdy[3][VEC], gv[3][VEC];
        float c1 = 0.f, c2 = 0.f;
        // every load of the row is issued before the first reduction (the old residual-gradient
        // row used to be fetched only after both wave sums: one more exposed HBM round trip per row)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = (i * 64 + lane) * VEC;
            if (a.g_init) { _Pragma("unroll") for (int j = 0; j < VEC; ++j) gv[i][j] = 0.f; }
            else ldf<VEC>(g + c, gv[i]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = (i * 64 + lane) * VEC;
            float xv[VEC], dv[VEC];
            ldf<VEC>(x + c, xv);
            if constexpr (F32IO) ldf<VEC>((const float*)a.dy + (long)row * D + c, dv);
            else ldb<VEC>((const bf16*)a.dy + (long)row * D + c, dv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                xh[i][j] = (xv[j] - mean) * rstd;
                wdy[i][j] = dv[j] * gm[i][j];
                c1 += wdy[i][j];
                c2 += wdy[i][j] * xh[i][j];
                s_dg[i][j] += dv[j] * xh[i][j];
                s_db[i][j] += dv[j];
            }
        }
        c1 = wave_sum(c1) * (1.0f / D);
        c2 = wave_sum(c2) * (1.0f / D);
        // stochastic depth: the gradient that enters the branch in front of this residual add (and that branch's bias
        // gradient, the third column sum) carries the branch's per-row factor; the residual-stream gradient g does not
        const float gs = a.gb_scale ? a.gb_scale[row] : 1.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = (i * 64 + lane) * VEC;
            float gbv[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                gv[i][j] += rstd * (wdy[i][j] - c1 - xh[i][j] * c2);
                gbv[j] = gv[i][j] * gs;
                s_g[i][j] += gbv[j];
            }
            stf<VEC>(g + c, gv[i]);
            if (a.gb) {
                if constexpr (F32IO) stf<VEC>((float*)a.gb + (long)row * a.gb_stride + c, gbv);
                else stb<VEC>((bf16*)a.gb + (long)row * a.gb_stride + c, gbv);
            }
        }
    }
    // block reduce of the three column sums -> partials[block][3][D]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = (i * 64 + lane) * VEC + j;
            red[wave][0][c] = s_dg[i][j];
            red[wave][1][c] = s_db[i][j];
            red[wave][2][c] = s_g[i][j];
        }
    __syncthreads();
    float* out = a.partials + (long)blockIdx.x * 3 * D;
    for (int idx = threadIdx.x; idx < 3 * D; idx += 256) {
        const int w = idx / D, c = idx - w * D;
        out[idx] = red[0][w][c] + red[1][w][c] + red[2][w][c] + red[3][w][c];
    }
}

// block = 64 columns x 4 row groups; each thread sums every 4th partial row, LDS combines
__global__ __launch_bounds__(256) void colsum_finalize_kernel(gv_colsum_finalize_args a) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (c < a.C) {
        const float* p = a.partials + (long)a.which * a.C + c;
        const long stride = (long)a.n_which * a.C;
        int b = grp;
        for (; b + 4 < a.n_blocks; b += 8) { s0 += p[b * stride]; s1 += p[(b + 4) * stride]; }
        if (b < a.n_blocks) s0 += p[b * stride];
    }
    red[grp][lane] = s0 + s1;
    __syncthreads();
    if (grp == 0 && c < a.C) {
        const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        a.out[c] = a.accumulate ? a.out[c] + s : s;
    }
}

// grid (C/64, 3 outputs, 32 row slices): 576+ blocks instead of 6; slices meet through atomics
__global__ __launch_bounds__(256) void ln_finalize_kernel(gv_ln_finalize_args a) {
    __shared__ float red[4][64];
    float* out = blockIdx.y == 0 ? a.out0 : blockIdx.y == 1 ? a.out1 : a.out2;
    if (!out) return;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int per = (a.n_blocks + gridDim.z - 1) / gridDim.z;
    const int b0 = blockIdx.z * per, b1 = min(a.n_blocks, b0 + per);
    float s = 0.f;
    if (c < a.C) {
        const float* p = a.partials + (long)blockIdx.y * a.C + c;
        const long stride = 3L * a.C;
#pragma unroll 4
        for (int b = b0 + grp; b < b1; b += 4) s += p[b * stride];
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && c < a.C) atomicAdd(out + c, (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

// column sums of [rows, C]: block (x: 128 columns as 64 lanes x 2, y: row slice)
template <bool F32>
__global__ __launch_bounds__(256) void colsum_kernel(gv_colsum_args a) {
    __shared__ float red[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 128 + lane * 2;
    const int nslice = gridDim.y;
    const int per = (a.rows + nslice - 1) / nslice;
    const int r0 = blockIdx.y * per, r1 = min(a.rows, r0 + per);
    float s0 = 0.f, s1 = 0.f;
    if (c < a.C) {
        for (int r = r0 + wave; r < r1; r += 4) {
            if constexpr (F32) {
                f32x2 v = *(const f32x2*)((const float*)a.x + (long)r * a.ld + c);
                s0 += v[0]; s1 += v[1];
            } else {
                bf16x2 v = *(const bf16x2*)((const bf16*)a.x + (long)r * a.ld + c);
                s0 += (float)v[0]; s1 += (float)v[1];
            }
        }
    }
    red[wave][lane * 2] = s0; red[wave][lane * 2 + 1] = s1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int cc = blockIdx.x * 128 + threadIdx.x;
        if (cc < a.C)
            a.workspace[(long)blockIdx.y * a.C + cc] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

}  // namespace

template <bool F32IO> static int ln_fwd_launch(const gv_layernorm_fwd_args* a, void* stream) {
    GV_REQUIRE(a && a->x && a->gamma && a->beta && a->y && a->mean && a->rstd, GV_E_NULL, "gv_layernorm_fwd: null pointer");
    GV_REQUIRE(a->D == 192 || a->D == 384 || a->D == 768, GV_E_SHAPE, "gv_layernorm_fwd: D=%d not in {192,384,768}", a->D);
    GV_REQUIRE(a->rows > 0, GV_E_SHAPE, "gv_layernorm_fwd: rows must be > 0");
    GV_REQUIRE(a->x_stride % 4 == 0 && gv_aligned(a->x, 16) && gv_aligned(a->y, 8), GV_E_ALIGN, "gv_layernorm_fwd: misaligned");
    const int blocks = (a->rows + 3) / 4;
    dim3 grid(blocks < 2048 ? blocks : 2048), block(256);     // 8 workgroups (32 waves) per CU, grid-stride over rows
    hipStream_t s = (hipStream_t)stream;
    if (a->D == 192) hipLaunchKernelGGL((ln_fwd_kernel<1, F32IO>), grid, block, 0, s, *a);
    else if (a->D == 384) hipLaunchKernelGGL((ln_fwd_kernel<2, F32IO>), grid, block, 0, s, *a);
    else hipLaunchKernelGGL((ln_fwd_kernel<4, F32IO>), grid, block, 0, s, *a);
    GV_LAUNCH_CHECK("gv_layernorm_fwd");
    return GV_OK;
}
extern "C" int gv_layernorm_fwd(const gv_layernorm_fwd_args* a, void* stream) { return ln_fwd_launch<false>(a, stream); }
extern "C" int gv_layernorm_fwd_f32(const gv_layernorm_fwd_args* a, void* stream) { return ln_fwd_launch<true>(a, stream); }

template <bool F32IO> static int ln_bwd_launch(const gv_layernorm_bwd_args* a, void* stream) {
    GV_REQUIRE(a && a->dy && a->x && a->mean && a->rstd && a->gamma && a->g && a->partials, GV_E_NULL, "gv_layernorm_bwd: null pointer");
    GV_REQUIRE(a->D == 192 || a->D == 384 || a->D == 768, GV_E_SHAPE, "gv_layernorm_bwd: D=%d not in {192,384,768}", a->D);
    GV_REQUIRE(a->rows > 0, GV_E_SHAPE, "gv_layernorm_bwd: rows must be > 0");
    GV_REQUIRE(a->x_stride % 4 == 0 && a->g_stride % 4 == 0 && a->gb_stride % 4 == 0, GV_E_ALIGN, "gv_layernorm_bwd: strides must be multiples of 4");
    dim3 grid(GV_LN_PARTIAL_BLOCKS), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (a->D == 192) hipLaunchKernelGGL((ln_bwd_kernel<1, F32IO>), grid, block, 0, s, *a);
    else if (a->D == 384) hipLaunchKernelGGL((ln_bwd_kernel<2, F32IO>), grid, block, 0, s, *a);
    else hipLaunchKernelGGL((ln_bwd_kernel<4, F32IO>), grid, block, 0, s, *a);
    GV_LAUNCH_CHECK("gv_layernorm_bwd");
    return GV_OK;
}
extern "C" int gv_layernorm_bwd(const gv_layernorm_bwd_args* a, void* stream) { return ln_bwd_launch<false>(a, stream); }
extern "C" int gv_layernorm_bwd_f32(const gv_layernorm_bwd_args* a, void* stream) { return ln_bwd_launch<true>(a, stream); }

extern "C" int gv_colsum_finalize(const gv_colsum_finalize_args* a, void* stream) {
    GV_REQUIRE(a && a->partials && a->out, GV_E_NULL, "gv_colsum_finalize: null pointer");
    GV_REQUIRE(a->C > 0 && a->n_blocks > 0 && a->which >= 0 && a->which < a->n_which, GV_E_SHAPE, "gv_colsum_finalize: bad shape");
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((a->C + 63) / 64), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_colsum_finalize");
    return GV_OK;
}

extern "C" int gv_ln_finalize(const gv_ln_finalize_args* a, void* stream) {
    GV_REQUIRE(a && a->partials, GV_E_NULL, "gv_ln_finalize: null pointer");
    GV_REQUIRE(a->C > 0 && a->n_blocks > 0, GV_E_SHAPE, "gv_ln_finalize: bad shape");
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((a->C + 63) / 64, 3, 32), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_ln_finalize");
    return GV_OK;
}

extern "C" int gv_colsum(const gv_colsum_args* a, void* stream) {
    GV_REQUIRE(a && a->x && a->workspace && a->out, GV_E_NULL, "gv_colsum: null pointer");
    GV_REQUIRE(a->rows > 0 && a->C > 0 && a->C % 2 == 0 && a->ld % 2 == 0, GV_E_SHAPE, "gv_colsum: C and ld must be even");
    const int nslice = a->rows >= 64 * 16 ? 64 : (a->rows + 15) / 16;
    dim3 grid((a->C + 127) / 128, nslice), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (a->x_is_f32) hipLaunchKernelGGL(colsum_kernel<true>, grid, block, 0, s, *a);
    else hipLaunchKernelGGL(colsum_kernel<false>, grid, block, 0, s, *a);
    GV_LAUNCH_CHECK("gv_colsum");
    gv_colsum_finalize_args f{a->workspace, nslice, 1, 0, a->C, a->out, a->accumulate};
    return gv_colsum_finalize(&f, stream);
}
