// Small HBM-bound row / elementwise kernels of the hot path: token assembly
// (prepare_tokens, vit.pyc@L235-246), DINOHead tail (F.normalize + weight_norm,
// vit.pyc@L315-330), CLS gather, casts, reductions, the supervised softmax+LSCE loss
// (reference train.py:1046,1053) and the positional-embedding resampling matmul.
#include "gv_common.h"
#include <type_traits>

namespace {

__global__ void cls_rows_kernel(gv_cls_rows_args a) {
    const int i = blockIdx.x;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x)
        a.x[(long)i * a.N * a.D + d] = a.cls[d] + a.pos[d];
}

// grid (N tokens, D/256, image chunks): each thread sums one (token, column) over its chunk of
// images and adds the partial into dpos with one f32 atomic (dpos is zeroed by the entry point
// unless it accumulates); patch rows are also re-emitted compact in bf16.
constexpr int TOK_IMG_PER_CHUNK = 8;
template <typename PT>      // PT: element type of the compact patch rows (bf16; float in the fp32 parity mode)
__global__ void tokens_bwd_kernel(gv_tokens_bwd_args a) {
    const int t = blockIdx.x;
    const int d = blockIdx.y * blockDim.x + threadIdx.x;
    if (d >= a.D) return;
    const int P = a.N - 1;
    const int i0 = blockIdx.z * TOK_IMG_PER_CHUNK, i1 = min(a.n_img, i0 + TOK_IMG_PER_CHUNK);
    float s = 0.f;
#pragma unroll 4
    for (int i = i0; i < i1; ++i) {
        const float v = a.g[((long)i * a.N + t) * a.D + d];
        s += v;
        if (t > 0) ((PT*)a.gpatch)[((long)i * P + (t - 1)) * a.D + d] = (PT)v;
    }
    atomicAdd(a.dpos + (long)t * a.D + d, s);
    if (t == 0 && a.dcls) atomicAdd(a.dcls + d, s);
}

// (pos-embed interpolation, CLS / pos gradients, classifier head: a few thousand outputs with a short or long k-loop.
// The element types are template parameters so that the k-loop unrolls into independent loads -- with the type test
// inside the loop the K = 196 interpolation ran one dependent load pair per 330 ns: 65 us for 14 k outputs.)
template <bool ABF, bool BBF>
__global__ void small_matmul_kernel(gv_small_matmul_args a) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= a.N) return;
    typedef typename std::conditional<ABF, bf16, float>::type AT;
    typedef typename std::conditional<BBF, bf16, float>::type BT;
    const AT* A = (const AT*)a.A + (long)m * a.sam;
    const BT* B = (const BT*)a.B + (long)n * a.sbn;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= a.K; k += 8) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = (float)A[(long)(k + u) * a.sak]; bv[u] = (float)B[(long)(k + u) * a.sbk]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += av[u] * bv[u];
    }
    for (; k < a.K; ++k) s += (float)A[(long)k * a.sak] * (float)B[(long)k * a.sbk];
    if (a.bias) s += a.bias[n];
    const long ic = (long)m * a.ldc + n;
    if (a.c_is_bf16) {
        bf16* c = (bf16*)a.C + ic;
        *c = (bf16)(a.accumulate ? (float)*c + s : s);
    } else {
        float* c = (float*)a.C + ic;
        *c = a.accumulate ? *c + s : s;
    }
}

// four elements of a buffer that is bf16 on the training path and f32 in the fp32 operand mode
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 ld4<bf16>(const bf16* p) { const bf16x4 v = *(const bf16x4*)p; return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void st4(bf16* p, f32x4 v) { *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}; }

// one wave per row, C = 256 (DINOHead bottleneck): 4 columns per lane
template <typename HT>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(gv_l2norm_fwd_args a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const float* x = a.x + (long)row * a.C;
    float s = 0.f;
    for (int c = lane * 4; c < a.C; c += 256) { f32x4 v = *(const f32x4*)(x + c); s += v[0]*v[0] + v[1]*v[1] + v[2]*v[2] + v[3]*v[3]; }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    HT* y = (HT*)a.y + (long)row * a.C;
    for (int c = lane * 4; c < a.C; c += 256) {
        f32x4 v = *(const f32x4*)(x + c);
        st4(y + c, f32x4{v[0]*inv, v[1]*inv, v[2]*inv, v[3]*inv});
    }
    if (lane == 0) a.inv_norm[row] = inv;
}

template <typename HT>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(gv_l2norm_bwd_args a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const float* dy = a.dy + (long)row * a.C;
    const HT* y = (const HT*)a.y + (long)row * a.C;
    float s = 0.f;
    for (int c = lane * 4; c < a.C; c += 256) {
        f32x4 d = *(const f32x4*)(dy + c); f32x4 yy = ld4(y + c);
        s += d[0]*yy[0] + d[1]*yy[1] + d[2]*yy[2] + d[3]*yy[3];
    }
    s = wave_sum(s);
    const float inv = a.inv_norm[row];
    HT* dx = (HT*)a.dx + (long)row * a.C;
    for (int c = lane * 4; c < a.C; c += 256) {
        f32x4 d = *(const f32x4*)(dy + c); f32x4 yy = ld4(y + c);
        st4(dx + c, f32x4{(d[0] - yy[0]*s)*inv, (d[1] - yy[1]*s)*inv, (d[2] - yy[2]*s)*inv, (d[3] - yy[3]*s)*inv});
    }
}

template <typename HT>
__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(gv_weightnorm_fwd_args a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const float* v = a.v + (long)row * a.C;
    float s = 0.f;
    for (int c = lane * 4; c < a.C; c += 256) { f32x4 x = *(const f32x4*)(v + c); s += x[0]*x[0] + x[1]*x[1] + x[2]*x[2] + x[3]*x[3]; }
    const float sc = a.g[row] / sqrtf(wave_sum(s));
    HT* w = (HT*)a.w + (long)row * a.C;
    for (int c = lane * 4; c < a.C; c += 256) {
        f32x4 x = *(const f32x4*)(v + c);
        st4(w + c, f32x4{x[0]*sc, x[1]*sc, x[2]*sc, x[3]*sc});
    }
}

// C = 256 (the DINO head's bottleneck width: 65536 rows of 1 KB): a lane holds its 16 bytes of the row, so the row is read once, and
// a wave takes TWO rows, both requested before either is reduced (one 1-KB request in flight per wave left the pass at ~60 % of the
// stream rate).  Same per-lane partial sums and reduction as the generic kernels: bit-identical results.
template <typename HT>
__global__ __launch_bounds__(256) void weightnorm_fwd256_kernel(gv_weightnorm_fwd_args a) {
    const int lane = threadIdx.x & 63, r0 = blockIdx.x * 8 + (threadIdx.x >> 6), r1 = r0 + 4;
    if (r0 >= a.rows) return;
    const bool two = r1 < a.rows;
    const f32x4 x0 = *(const f32x4*)(a.v + (long)r0 * 256 + lane * 4);
    const f32x4 x1 = two ? *(const f32x4*)(a.v + (long)r1 * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float sc0 = a.g[r0] / sqrtf(wave_sum(x0[0]*x0[0] + x0[1]*x0[1] + x0[2]*x0[2] + x0[3]*x0[3]));
    st4((HT*)a.w + (long)r0 * 256 + lane * 4, f32x4{x0[0]*sc0, x0[1]*sc0, x0[2]*sc0, x0[3]*sc0});
    if (two) {
        const float sc1 = a.g[r1] / sqrtf(wave_sum(x1[0]*x1[0] + x1[1]*x1[1] + x1[2]*x1[2] + x1[3]*x1[3]));
        st4((HT*)a.w + (long)r1 * 256 + lane * 4, f32x4{x1[0]*sc1, x1[1]*sc1, x1[2]*sc1, x1[3]*sc1});
    }
}

__global__ __launch_bounds__(256) void weightnorm_bwd256_kernel(gv_weightnorm_bwd_args a) {
    const int lane = threadIdx.x & 63, r0 = blockIdx.x * 8 + (threadIdx.x >> 6);
    if (r0 >= a.rows) return;
    f32x4 x[2], d[2], pv[2];
    bool has[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = r0 + 4 * k;
        has[k] = r < a.rows;
        const long off = (long)(has[k] ? r : r0) * 256 + lane * 4;
        x[k] = *(const f32x4*)(a.v + off);
        d[k] = *(const f32x4*)(a.dw + off);
        pv[k] = a.accumulate ? *(const f32x4*)(a.dv + off) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!has[k]) continue;
        const int r = r0 + 4 * k;
        const float nn = wave_sum(x[k][0]*x[k][0] + x[k][1]*x[k][1] + x[k][2]*x[k][2] + x[k][3]*x[k][3]);
        const float dot = wave_sum(x[k][0]*d[k][0] + x[k][1]*d[k][1] + x[k][2]*d[k][2] + x[k][3]*d[k][3]);
        const float inv = 1.0f / sqrtf(nn);
        const float gs = a.g[r] * inv, vd = dot * inv;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = gs * (d[k][j] - x[k][j] * inv * vd);
        if (a.accumulate) o += pv[k];
        *(f32x4*)(a.dv + (long)r * 256 + lane * 4) = o;
        if (lane == 0 && a.dg) a.dg[r] = a.accumulate ? a.dg[r] + vd : vd;
    }
}

__global__ __launch_bounds__(256) void weightnorm_bwd_kernel(gv_weightnorm_bwd_args a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const float* v = a.v + (long)row * a.C;
    const float* dw = a.dw + (long)row * a.C;
    float nn = 0.f, dot = 0.f;
    for (int c = lane * 4; c < a.C; c += 256) {
        f32x4 x = *(const f32x4*)(v + c); f32x4 d = *(const f32x4*)(dw + c);
        nn += x[0]*x[0] + x[1]*x[1] + x[2]*x[2] + x[3]*x[3];
        dot += x[0]*d[0] + x[1]*d[1] + x[2]*d[2] + x[3]*d[3];
    }
    nn = wave_sum(nn); dot = wave_sum(dot);
    const float inv = 1.0f / sqrtf(nn);
    const float gs = a.g[row] * inv;          // g / ||v||
    const float vd = dot * inv;               // vhat . dw
    float* dv = a.dv + (long)row * a.C;
    for (int c = lane * 4; c < a.C; c += 256) {
        f32x4 x = *(const f32x4*)(v + c); f32x4 d = *(const f32x4*)(dw + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = gs * (d[j] - x[j] * inv * vd);
        if (a.accumulate) { f32x4 p = *(const f32x4*)(dv + c); o += p; }
        *(f32x4*)(dv + c) = o;
    }
    if (lane == 0 && a.dg) a.dg[row] = a.accumulate ? a.dg[row] + vd : vd;
}

__global__ void store_f32_kernel(gv_store_f32_args a) {
    if ((int)threadIdx.x < a.n) a.dst[threadIdx.x] = a.vals[threadIdx.x];
}

__global__ void gather_cls_kernel(gv_gather_cls_args a) {
    const int i = blockIdx.x;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x)
        ((bf16*)a.y)[(long)i * a.D + d] = (bf16)a.x[(long)i * a.N * a.D + d];
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(gv_cast_bf16_args a) {
    const long n4 = a.n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = ((const f32x4*)a.src)[i];
        ((bf16x4*)a.dst)[i] = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        ((bf16*)a.dst)[i] = (bf16)a.src[i];
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(gv_sumsq_args a) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = a.n >> 2;
    // two pieces per trip, both requested before the first is used (one piece in flight per lane: 4.8 TB/s over the 176-MB gradient arena)
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const f32x4 v = ((const f32x4*)a.x)[i], w = ((const f32x4*)a.x)[i + stride];
        s += v[0]*v[0] + v[1]*v[1] + v[2]*v[2] + v[3]*v[3];
        s += w[0]*w[0] + w[1]*w[1] + w[2]*w[2] + w[3]*w[3];
    }
    if (i < n4) {
        const f32x4 v = ((const f32x4*)a.x)[i];
        s += v[0]*v[0] + v[1]*v[1] + v[2]*v[2] + v[3]*v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) { const float v = a.x[(n4 << 2) + threadIdx.x]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.workspace[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* ws, int n, float* out, int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += ws[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { const float t = red[0] + red[1] + red[2] + red[3]; out[0] = accumulate ? out[0] + t : t; }
}

__global__ void center_update_kernel(gv_center_update_args a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < a.K) a.center[k] = a.center[k] * a.momentum + a.center_sum[k] * a.inv_rows * (1.0f - a.momentum);
}

// one thread per sample, C <= 64.  loss = mean_b[(1-s) * nll + s * smooth] on
// logp = log_softmax(softmax(logits)); analytic gradient through both softmaxes.
__global__ void softmax_lsce_kernel(gv_softmax_lsce_args a) {
    __shared__ float red[256];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f;
    if (b < a.B) {
        const int C = a.C;
        const float* z = a.logits + (long)b * C;
        float p[64], q[64], w[64];
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) { p[c] = expf(z[c] - mx); s += p[c]; }
        for (int c = 0; c < C; ++c) p[c] /= s;
        // second softmax over p (values in [0,1], no max shift needed but keep it)
        float s2 = 0.f;
        for (int c = 0; c < C; ++c) { q[c] = expf(p[c]); s2 += q[c]; }
        const float lse2 = logf(s2);
        for (int c = 0; c < C; ++c) q[c] /= s2;
        const int t = (int)a.target[b];
        const float sm = a.smoothing;
        float smooth = 0.f;
        for (int c = 0; c < C; ++c) smooth += -(p[c] - lse2);
        smooth /= C;
        const float nll = -(p[t] - lse2);
        l = (1.0f - sm) * nll + sm * smooth;
        // dl/dp_c = q_c - [(1-s) 1{c=t} + s/C]
        float dot = 0.f;
        for (int c = 0; c < C; ++c) { w[c] = q[c] - ((c == t ? 1.0f - sm : 0.f) + sm / C); dot += w[c] * p[c]; }
        const float gs = (a.loss_scale ? *a.loss_scale : 1.0f) / a.B;
        for (int c = 0; c < C; ++c) a.dlogits[(long)b * C + c] = p[c] * (w[c] - dot) * gs;
        if (a.prob) for (int c = 0; c < C; ++c) a.prob[(long)b * C + c] = p[c];
    }
    red[threadIdx.x] = l;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) atomicAdd(a.loss, red[0] / a.B);
}

}  // namespace

extern "C" int gv_cls_rows(const gv_cls_rows_args* a, void* stream) {
    GV_REQUIRE(a && a->x && a->cls && a->pos, GV_E_NULL, "gv_cls_rows: null pointer");
    GV_REQUIRE(a->n_img > 0 && a->N > 0 && a->D > 0, GV_E_SHAPE, "gv_cls_rows: bad shape");
    hipLaunchKernelGGL(cls_rows_kernel, dim3(a->n_img), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_cls_rows");
    return GV_OK;
}

template <typename PT> static int tokens_bwd_launch(const gv_tokens_bwd_args* a, void* stream) {
    GV_REQUIRE(a && a->g && a->gpatch && a->dpos, GV_E_NULL, "gv_tokens_bwd: null pointer");
    GV_REQUIRE(a->n_img > 0 && a->N > 1 && a->D > 0, GV_E_SHAPE, "gv_tokens_bwd: bad shape");
    if (!a->accumulate) {
        hipError_t e = hipMemsetAsync(a->dpos, 0, (size_t)a->N * a->D * sizeof(float), (hipStream_t)stream);
        if (e == hipSuccess && a->dcls) e = hipMemsetAsync(a->dcls, 0, (size_t)a->D * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) { gv_set_error("gv_tokens_bwd: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    }
    hipLaunchKernelGGL(tokens_bwd_kernel<PT>, dim3(a->N, (a->D + 255) / 256, (a->n_img + TOK_IMG_PER_CHUNK - 1) / TOK_IMG_PER_CHUNK), dim3(256), 0,
                       (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_tokens_bwd");
    return GV_OK;
}
extern "C" int gv_tokens_bwd(const gv_tokens_bwd_args* a, void* stream) { return tokens_bwd_launch<bf16>(a, stream); }
extern "C" int gv_tokens_bwd_f32(const gv_tokens_bwd_args* a, void* stream) { return tokens_bwd_launch<float>(a, stream); }

extern "C" int gv_small_matmul(const gv_small_matmul_args* a, void* stream) {
    GV_REQUIRE(a && a->A && a->B && a->C, GV_E_NULL, "gv_small_matmul: null pointer");
    GV_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0 && a->M < 65536, GV_E_SHAPE, "gv_small_matmul: bad shape");
    const dim3 grid((a->N + 127) / 128, a->M);
    hipStream_t s = (hipStream_t)stream;
    if (a->a_is_bf16 && a->b_is_bf16) hipLaunchKernelGGL((small_matmul_kernel<true, true>), grid, dim3(128), 0, s, *a);
    else if (a->a_is_bf16) hipLaunchKernelGGL((small_matmul_kernel<true, false>), grid, dim3(128), 0, s, *a);
    else if (a->b_is_bf16) hipLaunchKernelGGL((small_matmul_kernel<false, true>), grid, dim3(128), 0, s, *a);
    else hipLaunchKernelGGL((small_matmul_kernel<false, false>), grid, dim3(128), 0, s, *a);
    GV_LAUNCH_CHECK("gv_small_matmul");
    return GV_OK;
}

#define GV_ROW_LAUNCH(kern, a, name)                                                             \
    GV_REQUIRE((a)->rows > 0 && (a)->C > 0 && (a)->C % 4 == 0, GV_E_SHAPE, name ": C must be a positive multiple of 4"); \
    hipLaunchKernelGGL(kern, dim3(((a)->rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, *(a)); \
    GV_LAUNCH_CHECK(name);                                                                       \
    return GV_OK;

extern "C" int gv_l2norm_fwd(const gv_l2norm_fwd_args* a, void* stream) {
    GV_REQUIRE(a && a->x && a->y && a->inv_norm, GV_E_NULL, "gv_l2norm_fwd: null pointer");
    GV_ROW_LAUNCH(l2norm_fwd_kernel<bf16>, a, "gv_l2norm_fwd")
}
extern "C" int gv_l2norm_bwd(const gv_l2norm_bwd_args* a, void* stream) {
    GV_REQUIRE(a && a->dy && a->y && a->inv_norm && a->dx, GV_E_NULL, "gv_l2norm_bwd: null pointer");
    GV_ROW_LAUNCH(l2norm_bwd_kernel<bf16>, a, "gv_l2norm_bwd")
}
extern "C" int gv_weightnorm_fwd(const gv_weightnorm_fwd_args* a, void* stream) {
    GV_REQUIRE(a && a->v && a->g && a->w, GV_E_NULL, "gv_weightnorm_fwd: null pointer");
    if (a->C == 256 && a->rows > 0) {
        hipLaunchKernelGGL(weightnorm_fwd256_kernel<bf16>, dim3((a->rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, *a);
        GV_LAUNCH_CHECK("gv_weightnorm_fwd");
        return GV_OK;
    }
    GV_ROW_LAUNCH(weightnorm_fwd_kernel<bf16>, a, "gv_weightnorm_fwd")
}
extern "C" int gv_l2norm_fwd_f32(const gv_l2norm_fwd_args* a, void* stream) {
    GV_REQUIRE(a && a->x && a->y && a->inv_norm, GV_E_NULL, "gv_l2norm_fwd: null pointer");
    GV_ROW_LAUNCH(l2norm_fwd_kernel<float>, a, "gv_l2norm_fwd_f32")
}
extern "C" int gv_l2norm_bwd_f32(const gv_l2norm_bwd_args* a, void* stream) {
    GV_REQUIRE(a && a->dy && a->y && a->inv_norm && a->dx, GV_E_NULL, "gv_l2norm_bwd: null pointer");
    GV_ROW_LAUNCH(l2norm_bwd_kernel<float>, a, "gv_l2norm_bwd_f32")
}
extern "C" int gv_weightnorm_fwd_f32(const gv_weightnorm_fwd_args* a, void* stream) {
    GV_REQUIRE(a && a->v && a->g && a->w, GV_E_NULL, "gv_weightnorm_fwd: null pointer");
    if (a->C == 256 && a->rows > 0) {
        hipLaunchKernelGGL(weightnorm_fwd256_kernel<float>, dim3((a->rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, *a);
        GV_LAUNCH_CHECK("gv_weightnorm_fwd_f32");
        return GV_OK;
    }
    GV_ROW_LAUNCH(weightnorm_fwd_kernel<float>, a, "gv_weightnorm_fwd_f32")
}
extern "C" int gv_weightnorm_bwd(const gv_weightnorm_bwd_args* a, void* stream) {
    GV_REQUIRE(a && a->dw && a->v && a->g && a->dv, GV_E_NULL, "gv_weightnorm_bwd: null pointer");
    if (a->C == 256 && a->rows > 0) {
        hipLaunchKernelGGL(weightnorm_bwd256_kernel, dim3((a->rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, *a);
        GV_LAUNCH_CHECK("gv_weightnorm_bwd");
        return GV_OK;
    }
    GV_ROW_LAUNCH(weightnorm_bwd_kernel, a, "gv_weightnorm_bwd")
}

extern "C" int gv_store_f32(const gv_store_f32_args* a, void* stream) {
    GV_REQUIRE(a && a->dst, GV_E_NULL, "gv_store_f32: null pointer");
    GV_REQUIRE(a->n > 0 && a->n <= 16, GV_E_SHAPE, "gv_store_f32: need 0 < n <= 16");
    hipLaunchKernelGGL(store_f32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_store_f32");
    return GV_OK;
}

namespace {
__global__ __launch_bounds__(256) void expand_rows_kernel(gv_expand_rows_args a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long)a.n_rep * a.T) { const long r = i / a.T; a.rows[i] = a.per_img[r * a.n_img + a.row_img[i - r * a.T]]; }
}
}  // namespace

extern "C" int gv_expand_rows(const gv_expand_rows_args* a, void* stream) {
    GV_REQUIRE(a && a->per_img && a->row_img && a->rows, GV_E_NULL, "gv_expand_rows: null pointer");
    GV_REQUIRE(a->n_rep > 0 && a->n_img > 0 && a->T > 0, GV_E_SHAPE, "gv_expand_rows: bad shape");
    const long n = (long)a->n_rep * a->T;
    hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_expand_rows");
    return GV_OK;
}

extern "C" int gv_gather_cls(const gv_gather_cls_args* a, void* stream) {
    GV_REQUIRE(a && a->x && a->y, GV_E_NULL, "gv_gather_cls: null pointer");
    GV_REQUIRE(a->n_img > 0 && a->N > 0 && a->D > 0, GV_E_SHAPE, "gv_gather_cls: bad shape");
    hipLaunchKernelGGL(gather_cls_kernel, dim3(a->n_img), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_gather_cls");
    return GV_OK;
}

extern "C" int gv_cast_bf16(const gv_cast_bf16_args* a, void* stream) {
    GV_REQUIRE(a && a->src && a->dst, GV_E_NULL, "gv_cast_bf16: null pointer");
    GV_REQUIRE(a->n > 0, GV_E_SHAPE, "gv_cast_bf16: n must be > 0");
    GV_REQUIRE(gv_aligned(a->src, 16) && gv_aligned(a->dst, 8), GV_E_ALIGN, "gv_cast_bf16: misaligned");
    long blocks = (a->n / 4 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_cast_bf16");
    return GV_OK;
}

// ---- dropout (include/gipvit.h gv_dropout / gv_dropout_add): counter-based keep masks
namespace {
__device__ __forceinline__ bool drop_keep(uint32_t seed, uint32_t idx, uint32_t thr) {
    uint32_t h = seed + 0x9E3779B9u * (idx + 1u);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h >= thr;
}
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(gv_dropout_args a) {
    T* x = (T*)a.x;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256)
        x[i] = drop_keep(a.seed, (uint32_t)i, a.threshold) ? (T)((float)x[i] * a.scale) : (T)0.0f;
}
__global__ __launch_bounds__(256) void dropout_add_kernel(gv_dropout_add_args a) {
    const long n = (long)a.rows * a.cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int m = (int)(i / a.cols);
        const float v = drop_keep(a.seed, (uint32_t)i, a.threshold) ? a.t[i] * a.scale : 0.0f;
        a.out[i] = a.resid[i] + (a.row_scale ? a.row_scale[m] : 1.0f) * v;
    }
}
}  // namespace

extern "C" int gv_dropout(const gv_dropout_args* a, void* stream) {
    GV_REQUIRE(a && a->x, GV_E_NULL, "gv_dropout: null pointer");
    GV_REQUIRE(a->n > 0 && a->n < (1ll << 32), GV_E_SHAPE, "gv_dropout: need 0 < n < 2^32 elements per site (got %ld)", (long)a->n);
    long blocks = (a->n + 255) / 256; if (blocks > 8192) blocks = 8192;
    if (a->x_is_f32) hipLaunchKernelGGL(dropout_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(dropout_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_dropout");
    return GV_OK;
}

extern "C" int gv_dropout_add(const gv_dropout_add_args* a, void* stream) {
    GV_REQUIRE(a && a->t && a->resid && a->out, GV_E_NULL, "gv_dropout_add: null pointer");
    GV_REQUIRE(a->rows > 0 && a->cols > 0 && (long)a->rows * a->cols < (1ll << 32), GV_E_SHAPE, "gv_dropout_add: need 0 < rows * cols < 2^32");
    long blocks = ((long)a->rows * a->cols + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dropout_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_dropout_add");
    return GV_OK;
}

extern "C" int gv_sumsq(const gv_sumsq_args* a, void* stream) {
    GV_REQUIRE(a && a->x && a->workspace && a->out, GV_E_NULL, "gv_sumsq: null pointer");
    GV_REQUIRE(a->n > 0 && gv_aligned(a->x, 16), GV_E_ALIGN, "gv_sumsq: x must be 16-byte aligned, n > 0");
    long blocks = (a->n / 4 + 255) / 256; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_sumsq");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a->workspace, (int)blocks, a->out, a->accumulate);
    GV_LAUNCH_CHECK("gv_sumsq(final)");
    return GV_OK;
}

extern "C" int gv_center_update(const gv_center_update_args* a, void* stream) {
    GV_REQUIRE(a && a->center && a->center_sum, GV_E_NULL, "gv_center_update: null pointer");
    GV_REQUIRE(a->K > 0, GV_E_SHAPE, "gv_center_update: K must be > 0");
    hipLaunchKernelGGL(center_update_kernel, dim3((a->K + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_center_update");
    return GV_OK;
}

extern "C" int gv_softmax_lsce(const gv_softmax_lsce_args* a, void* stream) {
    GV_REQUIRE(a && a->logits && a->target && a->loss && a->dlogits, GV_E_NULL, "gv_softmax_lsce: null pointer");
    GV_REQUIRE(a->B > 0 && a->C > 0 && a->C <= 64, GV_E_SHAPE, "gv_softmax_lsce: need 0 < C <= 64");
    hipError_t e = hipMemsetAsync(a->loss, 0, sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) GV_FAIL((int)e, "gv_softmax_lsce: memset failed");
    hipLaunchKernelGGL(softmax_lsce_kernel, dim3((a->B + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_softmax_lsce");
    return GV_OK;
}
