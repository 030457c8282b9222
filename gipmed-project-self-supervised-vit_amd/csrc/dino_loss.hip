// Fused DINO head loss: student log-softmax(s / tau_s), teacher softmax((t - c) /
// tau_t), cross-entropy over the (teacher global crop iq, student crop v != iq) pairs,
// its gradient w.r.t. the student logits and the teacher center partial sum.
// The reference has no DINO loop; spec = DINO paper Alg. 1 (rows D2/D3).
//
// HBM-bound: pass 1 (row_stats) reads every logit once, pass 2 reads them once more and
// writes the bf16 gradient: (V+G)*B*K*4*2 + V*B*K*2 bytes per call.
//   loss = 1/(n_pairs B) sum_{b, iq, v != iq} -sum_k t_iq[b,k] logp_v[b,k]
//   d loss / d s_v[b,k] = (n_v p_v[b,k] - sum_{iq != v} t_iq[b,k]) / (n_pairs B tau_s)
#include "gv_common.h"

namespace {

constexpr int MAXV = 16, MAXG = 4;

// DT: element type of the student-logit gradient -- bf16 on the training path (fast exp / log: their ~1e-6 relative
// error is far below the bf16 rounding of the result); float in the fp32 operand mode, which uses expf / logf
template <typename DT> __device__ __forceinline__ float ex(float x) { if constexpr (sizeof(DT) == 4) return expf(x); else return __expf(x); }
template <typename DT> __device__ __forceinline__ float lg(float x) { if constexpr (sizeof(DT) == 4) return logf(x); else return __logf(x); }

// one block per logits row: running max / sum-exp of the temperature-scaled row.  A thread takes RS_U 16-B pieces per trip, both
// requested before the first is used, folds them into one maximum and rescales its running sum once per trip.  Measured at the
// headline shape (768 rows x 256 KB, tools/loss_bench.py, whole call): one piece per trip 101 us, two 93, four 102, eight 119;
// 512- / 1024-thread blocks 92 - 98
template <typename DT, int NT, int RS_U>
__global__ __launch_bounds__(NT) void row_stats_kernel(gv_dino_loss_args a) {
    constexpr int NWV = NT / 64;
    __shared__ float red_m[NWV], red_s[NWV];
    const int row = blockIdx.x;
    const int ns = a.V * a.B;
    const bool teacher = row >= ns;
    if (a.hyper) { a.teacher_temp = a.hyper[GV_HYP_TEACHER_TEMP]; a.student_temp = a.hyper[GV_HYP_STUDENT_TEMP]; }
    const float* x = teacher ? a.teacher + (long)(row - ns) * a.K : a.student + (long)row * a.K;
    const float inv_t = 1.0f / (teacher ? a.teacher_temp : a.student_temp);
    float m = -INFINITY, s = 0.f;
    for (int k0 = threadIdx.x * 4; k0 < a.K; k0 += NT * 4 * RS_U) {
        f32x4 v[RS_U];
#pragma unroll
        for (int u = 0; u < RS_U; ++u) {
            const int k = k0 + u * NT * 4;
            v[u] = k < a.K ? *(const f32x4*)(x + k) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
        if (teacher) {
#pragma unroll
            for (int u = 0; u < RS_U; ++u) {
                const int k = k0 + u * NT * 4;
                if (k < a.K) v[u] -= *(const f32x4*)(a.center + k);
            }
        }
        float lm = -INFINITY;
#pragma unroll
        for (int u = 0; u < RS_U; ++u) {
            v[u] *= inv_t;
            lm = fmaxf(lm, fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3])));
        }
        if (lm > m) { s *= ex<DT>(m - lm); m = lm; }       // (piece 0 of a trip is always inside the row: lm is finite)
#pragma unroll
        for (int u = 0; u < RS_U; ++u) s += ex<DT>(v[u][0] - m) + ex<DT>(v[u][1] - m) + ex<DT>(v[u][2] - m) + ex<DT>(v[u][3] - m);
    }
    const float wm = wave_max(m);
    s = (m == -INFINITY) ? 0.f : s * ex<DT>(m - wm);   // idle lanes when K < 4 NT
    s = wave_sum(s);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red_m[wave] = wm; red_s[wave] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = red_m[0];
        for (int w = 1; w < NWV; ++w) M = fmaxf(M, red_m[w]);
        float S = 0.f;
        for (int w = 0; w < NWV; ++w) S += red_s[w] * ex<DT>(red_m[w] - M);
        a.workspace[2 * row] = M;
        a.workspace[2 * row + 1] = lg<DT>(S);
    }
}

// grid (K / 1024, bsplit); thread = FOUR consecutive classes k (16-B loads, 8-B / 16-B gradient stores -- one class per thread
// made every access 4 or 2 bytes per lane), loops over its slice of the batch
template <typename DT>
__global__ __launch_bounds__(256) void loss_grad_kernel(gv_dino_loss_args a, int b_per, float coef, float inv_pairs_b) {
    __shared__ float red[4];
    if (a.hyper) {   // coef was built with the by-value student temperature
        coef *= a.student_temp / a.hyper[GV_HYP_STUDENT_TEMP];
        a.teacher_temp = a.hyper[GV_HYP_TEACHER_TEMP]; a.student_temp = a.hyper[GV_HYP_STUDENT_TEMP];
    }
    if (a.loss_scale) coef *= *a.loss_scale;
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int b0 = blockIdx.y * b_per, b1 = min(a.B, b0 + b_per);
    const int V = a.V, G = a.G, B = a.B, K = a.K;
    const int ns = V * B;
    const float inv_ts = 1.0f / a.student_temp, inv_tt = 1.0f / a.teacher_temp;
    float loss = 0.f;
    if (k < K) {
        f32x4 csum = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 c = *(const f32x4*)(a.center + k);
        for (int b = b0; b < b1; ++b) {
            f32x4 t[MAXG], tsum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int iq = 0; iq < MAXG; ++iq) {
                t[iq] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (iq < G) {
                    const int row = ns + iq * B + b;
                    const f32x4 raw = *(const f32x4*)(a.teacher + (long)(iq * B + b) * K + k);
                    const float rm = a.workspace[2 * row], rl = a.workspace[2 * row + 1];
                    csum += raw;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[iq][e] = ex<DT>((raw[e] - c[e]) * inv_tt - rm - rl);
                    tsum += t[iq];
                }
            }
#pragma unroll 2
            for (int v = 0; v < V; ++v) {
                const int row = v * B + b;
                const f32x4 sv = *(const f32x4*)(a.student + (long)row * K + k);
                const float rm = a.workspace[2 * row], rl = a.workspace[2 * row + 1];
                f32x4 ts = tsum;
                float nv = (float)G;
                if (v < G) { ts -= (v == 0 ? t[0] : v == 1 ? t[1] : v == 2 ? t[2] : t[3]); nv -= 1.f; }
                f32x4 g;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float logp = sv[e] * inv_ts - rm - rl;
                    loss -= ts[e] * logp;
                    g[e] = coef * (nv * ex<DT>(logp) - ts[e]);
                }
                DT* dst = (DT*)a.dstudent + (long)row * K + k;
                if constexpr (sizeof(DT) == 4) *(f32x4*)dst = g;
                else *(bf16x4*)dst = bf16x4{(bf16)g[0], (bf16)g[1], (bf16)g[2], (bf16)g[3]};
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(a.center_sum + k + e, csum[e]);
    }
    loss = wave_sum(loss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.loss, (red[0] + red[1] + red[2] + red[3]) * inv_pairs_b);
}

template <typename DT> int dino_loss_launch(const gv_dino_loss_args* a, void* stream) {
    GV_REQUIRE(a && a->student && a->teacher && a->center && a->dstudent && a->loss && a->center_sum && a->workspace,
               GV_E_NULL, "gv_dino_loss: null pointer");
    GV_REQUIRE(a->B > 0 && a->V >= 2 && a->V <= MAXV && a->G >= 1 && a->G <= MAXG && a->G <= a->V, GV_E_SHAPE,
               "gv_dino_loss: need 1 <= G <= %d, G <= V <= %d (got V=%d G=%d)", MAXG, MAXV, a->V, a->G);
    GV_REQUIRE(a->K > 0 && a->K % 4 == 0, GV_E_SHAPE, "gv_dino_loss: K must be a positive multiple of 4");
    GV_REQUIRE(a->student_temp > 0.f && a->teacher_temp > 0.f, GV_E_SHAPE, "gv_dino_loss: temperatures must be > 0");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(a->loss, 0, sizeof(float), s);
    if (e == hipSuccess) e = hipMemsetAsync(a->center_sum, 0, sizeof(float) * a->K, s);
    if (e != hipSuccess) GV_FAIL((int)e, "gv_dino_loss: memset failed: %s", hipGetErrorString(e));
    const int rows = (a->V + a->G) * a->B;
    hipLaunchKernelGGL((row_stats_kernel<DT, 256, 2>), dim3(rows), dim3(256), 0, s, *a);
    GV_LAUNCH_CHECK("gv_dino_loss(row_stats)");
    const int kblocks = (a->K + 1023) / 1024;
    int bsplit = (1024 + kblocks - 1) / kblocks;
    if (bsplit > a->B) bsplit = a->B;
    if (bsplit < 1) bsplit = 1;
    const int b_per = (a->B + bsplit - 1) / bsplit;
    bsplit = (a->B + b_per - 1) / b_per;
    const int n_pairs = a->G * (a->V - 1);
    const float inv_pairs_b = 1.0f / ((float)n_pairs * (float)a->B);
    const float gs = a->grad_scale == 0.f ? 1.f : a->grad_scale;
    const float coef = gs * inv_pairs_b / a->student_temp;
    hipLaunchKernelGGL(loss_grad_kernel<DT>, dim3(kblocks, bsplit), dim3(256), 0, s, *a, b_per, coef, inv_pairs_b);
    GV_LAUNCH_CHECK("gv_dino_loss(loss_grad)");
    return GV_OK;
}

}  // namespace

extern "C" int gv_dino_loss(const gv_dino_loss_args* a, void* stream) { return dino_loss_launch<bf16>(a, stream); }
extern "C" int gv_dino_loss_f32(const gv_dino_loss_args* a, void* stream) { return dino_loss_launch<float>(a, stream); }
