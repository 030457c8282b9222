// Fused AdamW + teacher EMA + bf16 weight refresh over a flat parameter arena.
// Replaces optimizer.step() (reference train.py:1078; torch.optim.AdamW semantics) and
// the EMA update (train.py:1080-1081 ModelEmaV2 / DINO teacher) with one
// HBM-bound pass: per element 20 B read (g, p, m, v, teacher) + 20 B written
// (p, m, v, teacher, two bf16 copies).
#include "gv_common.h"

namespace {

__global__ __launch_bounds__(256) void adamw_ema_kernel(gv_adamw_ema_args a) {
    if (a.hyper) {
        a.lr = a.hyper[GV_HYP_LR]; a.weight_decay *= a.hyper[GV_HYP_WD];   // by-value wd is a 0/1 multiplier here
        a.bias_corr1 = a.hyper[GV_HYP_BC1]; a.bias_corr2 = a.hyper[GV_HYP_BC2];
        a.teacher_momentum = a.hyper[GV_HYP_TEACHER_MOM]; a.grad_scale = a.hyper[GV_HYP_GRAD_SCALE];
    }
    float gscale = a.grad_scale;
    if (a.loss_scale) {                                    // fp16 loss scaling: unscale; a non-finite gradient skips the step
        gscale /= a.loss_scale[0];
        if (!isfinite(*a.gnorm_sq)) a.mode = 3;
        // a skipped step is no optimizer step: Adam's bias corrections count the APPLIED steps (state[3]), not the host's calls
        const float t = a.loss_scale[3] + 1.0f;
        a.bias_corr1 = 1.0f - powf(a.beta1, t); a.bias_corr2 = 1.0f - powf(a.beta2, t);
    }
    if (a.clip_norm > 0.f && a.mode != 3) {
        const float nrm = sqrtf(*a.gnorm_sq) * fabsf(gscale);
        const float c = a.clip_norm / (nrm + 1e-6f);
        if (c < 1.0f) gscale *= c;
    }
    const float decay = 1.0f - a.lr * a.weight_decay;
    const float step = a.lr / a.bias_corr1;
    const float inv_sqrt_bc2 = 1.0f / sqrtf(a.bias_corr2);
    const float om = 1.0f - a.teacher_momentum;
    const long n4 = a.n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 g = a.mode == 3 ? f32x4{0.f, 0.f, 0.f, 0.f} : ((const f32x4*)a.grad)[i] * gscale;
        if (a.clip_value > 0.f) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = fminf(fmaxf(g[j], -a.clip_value), a.clip_value);
        }
        f32x4 p = ((f32x4*)a.p)[i], m = ((f32x4*)a.m)[i], v = ((f32x4*)a.v)[i];
        if (a.mode == 3) {
            // frozen range (no gradient: the optimizer skips the parameter); only the EMA below runs
        } else if (a.mode == 0) {
            p *= decay;
        } else {
            g += p * a.weight_decay;                        // L2 decay rides on the gradient
        }
        if (a.mode == 3) {
        } else if (a.mode == 2) {                                  // SGD, Nesterov momentum (torch.optim.SGD semantics)
            m = m * a.beta1 + g;
            p -= (g + m * a.beta1) * a.lr;
        } else {
            m = m * a.beta1 + g * (1.0f - a.beta1);
            v = v * a.beta2 + g * g * (1.0f - a.beta2);
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] -= step * m[j] / (sqrtf(v[j]) * inv_sqrt_bc2 + a.eps);
        }
        if (a.mode != 3) { ((f32x4*)a.p)[i] = p; ((f32x4*)a.m)[i] = m; ((f32x4*)a.v)[i] = v; }
        if (a.p_bf16) ((bf16x4*)a.p_bf16)[i] = bf16x4{(bf16)p[0], (bf16)p[1], (bf16)p[2], (bf16)p[3]};
        if (a.teacher) {
            f32x4 t = ((f32x4*)a.teacher)[i] * a.teacher_momentum + p * om;
            ((f32x4*)a.teacher)[i] = t;
            if (a.teacher_bf16) ((bf16x4*)a.teacher_bf16)[i] = bf16x4{(bf16)t[0], (bf16)t[1], (bf16)t[2], (bf16)t[3]};
        }
    }
}

__global__ void loss_scale_update_kernel(gv_loss_scale_update_args a) {
    float S = a.state[0], good = a.state[1];
    if (!isfinite(*a.gnorm_sq)) { S *= a.backoff_factor; good = 0.f; a.state[2] += 1.f; }
    else {
        a.state[3] += 1.f;
        if (++good >= (float)a.growth_interval) { S *= a.growth_factor; good = 0.f; }
    }
    a.state[0] = S; a.state[1] = good;
}

// adaptive gradient clipping: one wave per unit (include/gipvit.h gv_agc)
__global__ __launch_bounds__(256) void agc_kernel(gv_agc_args a) {
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= a.n_units) return;
    const int off = a.units[2 * u], len = a.units[2 * u + 1];
    const float* p = a.p + off; float* g = a.grad + off;
    float sp = 0.f, sg = 0.f;
    for (int i = lane; i < len; i += 64) { const float pv = p[i], gv = g[i] * a.grad_scale; sp += pv * pv; sg += gv * gv; }
    sp = wave_sum(sp); sg = wave_sum(sg);
    const float max_norm = fmaxf(sqrtf(sp), a.eps) * a.clip_factor, gn = sqrtf(sg);
    if (gn < max_norm) return;
    const float c = max_norm / fmaxf(gn, 1e-6f);
    for (int i = lane; i < len; i += 64) g[i] *= c;
}

// LAMB, both phases (include/gipvit.h gv_lamb).  One workgroup per table entry: a slice of ONE tensor.
__global__ __launch_bounds__(256) void lamb_kernel(gv_lamb_args a) {
    const int tensor = a.blocks[blockIdx.x * 3], lo = a.blocks[blockIdx.x * 3 + 1], hi = a.blocks[blockIdx.x * 3 + 2];
    float gscale = a.grad_scale;
    {
        float nrm = sqrtf(*a.gnorm_sq) * fabsf(a.grad_scale);
        if (a.clip_norm > 0.f) { const float c = a.clip_norm / (nrm + 1e-6f); if (c < 1.0f) { gscale *= c; nrm *= c; } }
        if (a.max_grad_norm > 0.f && nrm > a.max_grad_norm) gscale *= a.max_grad_norm / nrm;
    }
    const float inv_bc1 = 1.0f / a.bias_corr1, inv_sqrt_bc2 = 1.0f / sqrtf(a.bias_corr2);
    float trust = 1.0f;
    if (a.phase == 1 && a.weight_decay != 0.f) {
        const float wn = sqrtf(a.stats[2 * tensor]), un = sqrtf(a.stats[2 * tensor + 1]);
        trust = (wn > 0.f && un > 0.f) ? wn / un : 1.0f;
    }
    const float om = 1.0f - a.teacher_momentum;
    float sp = 0.f, su = 0.f;
    for (long i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += 256) {
        f32x4 p = ((f32x4*)a.p)[i], m = ((f32x4*)a.m)[i], v = ((f32x4*)a.v)[i];
        if (a.phase == 0) {
            const f32x4 g = ((const f32x4*)a.grad)[i] * gscale;
            m = m * a.beta1 + g * (1.0f - a.beta1);
            v = v * a.beta2 + g * g * (1.0f - a.beta2);
            ((f32x4*)a.m)[i] = m; ((f32x4*)a.v)[i] = v;
        }
        f32x4 u;
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = (m[j] * inv_bc1) / (sqrtf(v[j]) * inv_sqrt_bc2 + a.eps) + a.weight_decay * p[j];
        if (a.phase == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { sp += p[j] * p[j]; su += u[j] * u[j]; }
        } else {
            p -= u * (a.lr * trust);
            ((f32x4*)a.p)[i] = p;
            if (a.p_bf16) ((bf16x4*)a.p_bf16)[i] = bf16x4{(bf16)p[0], (bf16)p[1], (bf16)p[2], (bf16)p[3]};
            if (a.teacher) {
                f32x4 t = ((f32x4*)a.teacher)[i] * a.teacher_momentum + p * om;
                ((f32x4*)a.teacher)[i] = t;
                if (a.teacher_bf16) ((bf16x4*)a.teacher_bf16)[i] = bf16x4{(bf16)t[0], (bf16)t[1], (bf16)t[2], (bf16)t[3]};
            }
        }
    }
    if (a.phase == 0) {
        sp = wave_sum(sp); su = wave_sum(su);
        __shared__ float red[8];
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = sp; red[4 + (threadIdx.x >> 6)] = su; }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(a.stats + 2 * tensor, red[0] + red[1] + red[2] + red[3]);
            atomicAdd(a.stats + 2 * tensor + 1, red[4] + red[5] + red[6] + red[7]);
        }
    }
}

}  // namespace

extern "C" int gv_loss_scale_update(const gv_loss_scale_update_args* a, void* stream) {
    GV_REQUIRE(a && a->state && a->gnorm_sq, GV_E_NULL, "gv_loss_scale_update: null pointer");
    GV_REQUIRE(a->growth_factor >= 1.f && a->backoff_factor > 0.f && a->backoff_factor <= 1.f && a->growth_interval > 0, GV_E_SHAPE,
               "gv_loss_scale_update: need growth_factor >= 1, 0 < backoff_factor <= 1, growth_interval > 0");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_loss_scale_update");
    return GV_OK;
}

extern "C" int gv_agc(const gv_agc_args* a, void* stream) {
    GV_REQUIRE(a && a->p && a->grad && a->units, GV_E_NULL, "gv_agc: null pointer");
    GV_REQUIRE(a->n_units > 0 && a->clip_factor > 0.f && a->eps > 0.f, GV_E_SHAPE, "gv_agc: need n_units > 0, clip_factor > 0, eps > 0");
    hipLaunchKernelGGL(agc_kernel, dim3((a->n_units + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_agc");
    return GV_OK;
}

extern "C" int gv_lamb(const gv_lamb_args* a, void* stream) {
    GV_REQUIRE(a && a->p && a->grad && a->m && a->v && a->blocks && a->stats && a->gnorm_sq, GV_E_NULL, "gv_lamb: null pointer");
    GV_REQUIRE(a->phase == 0 || a->phase == 1, GV_E_UNSUPPORTED, "gv_lamb: phase must be 0 (moments + norms) or 1 (apply)");
    GV_REQUIRE(a->n_blocks > 0, GV_E_SHAPE, "gv_lamb: empty block table");
    GV_REQUIRE(gv_aligned(a->p, 16) && gv_aligned(a->grad, 16) && gv_aligned(a->m, 16) && gv_aligned(a->v, 16), GV_E_ALIGN,
               "gv_lamb: buffers must be 16-byte aligned");
    GV_REQUIRE(a->bias_corr1 > 0.f && a->bias_corr2 > 0.f, GV_E_SHAPE, "gv_lamb: bias corrections must be > 0");
    hipLaunchKernelGGL(lamb_kernel, dim3((unsigned)a->n_blocks), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_lamb");
    return GV_OK;
}

extern "C" int gv_adamw_ema(const gv_adamw_ema_args* a, void* stream) {
    GV_REQUIRE(a && a->p && a->grad && a->m && a->v, GV_E_NULL, "gv_adamw_ema: null pointer");
    GV_REQUIRE(a->mode >= 0 && a->mode <= 3, GV_E_UNSUPPORTED, "gv_adamw_ema: mode must be 0 (AdamW), 1 (Adam+L2), 2 (SGD Nesterov) or 3 (frozen: EMA only)");
    GV_REQUIRE(a->n > 0 && a->n % 4 == 0, GV_E_SHAPE, "gv_adamw_ema: n=%ld must be a positive multiple of 4 (pad the arena)", (long)a->n);
    GV_REQUIRE(gv_aligned(a->p, 16) && gv_aligned(a->grad, 16) && gv_aligned(a->m, 16) && gv_aligned(a->v, 16), GV_E_ALIGN,
               "gv_adamw_ema: buffers must be 16-byte aligned");
    if (a->clip_norm > 0.f) GV_REQUIRE(a->gnorm_sq, GV_E_NULL, "gv_adamw_ema: clip_norm needs gnorm_sq");
    if (a->loss_scale) GV_REQUIRE(a->gnorm_sq, GV_E_NULL, "gv_adamw_ema: loss_scale needs gnorm_sq (the finite check)");
    GV_REQUIRE(!(a->clip_norm > 0.f && a->clip_value > 0.f), GV_E_UNSUPPORTED, "gv_adamw_ema: clip_norm and clip_value exclude each other (--clip-mode norm | value)");
    if (!a->hyper) GV_REQUIRE(a->bias_corr1 > 0.f && a->bias_corr2 > 0.f, GV_E_SHAPE, "gv_adamw_ema: bias corrections must be > 0");
    long blocks = (a->n / 4 + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    GV_LAUNCH_CHECK("gv_adamw_ema");
    return GV_OK;
}
