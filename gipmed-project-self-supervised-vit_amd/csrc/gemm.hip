// bf16 MFMA GEMM for gfx950 with fused epilogues -- gv_linear (include/gipvit.h).
//
// Replaces the ATen addmm / mm calls behind nn.Linear forward and backward in the
// reference's ViT blocks, patch embedding and DINOHead (vit.pyc@L98-104, L119-131,
// L167-170, L326-330; reference train.py:1045 forward, :1071 backward).
//
// The path's GEMMs are skinny: tens of thousands of token rows against K = 384..2048 (6..32
// k-steps per tile), or -- for the weight gradients -- a tiny output reduced over all tokens.
// Design (MI355X_MICROARCH / cdna_hip_programming section 5; geometry, ring depth and schedule
// were chosen with tools/gemm_lab.hip on MI355X, numbers in DESIGN.md):
//   * one 256-thread workgroup (2x2 waves, 64x64 per wave = 4x4 v_mfma_f32_16x16x32_bf16 per
//     32-deep k-step) per 128x128 output tile, 2 workgroups per CU: the partner workgroup's
//     k-loop covers this one's prologue / epilogue.  BK = 64, 2-stage LDS ring (64 KiB).
//   * the ring is filled by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, issued
//     from inline asm so hipcc's waitcnt pass does not drain it) one k-step ahead of the MFMAs,
//     behind a hand-counted s_waitcnt vmcnt(N) and ONE raw s_barrier per k-step.
//   * the LDS image is lane-linear, so bank swizzles are applied to the per-lane SOURCE address
//     and again on the read (rule 21):
//       "natural" operand (reduction index contiguous, x[M,K], W[N,K]): 128-B rows, 16-B chunk
//        index XOR (row & 7) -> conflict-free ds_read_b128;
//       "transposed" operand (reduction index strided: dY[tokens,N] for dW, W[N,K] for dX):
//        [64 k][128 cols] image, 32-B chunk index XOR f(k), fragments by two
//        ds_read_b64_tr_b16 (hardware transpose) -- no transposed copies in HBM.
//   * MFMA roles are swapped (weight fragment as the A operand) so each lane ends up with 4
//     consecutive output columns; the accumulators then pass through a per-wave LDS image so
//     that bias / residual / saved-activation loads and all stores are whole 128..256-B row
//     segments.  Epilogue operands (residual, pos-embed, dGELU input, bias) are fetched BEFORE
//     the k-loop; the epilogue mask is a template parameter (a runtime mask cost a load + wait
//     per element).  GELU / GELU' are clamped odd polynomials (gv_common.h), no transcendentals.
//   * split-K (dW over tens of thousands of tokens, dX of the 65536-class head): partial tiles
//     go to a caller-provided slab with plain row stores and a reduce kernel adds them into C
//     (f32 atomics run at ~1.3 TB/s on this part; they remain the fallback without a workspace).
//     The bias gradient (column sums of dY) rides along as one extra MFMA per A fragment
//     against a ones fragment.
//   * workgroup -> tile map is XCD-aware: each XCD's workgroups walk a contiguous band of tiles.
#ifndef GV_GEMM_BM
#define GV_GEMM_BM 128
#define GV_GEMM_BN 128
#define GV_GEMM_BK 64
#define GV_GEMM_WM 2
#define GV_GEMM_WN 2
#define GV_GEMM_NSTAGE 2
#endif
#include "gemm_core.h"
#include "gemm_dw8.h"
#include "timing.h"
#include <type_traits>
#include <stdlib.h>
#include <mutex>
#include <vector>
#include <string.h>

namespace {

using namespace gvgemm;

// production geometry (chosen with tools/gemm_lab.hip on MI355X)
using PCfg = Cfg<GV_GEMM_BM, GV_GEMM_BN, GV_GEMM_BK, GV_GEMM_WM, GV_GEMM_WN, GV_GEMM_NSTAGE>;
constexpr int BM = PCfg::BM, BN = PCfg::BN, BK = PCfg::BK;
constexpr int WGS_PER_CU = (160 * 1024 / PCfg::LDS) < 2 ? 1 : ((160 * 1024 / PCfg::LDS) >= 3 ? 3 : 2);
constexpr int PERSISTENT_GRID = 256 * WGS_PER_CU;

template <bool TA, bool TB, typename OutT, bool ATOMIC, int EPI>
__global__ __launch_bounds__(PCfg::THREADS, PCfg::THREADS * WGS_PER_CU / 256) void gemm_kernel(const GemmP g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    gemm_body<PCfg, TA, TB, OutT, ATOMIC, EPI>(g, (GV_LDS char*)smem_raw);
}

// Few tiles, long reduction (the DINO head's 2048-wide layers and the CLS-only tail at 128 / 640 rows: 15 - 80 workgroups, 24 - 32
// K-steps each): with one workgroup per CU and a two-stage ring every K-step pays a whole L2 / HBM round trip, ~1 us -- 34 us for
// K = 2048 whatever M is.  The same body on a FOUR-stage ring (128 KB of LDS, three K-steps in flight): same tiles, same
// accumulation order, bit-identical results.
using DCfg = Cfg<GV_GEMM_BM, GV_GEMM_BN, GV_GEMM_BK, GV_GEMM_WM, GV_GEMM_WN, 4>;
static_assert(DCfg::LDS <= 160 * 1024, "deep ring");
template <bool TA, bool TB, typename OutT, bool ATOMIC, int EPI>
__global__ __launch_bounds__(DCfg::THREADS, 1) void gemm_deep_kernel(const GemmP g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    gemm_body<DCfg, TA, TB, OutT, ATOMIC, EPI>(g, (GV_LDS char*)smem_raw);
}

template <bool TA, bool TB, typename OutT, bool ATOMIC, int EPI>
const char* kernel_name(bool deep = false) {
    static char name[2][96] = {"", ""};
    if (!name[deep][0])
        snprintf(name[deep], sizeof(name[deep]), "gemm_%skernel<%s, %s, %s, %s, %d>", deep ? "deep_" : "", TA ? "true" : "false", TB ? "true" : "false",
                 sizeof(OutT) == 4 ? "float" : "bf16", ATOMIC ? "true" : "false", EPI);
    return name[deep];
}

template <bool TA, bool TB, typename OutT, bool ATOMIC, int EPI = -1>
int launch(const GemmP& p, hipStream_t s) {
    // persistent workgroups walk the item list; split-K launches are never persistent (their
    // atomic epilogue reuses the ring): one workgroup per (tile, k-slice)
    const int items = p.tiles_m * p.tiles_n * p.ksplit;
    // measured (tools/gemm_lab): with K = 384..2048 one workgroup per item beats a persistent walk
    const int grid = items;
    (void)PERSISTENT_GRID;
    // at most one workgroup per CU anyway and >= 8 K-steps per workgroup: the four-stage ring
    const bool deep = items <= gv_cu_budget() && (p.ksplit > 1 ? p.k_per_split : p.K) >= 8 * BK;
    auto kern = deep ? gemm_deep_kernel<TA, TB, OutT, ATOMIC, EPI> : gemm_kernel<TA, TB, OutT, ATOMIC, EPI>;
    const int lds = deep ? DCfg::LDS : PCfg::LDS;
    static GvLdsOptIn opt_in[2];     // > 64 KiB of dynamic LDS: once per kernel instantiation and device
    if (int rc = gv_lds_opt_in(opt_in[deep], (const void*)kern, lds, "gemm")) return rc;
    int th = -1;
    if (gvtime::enabled()) {      // algorithmic bytes: operands once, output once, epilogue operands once
        const int e = EPI >= 0 ? EPI : p.epi;
        const double mn = (double)p.M * p.N;
        double bytes = 2.0 * p.K * ((double)p.M + p.N) + mn * sizeof(OutT);
        if (e & GV_EPI_ACCUM) bytes += mn * 4;
        if (e & GV_EPI_RESID) bytes += mn * 4;
        if (e & GV_EPI_DGELU) bytes += mn * 2;
        if (e & GV_EPI_SAVE_PRE) bytes += mn * 2;
        th = gvtime::begin(kernel_name<TA, TB, OutT, ATOMIC, EPI>(deep), 2.0 * p.M * p.N * p.K, bytes, s);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PCfg::THREADS), lds, s, p);
    gvtime::end(th, s);
    GV_LAUNCH_CHECK("gv_linear");
    return GV_OK;
}

// C[m][n] += sum_s slab[s][m][n]  (split-K reduce; 16 B per lane, fully coalesced)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, long MN4, int S, int N4, long ldc4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN4; i += (long)gridDim.x * 256) {
        // four slab loads in flight per thread (the slabs are independent streams; a rolled loop keeps one)
        f32x4 acc = ((const f32x4*)slab)[i];
        int s2 = 1;
        for (; s2 + 3 < S; s2 += 4) {
            const f32x4 a0 = ((const f32x4*)slab)[(long)s2 * MN4 + i], a1 = ((const f32x4*)slab)[(long)(s2 + 1) * MN4 + i];
            const f32x4 a2 = ((const f32x4*)slab)[(long)(s2 + 2) * MN4 + i], a3 = ((const f32x4*)slab)[(long)(s2 + 3) * MN4 + i];
            acc += (a0 + a1) + (a2 + a3);
        }
        for (; s2 < S; ++s2) acc += ((const f32x4*)slab)[(long)s2 * MN4 + i];
        const long m = i / N4, n4 = i - m * N4;
        f32x4* dst = (f32x4*)C + m * ldc4 + n4;
        *dst = *dst + acc;
    }
}

// split-K with an epilogue other than ACCUM: C = epilogue(sum_s slab[s]) -- the epilogue of gemm_core.h in its order (alpha is in the
// partial tiles already; bias, saved pre-activation, GELU, GELU' of the saved input, row factor + residual), 16 B of output columns per lane
struct SplitEpiP {
    const float* slab; void* C; const float* bias; const float* resid; const bf16* aux_in; bf16* aux_out; const float* row_scale;
    long ldc, ldr, ld_aux; int M, N, S, epi;
};
template <typename OutT>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const SplitEpiP q) {
    const int N4 = q.N >> 2;
    const long MN = (long)q.M * q.N, MN4 = MN >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN4; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N4), n = (int)(i - (long)m * N4) * 4;
        f32x4 v = *(const f32x4*)(q.slab + i * 4);
        for (int s2 = 1; s2 < q.S; ++s2) v += *(const f32x4*)(q.slab + (long)s2 * MN + i * 4);
        if (q.epi & GV_EPI_BIAS) v += *(const f32x4*)(q.bias + n);
        if (q.epi & GV_EPI_SAVE_PRE) *(bf16x4*)(q.aux_out + (long)m * q.ld_aux + n) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        if (q.epi & GV_EPI_GELU) { v[0] = gelu_f(v[0]); v[1] = gelu_f(v[1]); v[2] = gelu_f(v[2]); v[3] = gelu_f(v[3]); }
        if (q.epi & GV_EPI_DGELU) {
            const bf16x4 a = *(const bf16x4*)(q.aux_in + (long)m * q.ld_aux + n);
            v[0] *= dgelu_f((float)a[0]); v[1] *= dgelu_f((float)a[1]); v[2] *= dgelu_f((float)a[2]); v[3] *= dgelu_f((float)a[3]);
        }
        if (q.epi & GV_EPI_RESID) {
            if (q.row_scale) v *= q.row_scale[m];
            v += *(const f32x4*)(q.resid + (long)m * q.ldr + n);
        }
        if constexpr (sizeof(OutT) == 4) *(f32x4*)((float*)q.C + (long)m * q.ldc + n) = v;
        else *(bf16x4*)((bf16*)q.C + (long)m * q.ldc + n) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    }
}

// one launch over up to GV_DW_GROUP_MAX weight-gradient products that reduce over the same token rows: every workgroup
// runs ONE (problem, tile, k-slice) item of the split-K dW kernel, partial tiles go to the shared slab, one reduce
// launch folds all problems.  Items are slice-major over the concatenated tile lists, XCD-contiguous like make_walk.
struct GroupP { GemmP prob[GV_DW_GROUP_MAX]; int n, total_tiles; int tile_base[GV_DW_GROUP_MAX + 1]; };

__global__ __launch_bounds__(PCfg::THREADS, PCfg::THREADS * WGS_PER_CU / 256) void gemm_dw_group_kernel(const GroupP G) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int nb = gridDim.x, xcd = blockIdx.x & 7, lw = blockIdx.x >> 3;
    const int qd = nb >> 3, rm = nb & 7;
    const int b = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + lw;      // bijective XCD-contiguous remap
    const int slice = b / G.total_tiles, tg = b - slice * G.total_tiles;
    int q = 0;
#pragma unroll
    for (int i = 1; i < GV_DW_GROUP_MAX; ++i) q += (i < G.n && tg >= G.tile_base[i]) ? 1 : 0;
    const GemmP& g = G.prob[q];
    Walk wk;
    wk.first = slice * (g.tiles_m * g.tiles_n) + (tg - G.tile_base[q]); wk.stride = 1 << 30; wk.count = 1; wk.mlo = 0; wk.mcnt = g.tiles_m;
    gemm_body_w<PCfg, true, true, float, true, GV_EPI_ACCUM>(g, (GV_LDS char*)smem_raw, wk);
}

struct ReduceGroupP { const float* slab[GV_DW_GROUP_MAX]; float* C[GV_DW_GROUP_MAX]; long MN4[GV_DW_GROUP_MAX]; int N4[GV_DW_GROUP_MAX]; long ldc4[GV_DW_GROUP_MAX]; int S; };
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const ReduceGroupP R) {
    const int q = blockIdx.y;
    const float* __restrict__ slab = R.slab[q];
    const long MN4 = R.MN4[q];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN4; i += (long)gridDim.x * 256) {
        f32x4 acc = ((const f32x4*)slab)[i];
        int s2 = 1;
        for (; s2 + 1 < R.S; s2 += 2) acc += ((const f32x4*)slab)[(long)s2 * MN4 + i] + ((const f32x4*)slab)[(long)(s2 + 1) * MN4 + i];
        for (; s2 < R.S; ++s2) acc += ((const f32x4*)slab)[(long)s2 * MN4 + i];
        const long m = i / R.N4[q], n4 = i - m * R.N4[q];
        f32x4* dst = (f32x4*)R.C[q] + m * R.ldc4[q] + n4;
        *dst = *dst + acc;
    }
}

constexpr long WORKSPACE_BYTES = 128L << 20;
// gv_workspace_bytes: the entry points run their own kernel selection and k-slice sizing in PLAN mode -- nothing is launched, the
// scratch the call would use is recorded instead (so the answer cannot drift from what a launch does)
thread_local bool g_plan = false;
thread_local long g_plan_bytes = 0;

// k-slices for `tiles` one-per-CU workgroups over `ktiles` 64-deep K-tiles: the S that minimises  rounds of 256 workgroups x
// K-tiles per slice  (ViT-S block: 36 tiles -> S = 7, one round; ViT-B block: 144 tiles -> S = 3, two rounds of 2/3 the
// depth of S = 1), smallest S on ties, at least four K-tiles per slice, slabs within the workspace
int dw8_pick_slices(int tiles, int ktiles, long mn_floats, long workspace_bytes) {
    int best = 1; long best_cost = -1;
    for (int S = 1; S <= 64; ++S) {
        if (S > 1 && (ktiles / S < 4 || (long)S * mn_floats * 4 > workspace_bytes)) break;
        const int cus = gv_cu_budget();
        const long cost = (long)((tiles * S + cus - 1) / cus) * ((ktiles + S - 1) / S);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = S; }
    }
    return best;
}

// dW over many token rows on the 8-wave ping-pong kernel (gemm_dw8.h).  Returns -1 when the shape is not one of its own
// (the caller then takes the 128x128 split-K path), else the launch status.
template <bool SWAP>
int launch_dw8(const Dw8P& q, float* C, long ldc, int M, int N, hipStream_t s) {
    auto kern = dw8_kernel<SWAP>;
    static GvLdsOptIn opt_in;
    if (int rc = gv_lds_opt_in(opt_in, (const void*)kern, DW8_LDS, "gemm(dw8)")) return rc;
    const int th = gvtime::enabled() ? gvtime::begin(SWAP ? "dw8_kernel<true>" : "dw8_kernel<false>", 2.0 * M * N * q.K, 2.0 * q.K * ((double)M + N) + 8.0 * M * N, s) : -1;
    hipLaunchKernelGGL(kern, dim3(q.tiles_p * q.tiles_q * q.ksplit), dim3(512), DW8_LDS, s, q);
    gvtime::end(th, s);
    GV_LAUNCH_CHECK("gv_linear(dw8)");
    const long MN4 = (long)M * N / 4;
    long blocks = (MN4 + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)q.slab, C, MN4, q.ksplit, N / 4, ldc / 4);
    GV_LAUNCH_CHECK("gv_linear(dw8 reduce)");
    return GV_OK;
}

int try_dw8(const gv_linear_args* a, hipStream_t s) {
    if (!(a->trans_a && a->trans_b && a->c_is_f32 && a->epilogue == GV_EPI_ACCUM && a->workspace && gv_aligned(a->workspace, 16))) return -1;
    if (a->alpha != 0.f && a->alpha != 1.f) return -1;
    if (a->K < 2048) return -1;
    const bool normal = a->M % 128 == 0 && a->N % 384 == 0;
    const bool swapped = !normal && a->N % 128 == 0 && a->M % 384 == 0 && (!a->colsum_a || a->N / 128 >= 12);
    if (!normal && !swapped) return -1;
    Dw8P q;
    q.K = a->K; q.slab = (float*)a->workspace; q.colsum = a->colsum_a;
    if (normal) { q.P = (const bf16*)a->A; q.ldp = a->lda; q.Pn = a->M; q.Q = (const bf16*)a->B; q.ldq = a->ldb; q.Qn = a->N; }
    else { q.P = (const bf16*)a->B; q.ldp = a->ldb; q.Pn = a->N; q.Q = (const bf16*)a->A; q.ldq = a->lda; q.Qn = a->M; }
    q.tiles_p = q.Pn / 128; q.tiles_q = q.Qn / 384;
    const int tiles = q.tiles_p * q.tiles_q;
    if (tiles > 256) return -1;
    const int ktiles = (a->K + 63) / 64;
    int S = dw8_pick_slices(tiles, ktiles, (long)a->M * a->N, a->workspace_bytes);
    const int per = (ktiles + S - 1) / S;
    S = (ktiles + per - 1) / per;
    if ((long)S * a->M * a->N * 4 > a->workspace_bytes) return -1;
    if (g_plan) { g_plan_bytes = (long)S * a->M * a->N * 4; return GV_OK; }
    q.ksplit = S; q.k_per_split = per * 64;
    return normal ? launch_dw8<false>(q, (float*)a->C, a->ldc, a->M, a->N, s) : launch_dw8<true>(q, (float*)a->C, a->ldc, a->M, a->N, s);
}

}  // namespace

extern "C" int64_t gv_linear_workspace_bytes(void) { return WORKSPACE_BYTES; }

extern "C" int64_t gv_workspace_bytes(int32_t op, const void* args) {
    if (!args) { gv_set_error("gv_workspace_bytes: null args"); return -1; }
    g_plan = true; g_plan_bytes = 0;
    int rc;
    if (op == GV_OP_LINEAR) {
        gv_linear_args a = *(const gv_linear_args*)args;
        a.workspace = (float*)(uintptr_t)256; a.workspace_bytes = WORKSPACE_BYTES;      // "as if the full scratch were offered" (never dereferenced)
        rc = gv_linear(&a, nullptr);
    } else if (op == GV_OP_LINEAR_DW_GROUP) {
        gv_linear_dw_group_args a = *(const gv_linear_dw_group_args*)args;
        a.workspace = (float*)(uintptr_t)256; a.workspace_bytes = WORKSPACE_BYTES;
        rc = gv_linear_dw_group(&a, nullptr);
    } else {
        g_plan = false;
        gv_set_error("gv_workspace_bytes: op %d takes no scratch query (GV_OP_LINEAR, GV_OP_LINEAR_DW_GROUP)", op);
        return -1;
    }
    g_plan = false;
    return rc == GV_OK ? (int64_t)g_plan_bytes : -1;
}

// ---- live per-kernel timing (timing.h; gv_linear_timing / gv_linear_timing_read of include/gipvit.h)
namespace {
struct TimingRec { const char* name; hipEvent_t e0, e1; double flops, bytes; };
struct Timing {
    std::mutex mu;
    bool on = false;
    std::vector<TimingRec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
Timing& timing() { static Timing t; return t; }
}  // namespace

bool gvtime::enabled() { return timing().on; }
int gvtime::begin(const char* kernel_name, double flops, double bytes, hipStream_t s) {
    Timing& tm = timing();
    std::lock_guard<std::mutex> lk(tm.mu);
    if (!tm.on) return -1;
    TimingRec r{kernel_name, tm.get(), tm.get(), flops, bytes};
    (void)hipEventRecord(r.e0, s);
    tm.recs.push_back(r);
    return (int)tm.recs.size() - 1;
}
void gvtime::end(int handle, hipStream_t s) {
    if (handle < 0) return;
    Timing& tm = timing();
    std::lock_guard<std::mutex> lk(tm.mu);
    if (handle < (int)tm.recs.size()) (void)hipEventRecord(tm.recs[handle].e1, s);
}

extern "C" int gv_linear_timing(int enable) {
    Timing& tm = timing();
    std::lock_guard<std::mutex> lk(tm.mu);
    if (enable) {
        for (auto& r : tm.recs) { tm.pool.push_back(r.e0); tm.pool.push_back(r.e1); }
        tm.recs.clear();
    }
    tm.on = enable != 0;
    return GV_OK;
}

extern "C" int gv_linear_timing_read(gv_linear_timing_row* rows, int max_rows) {
    GV_REQUIRE(rows && max_rows > 0, GV_E_NULL, "gv_linear_timing_read: null rows");
    Timing& tm = timing();
    std::lock_guard<std::mutex> lk(tm.mu);
    int n = 0;
    for (auto& r : tm.recs) {
        hipError_t e = hipEventSynchronize(r.e1);
        if (e != hipSuccess) { gv_set_error("gv_linear_timing_read: %s", hipGetErrorString(e)); return (int)e; }
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.e0, r.e1);
        if (e != hipSuccess) { gv_set_error("gv_linear_timing_read: %s", hipGetErrorString(e)); return (int)e; }
        int i = 0;
        while (i < n && strcmp(rows[i].name, r.name) != 0) ++i;
        if (i == n) {
            if (n == max_rows) continue;
            memset(&rows[n], 0, sizeof(rows[n]));
            strncpy(rows[n].name, r.name, sizeof(rows[n].name) - 1);
            ++n;
        }
        rows[i].launches += 1; rows[i].seconds += ms * 1e-3; rows[i].flops += r.flops; rows[i].bytes += r.bytes;
    }
    return n;
}

extern "C" int gv_linear(const gv_linear_args* a, void* stream) {
    GV_REQUIRE(a && a->A && a->B && a->C, GV_E_NULL, "gv_linear: null operand");
    GV_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, GV_E_SHAPE, "gv_linear: M,N,K must be > 0 (got %d,%d,%d)", a->M, a->N, a->K);
    const bool ta = a->trans_a != 0, tb = a->trans_b != 0;
    GV_REQUIRE(!(ta && !tb), GV_E_UNSUPPORTED, "gv_linear: (trans_a=1, trans_b=0) is not built");
    if (!ta || !tb)
        GV_REQUIRE(a->K % 32 == 0, GV_E_SHAPE, "gv_linear: K=%d must be a multiple of 32 when an operand is k-contiguous", a->K);
    if (ta) GV_REQUIRE(a->M % 8 == 0, GV_E_SHAPE, "gv_linear: M=%d must be a multiple of 8 with trans_a", a->M);
    if (tb) GV_REQUIRE(a->N % 8 == 0, GV_E_SHAPE, "gv_linear: N=%d must be a multiple of 8 with trans_b", a->N);
    GV_REQUIRE(a->lda % 8 == 0 && a->ldb % 8 == 0, GV_E_ALIGN, "gv_linear: lda/ldb must be multiples of 8 elements");
    GV_REQUIRE(gv_aligned(a->A, 16) && gv_aligned(a->B, 16) && gv_aligned(a->C, 16), GV_E_ALIGN, "gv_linear: A/B/C must be 16-byte aligned");
    const int e = a->epilogue;
    GV_REQUIRE((e & ~GV_EPI_ALL) == 0, GV_E_UNSUPPORTED, "gv_linear: epilogue 0x%x has bits outside the GV_EPI_* mask 0x%x", e, GV_EPI_ALL);
    if (e & GV_EPI_BIAS) GV_REQUIRE(a->bias && gv_aligned(a->bias, 16), GV_E_NULL, "gv_linear: BIAS needs an aligned bias");
    if (e & GV_EPI_RESID) GV_REQUIRE(a->resid && a->ldr % 4 == 0, GV_E_NULL, "gv_linear: RESID needs resid with ldr %% 4 == 0");
    if (e & GV_EPI_DGELU) GV_REQUIRE(a->aux_in && a->ld_aux % 4 == 0, GV_E_NULL, "gv_linear: DGELU needs aux_in");
    if (e & GV_EPI_SAVE_PRE) GV_REQUIRE(a->aux_out && a->ld_aux % 4 == 0, GV_E_NULL, "gv_linear: SAVE_PRE needs aux_out");
    if (e & GV_EPI_POS) GV_REQUIRE(a->pos && a->P > 0, GV_E_NULL, "gv_linear: POS needs pos and P");
    if (e & GV_EPI_ACCUM) GV_REQUIRE(a->c_is_f32, GV_E_UNSUPPORTED, "gv_linear: ACCUM needs an f32 C");
    GV_REQUIRE(a->ldc % 4 == 0, GV_E_ALIGN, "gv_linear: ldc must be a multiple of 4");

    {
        int rc = try_dw8(a, (hipStream_t)stream);
        if (rc != -1) return rc;
        rc = gv_panel_wide(a, g_plan ? (hipStream_t)(intptr_t)-1 : (hipStream_t)stream);      // ((hipStream_t)-1: "would this call run there?")
        if (rc != -1) return rc;
    }
    GemmP p;
    p.A = (const bf16*)a->A; p.B = (const bf16*)a->B; p.C = a->C;
    p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc;
    p.epi = e; p.bias = a->bias; p.resid = a->resid; p.ldr = a->ldr;
    p.aux_in = (const bf16*)a->aux_in; p.ld_aux = a->ld_aux; p.aux_out = (bf16*)a->aux_out;
    p.pos = a->pos; p.P = a->P; p.alpha = a->alpha == 0.f ? 1.f : a->alpha;
    p.tiles_m = (a->M + BM - 1) / BM; p.tiles_n = (a->N + BN - 1) / BN;
    p.ksplit = 1; p.k_per_split = ((a->K + BK - 1) / BK) * BK;
    p.order = 0;
    p.colsum_a = a->colsum_a;
    p.row_scale = a->row_scale;
    p.slab = nullptr;
    if (a->colsum_a) GV_REQUIRE(ta, GV_E_UNSUPPORTED, "gv_linear: colsum_a needs trans_a (it sums the dW product's A operand)");
    hipStream_t s = (hipStream_t)stream;

    // Split K when the output grid alone cannot fill 256 CUs (dW: reduction over tens of
    // thousands of tokens; the 65536-class DINO head's dX).  Partial sums meet in C through
    // f32 atomics, which needs pure ACCUM semantics (C already holds the value to add to).
    {
        const int tiles = p.tiles_m * p.tiles_n;
        const bool only_accum = a->c_is_f32 && e == GV_EPI_ACCUM;
        if (only_accum && tiles < 192 * WGS_PER_CU && (a->N & 7) == 0) {
            // at most as many workgroups as are resident at once: one more would run a second round alone
            constexpr int SLOTS = 256 * WGS_PER_CU;      // (halving the slices -- 16 MB of slab instead of 32 -- measured 80 vs 63 us per launch)
            const int want = SLOTS / tiles > 0 ? SLOTS / tiles : 1;
            const int ksteps = (a->K + BK - 1) / BK;
            const int min_steps = 512 / BK;
            const int maxs = ksteps / min_steps > 0 ? ksteps / min_steps : 1;   // at least 512 deep per slice
            const int S = want < maxs ? want : maxs;
            if (S > 1) {
                const int per = (ksteps + S - 1) / S;
                p.k_per_split = per * BK;
                p.ksplit = (ksteps + per - 1) / per;
            }
        }
    }
    // Few tiles, a long reduction and an epilogue other than ACCUM (the DINO head's 2048-wide layers, the CLS-only tail's fc2 at 128 /
    // 640 rows: 15 - 80 workgroups with 24 - 32 serial K-steps of ~1 us): with a caller's scratch, split K over the idle CUs as well --
    // >= 4 K-steps per slice, at most one workgroup per CU -- and let ONE pass sum the partial tiles and apply the epilogue.
    bool epi_split = false;
    if (p.ksplit == 1 && !ta && !(e & (GV_EPI_ACCUM | GV_EPI_POS)) && (a->N & 7) == 0 && a->workspace && gv_aligned(a->workspace, 16) && !a->colsum_a &&
        (!(e & (GV_EPI_DGELU | GV_EPI_SAVE_PRE)) || a->ld_aux % 4 == 0)) {
        const int tiles = p.tiles_m * p.tiles_n, ksteps = (a->K + BK - 1) / BK;
        if (tiles <= 96 && ksteps >= 16) {
            int S = gv_cu_budget() / tiles < ksteps / 4 ? gv_cu_budget() / tiles : ksteps / 4;
            if (S >= 2) {
                const int per = (ksteps + S - 1) / S;
                S = (ksteps + per - 1) / per;
                if ((long)S * a->M * a->N * 4 <= a->workspace_bytes) { p.k_per_split = per * BK; p.ksplit = S; epi_split = true; }
            }
        }
    }
    // plan mode (gv_workspace_bytes) applies the launch's own fit test below: a slab that does not fit the scratch is not used (atomics)
    if (g_plan) { const long need = p.ksplit > 1 ? (long)p.ksplit * a->M * a->N * 4 : 0; g_plan_bytes = need <= a->workspace_bytes ? need : 0; return GV_OK; }
    bool slab = false;
    if (p.ksplit > 1 && a->workspace && (long)p.ksplit * a->M * a->N * 4 <= a->workspace_bytes && gv_aligned(a->workspace, 16)) {
        p.slab = a->workspace;
        slab = true;
    }
    auto finish = [&](int rc) -> int {
        if (rc != GV_OK || !slab) return rc;
        if (epi_split) {
            SplitEpiP q{(const float*)p.slab, a->C, a->bias, a->resid, (const bf16*)a->aux_in, (bf16*)a->aux_out, a->row_scale,
                        a->ldc, a->ldr, a->ld_aux, a->M, a->N, p.ksplit, e};
            const long MN4e = (long)a->M * a->N / 4;
            long blk = (MN4e + 255) / 256; if (blk > 2048) blk = 2048;
            if (a->c_is_f32) hipLaunchKernelGGL(splitk_epilogue_kernel<float>, dim3((unsigned)blk), dim3(256), 0, s, q);
            else hipLaunchKernelGGL(splitk_epilogue_kernel<bf16>, dim3((unsigned)blk), dim3(256), 0, s, q);
            GV_LAUNCH_CHECK("gv_linear(splitk_epilogue)");
            return GV_OK;
        }
        const long MN4 = (long)a->M * a->N / 4;
        long blocks = (MN4 + 255) / 256; if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)p.slab, (float*)a->C, MN4, p.ksplit,
                           a->N / 4, a->ldc / 4);
        GV_LAUNCH_CHECK("gv_linear(splitk_reduce)");
        return GV_OK;
    };
    // specialised epilogue masks of the hot path; anything else (and any N % 4 != 0) runs the
    // runtime-mask build
    const bool generic = (a->N & 7) != 0;
    constexpr int E_B = GV_EPI_BIAS, E_BGS = GV_EPI_BIAS | GV_EPI_GELU | GV_EPI_SAVE_PRE, E_BG = GV_EPI_BIAS | GV_EPI_GELU, E_BR = GV_EPI_BIAS | GV_EPI_RESID,
                  E_BP = GV_EPI_BIAS | GV_EPI_POS, E_DG = GV_EPI_DGELU, E_ACC = GV_EPI_ACCUM;
    if (ta && tb) {
        if (p.ksplit > 1 && !generic) return finish(launch<true, true, float, true, E_ACC>(p, s));
        if (a->c_is_f32) {
            if (!generic && e == 0) return launch<true, true, float, false, 0>(p, s);
            if (!generic && e == E_ACC) return launch<true, true, float, false, E_ACC>(p, s);
            return launch<true, true, float, false>(p, s);
        }
        return launch<true, true, bf16, false>(p, s);
    }
    if (!ta && tb) {
        if (p.ksplit > 1 && !generic) return finish(launch<false, true, float, true, E_ACC>(p, s));
        if (a->c_is_f32) return launch<false, true, float, false>(p, s);
        if (!generic && e == 0) return launch<false, true, bf16, false, 0>(p, s);
        if (!generic && e == E_DG) return launch<false, true, bf16, false, E_DG>(p, s);
        return launch<false, true, bf16, false>(p, s);
    }
    if (p.ksplit > 1 && !generic) return finish(launch<false, false, float, true, E_ACC>(p, s));
    if (a->c_is_f32) {
        if (!generic && e == 0) return launch<false, false, float, false, 0>(p, s);
        if (!generic && e == E_B) return launch<false, false, float, false, E_B>(p, s);
        if (!generic && e == E_BR) return launch<false, false, float, false, E_BR>(p, s);
        if (!generic && e == E_BP) return launch<false, false, float, false, E_BP>(p, s);
        return launch<false, false, float, false>(p, s);
    }
    if (!generic && e == 0) return launch<false, false, bf16, false, 0>(p, s);
    if (!generic && e == E_B) return launch<false, false, bf16, false, E_B>(p, s);
    if (!generic && e == E_BGS) return launch<false, false, bf16, false, E_BGS>(p, s);
    if (!generic && e == E_BG) return launch<false, false, bf16, false, E_BG>(p, s);
    return launch<false, false, bf16, false>(p, s);
}

extern "C" int gv_linear_dw_group(const gv_linear_dw_group_args* a, void* stream) {
    GV_REQUIRE(a && a->n >= 1 && a->n <= GV_DW_GROUP_MAX, GV_E_SHAPE, "gv_linear_dw_group: 1..%d problems", GV_DW_GROUP_MAX);
    GV_REQUIRE(a->K > 0 && a->workspace && gv_aligned(a->workspace, 16), GV_E_NULL, "gv_linear_dw_group: K > 0 and an aligned workspace are required");
    hipStream_t s = (hipStream_t)stream;
    // ---- ping-pong kernel (gemm_dw8.h) when every problem tiles into 128 x 384 pieces
    {
        bool ok8 = a->K >= 2048;
        int tiles8 = 0;
        for (int q = 0; q < a->n && ok8; ++q) {
            const auto& pr = a->prob[q];
            ok8 = pr.dY && pr.X && pr.dW && pr.M > 0 && pr.N > 0 && pr.M % 128 == 0 && pr.N % 384 == 0 && pr.ldy % 8 == 0 && pr.ldx % 8 == 0 && pr.ldw % 4 == 0 &&
                  gv_aligned(pr.dY, 16) && gv_aligned(pr.X, 16) && gv_aligned(pr.dW, 16);
            if (ok8) tiles8 += (pr.M / 128) * (pr.N / 384);
        }
        if (ok8 && tiles8 <= 256) {
            const int ktiles = (a->K + 63) / 64;
            long mn = 0;
            for (int q = 0; q < a->n; ++q) mn += (long)a->prob[q].M * a->prob[q].N;
            int S = dw8_pick_slices(tiles8, ktiles, mn, a->workspace_bytes);
            const int per = (ktiles + S - 1) / S;
            S = (ktiles + per - 1) / per;
            Dw8GroupP G8{};
            G8.n = a->n;
            long slab_floats = 0; int tb = 0; double flops = 0, bytes = 0;
            for (int q = 0; q < a->n; ++q) {
                const auto& pr = a->prob[q];
                Dw8P& d = G8.prob[q];
                d.P = (const bf16*)pr.dY; d.ldp = pr.ldy; d.Pn = pr.M; d.Q = (const bf16*)pr.X; d.ldq = pr.ldx; d.Qn = pr.N;
                d.K = a->K; d.tiles_p = pr.M / 128; d.tiles_q = pr.N / 384; d.ksplit = S; d.k_per_split = per * 64;
                d.slab = a->workspace + slab_floats; d.colsum = pr.colsum_dy;
                slab_floats += (long)S * pr.M * pr.N;
                G8.tile_base[q] = tb; tb += d.tiles_p * d.tiles_q;
                flops += 2.0 * pr.M * pr.N * a->K; bytes += 2.0 * a->K * ((double)pr.M + pr.N) + 8.0 * pr.M * pr.N;
            }
            for (int q = a->n; q <= GV_DW_GROUP_MAX; ++q) G8.tile_base[q] = tb;
            G8.total_tiles = tb;
            if (slab_floats * 4 <= a->workspace_bytes) {
                if (g_plan) { g_plan_bytes = slab_floats * 4; return GV_OK; }
                static GvLdsOptIn opt_in8;
                if (int rc = gv_lds_opt_in(opt_in8, (const void*)dw8_group_kernel, DW8_LDS, "gv_linear_dw_group")) return rc;
                const int th = gvtime::enabled() ? gvtime::begin("dw8_group_kernel", flops, bytes, s) : -1;
                hipLaunchKernelGGL(dw8_group_kernel, dim3(tb * S), dim3(512), DW8_LDS, s, G8);
                gvtime::end(th, s);
                GV_LAUNCH_CHECK("gv_linear_dw_group(dw8)");
                ReduceGroupP R{};
                long max4 = 0;
                for (int q = 0; q < a->n; ++q) {
                    const auto& pr = a->prob[q];
                    R.slab[q] = G8.prob[q].slab; R.C[q] = pr.dW; R.MN4[q] = (long)pr.M * pr.N / 4; R.N4[q] = pr.N / 4; R.ldc4[q] = pr.ldw / 4;
                    if (R.MN4[q] > max4) max4 = R.MN4[q];
                }
                R.S = S;
                long blocks = (max4 + 255) / 256; if (blocks > 1024) blocks = 1024;
                hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3((unsigned)blocks, a->n), dim3(256), 0, s, R);
                GV_LAUNCH_CHECK("gv_linear_dw_group(dw8 reduce)");
                return GV_OK;
            }
        }
    }
    GroupP G{};
    G.n = a->n;
    int tiles = 0;
    for (int q = 0; q < a->n; ++q) {
        const auto& pr = a->prob[q];
        GV_REQUIRE(pr.dY && pr.X && pr.dW, GV_E_NULL, "gv_linear_dw_group: null operand in problem %d", q);
        GV_REQUIRE(pr.M > 0 && pr.N > 0 && pr.M % 8 == 0 && pr.N % 8 == 0, GV_E_SHAPE, "gv_linear_dw_group: M, N must be positive multiples of 8 (problem %d: %d x %d)", q, pr.M, pr.N);
        GV_REQUIRE(pr.ldy % 8 == 0 && pr.ldx % 8 == 0 && pr.ldw % 4 == 0, GV_E_ALIGN, "gv_linear_dw_group: leading dimensions misaligned (problem %d)", q);
        GV_REQUIRE(gv_aligned(pr.dY, 16) && gv_aligned(pr.X, 16) && gv_aligned(pr.dW, 16), GV_E_ALIGN, "gv_linear_dw_group: operands must be 16-byte aligned");
        GemmP& p = G.prob[q];
        p.A = (const bf16*)pr.dY; p.B = (const bf16*)pr.X; p.C = pr.dW;
        p.M = pr.M; p.N = pr.N; p.K = a->K; p.lda = pr.ldy; p.ldb = pr.ldx; p.ldc = pr.ldw;
        p.epi = GV_EPI_ACCUM; p.bias = nullptr; p.resid = nullptr; p.ldr = 0; p.aux_in = nullptr; p.ld_aux = 0; p.aux_out = nullptr;
        p.pos = nullptr; p.P = 0; p.alpha = 1.f;
        p.tiles_m = (pr.M + BM - 1) / BM; p.tiles_n = (pr.N + BN - 1) / BN;
        p.order = 0; p.colsum_a = pr.colsum_dy; p.row_scale = nullptr;
        G.tile_base[q] = tiles;
        tiles += p.tiles_m * p.tiles_n;
    }
    for (int q = a->n; q <= GV_DW_GROUP_MAX; ++q) G.tile_base[q] = tiles;
    G.total_tiles = tiles;
    // as many k-slices as fill the resident workgroup slots once (at least 512 rows per slice)
    constexpr int SLOTS = 256 * WGS_PER_CU;
    const int ksteps = (a->K + BK - 1) / BK;
    int S = SLOTS / tiles > 0 ? SLOTS / tiles : 1;
    const int maxs = ksteps / (512 / BK) > 0 ? ksteps / (512 / BK) : 1;
    if (S > maxs) S = maxs;
    const int per = (ksteps + S - 1) / S;
    const int ksplit = (ksteps + per - 1) / per;
    long slab_floats = 0;
    for (int q = 0; q < a->n; ++q) {
        GemmP& p = G.prob[q];
        p.k_per_split = per * BK; p.ksplit = ksplit;
        p.slab = a->workspace + slab_floats;
        slab_floats += (long)ksplit * p.M * p.N;
    }
    if (g_plan) { g_plan_bytes = slab_floats * 4; return GV_OK; }
    GV_REQUIRE(slab_floats * 4 <= a->workspace_bytes, GV_E_SHAPE, "gv_linear_dw_group: workspace too small (%ld bytes needed)", slab_floats * 4);
    static GvLdsOptIn opt_in;
    if (int rc = gv_lds_opt_in(opt_in, (const void*)gemm_dw_group_kernel, PCfg::LDS, "gv_linear_dw_group")) return rc;
    double flops = 0, bytes = 0;
    for (int q = 0; q < a->n; ++q) { flops += 2.0 * G.prob[q].M * G.prob[q].N * a->K; bytes += 2.0 * a->K * ((double)G.prob[q].M + G.prob[q].N) + 8.0 * G.prob[q].M * G.prob[q].N; }
    const int th = gvtime::enabled() ? gvtime::begin("gemm_dw_group_kernel", flops, bytes, s) : -1;
    hipLaunchKernelGGL(gemm_dw_group_kernel, dim3(tiles * ksplit), dim3(PCfg::THREADS), PCfg::LDS, s, G);
    gvtime::end(th, s);
    GV_LAUNCH_CHECK("gv_linear_dw_group");
    ReduceGroupP R{};
    long max4 = 0;
    for (int q = 0; q < a->n; ++q) {
        R.slab[q] = G.prob[q].slab; R.C[q] = (float*)G.prob[q].C; R.MN4[q] = (long)G.prob[q].M * G.prob[q].N / 4;
        R.N4[q] = G.prob[q].N / 4; R.ldc4[q] = G.prob[q].ldc / 4;
        if (R.MN4[q] > max4) max4 = R.MN4[q];
    }
    R.S = ksplit;
    long blocks = (max4 + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3((unsigned)blocks, a->n), dim3(256), 0, s, R);
    GV_LAUNCH_CHECK("gv_linear_dw_group(reduce)");
    return GV_OK;
}
