// bf16 MFMA GEMM for gfx950 with fused epilogues -- gv_linear (include/gipvit.h).
//
// Replaces the ATen addmm / mm calls behind nn.Linear forward and backward in the
// reference's ViT blocks, patch embedding and DINOHead (vit.pyc@L98-104, L119-131,
// L167-170, L326-330; reference train.py:1045 forward, :1071 backward).
//
// Design (MI355X_MICROARCH / cdna_hip_programming section 5):
//   * 128x128 output tile, BK = 64, 256 threads = 4 waves (2x2), each wave 64x64 =
//     4x4 v_mfma_f32_16x16x32_bf16 tiles, f32 accumulators.
//   * Operand tiles go global -> LDS with LDS-DMA (global_load_lds_dwordx4, 16 B per
//     lane, 1 KiB per wave-instruction), double buffered.  The LDS image is lane-
//     linear, so the bank swizzle is applied to the per-lane SOURCE address and
//     again on the read (rule 21).
//   * "natural" operand (reduction index contiguous, e.g. x[M,K], W[N,K]): LDS rows of
//     128 B, 16-B chunk index XOR (row & 7), fragments by ds_read_b128.
//   * "transposed" operand (reduction index strided, e.g. dY[tokens,N] for dW, W[N,K]
//     for dX): LDS image [64 k][128 cols], 32-B chunk index XOR f(k), fragments by
//     two ds_read_b64_tr_b16 (hardware transpose) -- no transposed copies in HBM.
//   * MFMA roles are swapped (weights fragment as the A operand) so each lane ends up
//     with 4 consecutive output columns: 8-B (bf16) / 16-B (f32) stores.
//   * split-K (dW over tens of thousands of tokens) accumulates with f32 atomics
//     shaped as 256 contiguous bytes per wave-instruction via an LDS transpose.
//   * blockIdx -> tile map is XCD-aware (bijective T1 remap): blocks that share an
//     A row panel land on one XCD's L2.
#include "gv_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;       // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;      // A + B
constexpr int LDS_BYTES = 2 * BUF_BYTES;       // double buffered = 64 KiB

__device__ __attribute__((aligned(256))) unsigned short gv_zero_page[128];   // 256 B of zeros

struct GemmP {
    const bf16* A; const bf16* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    int epi;
    const float* bias; const float* resid; long ldr;
    const bf16* aux_in; long ld_aux; bf16* aux_out;
    const float* pos; int P;
    float alpha;
    int tiles_m, tiles_n, ksplit, k_per_split;
};

__device__ __forceinline__ int swz_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

__device__ __forceinline__ void glds16(const void* src, GV_LDS char* dst) {
    __builtin_amdgcn_global_load_lds((const GV_GLOBAL void*)src, (GV_LDS void*)dst, 16, 0, 0);
}

// Stage one 128 x 64 (natural) or 64 x 128 (transposed) operand tile.  Each wave
// issues 4 LDS-DMA pieces of 1 KiB.  `o0` = first output row/col of the tile,
// `lim` = number of valid output rows/cols, `k0` = first reduction index,
// `klim` = reduction length.
template <bool T>
__device__ __forceinline__ void stage_tile(const bf16* __restrict__ base, long ld, int o0, int lim, int k0, int klim,
                                           GV_LDS char* tile, int wave, int lane) {
    if constexpr (!T) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = wave * 4 + p;
            const int r = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            int row = o0 + r;
            row = row < lim ? row : lim - 1;
            const bf16* src = base + (long)row * ld + k0 + c * 8;
            glds16(src, tile + __builtin_amdgcn_readfirstlane(piece * 1024));
        }
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = wave * 4 + p;
            const int r = piece * 4 + (lane >> 4);
            const int s16 = lane & 15;
            const int logical = (s16 >> 1) ^ swz_f(r);
            const int col = o0 + logical * 16 + (s16 & 1) * 8;
            const int kk = k0 + r;
            const bool ok = (kk < klim) && (col + 8 <= lim);
            const bf16* src = ok ? base + (long)kk * ld + col : (const bf16*)gv_zero_page + s16 * 8;
            glds16(src, tile + __builtin_amdgcn_readfirstlane(piece * 1024));
        }
    }
}

// Fragment (8 bf16 along the reduction index for one output row/col) of the 16-wide
// block `blk16` (index in units of 16 rows/cols inside the tile), k-step ks (0/1).
template <bool T>
__device__ __forceinline__ bf16x8 read_frag(GV_LDS char* tile, int blk16, int ks, int lane) {
    if constexpr (!T) {
        const int row = blk16 * 16 + (lane & 15);
        const int chunk = ks * 4 + (lane >> 4);
        return *(GV_LDS bf16x8*)(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
    } else {
        const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
        const int kr = ks * 32 + g * 8 + q;
        GV_LDS char* a0 = tile + kr * 256 + ((blk16 ^ swz_f(kr)) << 5) + p * 8;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((GV_LDS bf16x4*)a0);
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((GV_LDS bf16x4*)(a0 + 4 * 256));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <typename OutT> __device__ __forceinline__ void store4(OutT* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) {
    *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
}
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const float* v) {
    *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}

template <bool TA, bool TB, typename OutT, bool ATOMIC>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmP g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    GV_LDS char* smem = (GV_LDS char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware, bijective block remap (blocks b and b+8 share an XCD)
    const int nwg = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_n = bid % g.tiles_n, tile_m = bid / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.y * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nt = (kend - kbeg + BK - 1) / BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nt > 0) {
        stage_tile<TA>(g.A, g.lda, m0, g.M, kbeg, kend, smem, wave, lane);
        stage_tile<TB>(g.B, g.ldb, n0, g.N, kbeg, kend, smem + TILE_BYTES, wave, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA landed (own pieces) ...
    __syncthreads();                                    // ... and everybody else's

    for (int t = 0; t < nt; ++t) {
        GV_LDS char* cur = smem + (t & 1) * BUF_BYTES;
        if (t + 1 < nt) {
            GV_LDS char* nxt = smem + ((t + 1) & 1) * BUF_BYTES;
            const int k0 = kbeg + (t + 1) * BK;
            stage_tile<TA>(g.A, g.lda, m0, g.M, k0, kend, nxt, wave, lane);
            stage_tile<TB>(g.B, g.ldb, n0, g.N, k0, kend, nxt + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = read_frag<TA>(cur, wm * 4 + i, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = read_frag<TB>(cur + TILE_BYTES, wn * 4 + j, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue.  acc[i][j][r]: m = m0 + wm*64 + i*16 + (lane&15),
    //                                n = n0 + wn*64 + j*16 + (lane>>4)*4 + r
    const int li = lane & 15, gq = lane >> 4;
    if constexpr (ATOMIC) {
        // per-wave [64 m][64 n] f32 image in LDS, then 256-B-per-instruction atomics
        GV_LDS float* img = (GV_LDS float*)(smem + wave * 16384);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[i][j] * g.alpha;
                *(GV_LDS f32x4*)(img + (i * 16 + li) * 64 + j * 16 + gq * 4) = v;
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): own writes visible to own reads
        float* C = (float*)g.C;
        const int n = n0 + wn * 64 + lane;
        for (int r = 0; r < 64; ++r) {
            const int m = m0 + wm * 64 + r;
            if (m < g.M && n < g.N) atomicAdd(C + (long)m * g.ldc + n, img[r * 64 + lane]);
        }
    } else {
        OutT* C = (OutT*)g.C;
        const int epi = g.epi;
        const bool vec_ok = ((g.N & 3) == 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + li;
            if (m >= g.M) continue;
            long orow = m;
            int prow = 0;
            if (epi & GV_EPI_POS) { orow = m + m / g.P + 1; prow = (m % g.P) + 1; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + gq * 4;
                if (n >= g.N) continue;
                float v[4] = {acc[i][j][0] * g.alpha, acc[i][j][1] * g.alpha, acc[i][j][2] * g.alpha, acc[i][j][3] * g.alpha};
                if (vec_ok) {
                    if (epi & GV_EPI_BIAS) { f32x4 b = *(const f32x4*)(g.bias + n); v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3]; }
                    if (epi & GV_EPI_SAVE_PRE) store4<bf16>(g.aux_out + orow * g.ld_aux + n, v);
                    if (epi & GV_EPI_GELU) { v[0] = gelu_f(v[0]); v[1] = gelu_f(v[1]); v[2] = gelu_f(v[2]); v[3] = gelu_f(v[3]); }
                    if (epi & GV_EPI_DGELU) {
                        bf16x4 a = *(const bf16x4*)(g.aux_in + orow * g.ld_aux + n);
                        v[0] *= dgelu_f((float)a[0]); v[1] *= dgelu_f((float)a[1]); v[2] *= dgelu_f((float)a[2]); v[3] *= dgelu_f((float)a[3]);
                    }
                    if (epi & GV_EPI_RESID) { f32x4 r = *(const f32x4*)(g.resid + orow * g.ldr + n); v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3]; }
                    if (epi & GV_EPI_POS) { f32x4 r = *(const f32x4*)(g.pos + (long)prow * g.N + n); v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3]; }
                    OutT* dst = C + orow * g.ldc + n;
                    if constexpr (sizeof(OutT) == 4) {
                        if (epi & GV_EPI_ACCUM) { f32x4 c = *(const f32x4*)dst; v[0] += c[0]; v[1] += c[1]; v[2] += c[2]; v[3] += c[3]; }
                    }
                    store4<OutT>(dst, v);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (n + r >= g.N) break;
                        float x = v[r];
                        if (epi & GV_EPI_BIAS) x += g.bias[n + r];
                        if (epi & GV_EPI_SAVE_PRE) g.aux_out[orow * g.ld_aux + n + r] = (bf16)x;
                        if (epi & GV_EPI_GELU) x = gelu_f(x);
                        if (epi & GV_EPI_DGELU) x *= dgelu_f((float)g.aux_in[orow * g.ld_aux + n + r]);
                        if (epi & GV_EPI_RESID) x += g.resid[orow * g.ldr + n + r];
                        if (epi & GV_EPI_POS) x += g.pos[(long)prow * g.N + n + r];
                        OutT* dst = C + orow * g.ldc + n + r;
                        if constexpr (sizeof(OutT) == 4) { if (epi & GV_EPI_ACCUM) x += *dst; }
                        *dst = (OutT)x;
                    }
                }
            }
        }
    }
}

template <bool TA, bool TB, typename OutT, bool ATOMIC>
int launch(const GemmP& p, hipStream_t s) {
    auto kern = gemm_kernel<TA, TB, OutT, ATOMIC>;
    static bool attr_done = false;   // 64 KiB dynamic LDS needs the opt-in once per kernel
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) { gv_set_error("gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    dim3 grid(p.tiles_m * p.tiles_n, p.ksplit, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, s, p);
    GV_LAUNCH_CHECK("gv_linear");
    return GV_OK;
}

}  // namespace

extern "C" int gv_linear(const gv_linear_args* a, void* stream) {
    GV_REQUIRE(a && a->A && a->B && a->C, GV_E_NULL, "gv_linear: null operand");
    GV_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, GV_E_SHAPE, "gv_linear: M,N,K must be > 0 (got %d,%d,%d)", a->M, a->N, a->K);
    const bool ta = a->trans_a != 0, tb = a->trans_b != 0;
    GV_REQUIRE(!(ta && !tb), GV_E_UNSUPPORTED, "gv_linear: (trans_a=1, trans_b=0) is not built");
    if (!ta || !tb)
        GV_REQUIRE(a->K % 64 == 0, GV_E_SHAPE, "gv_linear: K=%d must be a multiple of 64 when an operand is k-contiguous", a->K);
    if (ta) GV_REQUIRE(a->M % 8 == 0, GV_E_SHAPE, "gv_linear: M=%d must be a multiple of 8 with trans_a", a->M);
    if (tb) GV_REQUIRE(a->N % 8 == 0, GV_E_SHAPE, "gv_linear: N=%d must be a multiple of 8 with trans_b", a->N);
    GV_REQUIRE(a->lda % 8 == 0 && a->ldb % 8 == 0, GV_E_ALIGN, "gv_linear: lda/ldb must be multiples of 8 elements");
    GV_REQUIRE(gv_aligned(a->A, 16) && gv_aligned(a->B, 16) && gv_aligned(a->C, 16), GV_E_ALIGN, "gv_linear: A/B/C must be 16-byte aligned");
    const int e = a->epilogue;
    if (e & GV_EPI_BIAS) GV_REQUIRE(a->bias && gv_aligned(a->bias, 16), GV_E_NULL, "gv_linear: BIAS needs an aligned bias");
    if (e & GV_EPI_RESID) GV_REQUIRE(a->resid && a->ldr % 4 == 0, GV_E_NULL, "gv_linear: RESID needs resid with ldr %% 4 == 0");
    if (e & GV_EPI_DGELU) GV_REQUIRE(a->aux_in && a->ld_aux % 4 == 0, GV_E_NULL, "gv_linear: DGELU needs aux_in");
    if (e & GV_EPI_SAVE_PRE) GV_REQUIRE(a->aux_out && a->ld_aux % 4 == 0, GV_E_NULL, "gv_linear: SAVE_PRE needs aux_out");
    if (e & GV_EPI_POS) GV_REQUIRE(a->pos && a->P > 0, GV_E_NULL, "gv_linear: POS needs pos and P");
    if (e & GV_EPI_ACCUM) GV_REQUIRE(a->c_is_f32, GV_E_UNSUPPORTED, "gv_linear: ACCUM needs an f32 C");
    GV_REQUIRE(a->ldc % 4 == 0, GV_E_ALIGN, "gv_linear: ldc must be a multiple of 4");

    GemmP p;
    p.A = (const bf16*)a->A; p.B = (const bf16*)a->B; p.C = a->C;
    p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc;
    p.epi = e; p.bias = a->bias; p.resid = a->resid; p.ldr = a->ldr;
    p.aux_in = (const bf16*)a->aux_in; p.ld_aux = a->ld_aux; p.aux_out = (bf16*)a->aux_out;
    p.pos = a->pos; p.P = a->P; p.alpha = a->alpha == 0.f ? 1.f : a->alpha;
    p.tiles_m = (a->M + BM - 1) / BM; p.tiles_n = (a->N + BN - 1) / BN;
    p.ksplit = 1; p.k_per_split = ((a->K + BK - 1) / BK) * BK;
    hipStream_t s = (hipStream_t)stream;

    // Split K when the output grid alone cannot fill 256 CUs (dW: reduction over tens of
    // thousands of tokens; the 65536-class DINO head's dX).  Partial sums meet in C through
    // f32 atomics, which needs pure ACCUM semantics (C already holds the value to add to).
    {
        const int tiles = p.tiles_m * p.tiles_n;
        const bool only_accum = a->c_is_f32 && e == GV_EPI_ACCUM;
        if (only_accum && tiles < 384) {
            const int want = (512 + tiles - 1) / tiles;
            const int ksteps = (a->K + BK - 1) / BK;
            const int maxs = ksteps / 8 > 0 ? ksteps / 8 : 1;   // at least 8 k-steps (512 deep) per slice
            const int S = want < maxs ? want : maxs;
            if (S > 1) {
                const int per = (ksteps + S - 1) / S;
                p.k_per_split = per * BK;
                p.ksplit = (ksteps + per - 1) / per;
            }
        }
    }
    if (ta && tb) {
        if (p.ksplit > 1) return launch<true, true, float, true>(p, s);
        return a->c_is_f32 ? launch<true, true, float, false>(p, s) : launch<true, true, bf16, false>(p, s);
    }
    if (!ta && tb) {
        if (p.ksplit > 1) return launch<false, true, float, true>(p, s);
        return a->c_is_f32 ? launch<false, true, float, false>(p, s) : launch<false, true, bf16, false>(p, s);
    }
    if (p.ksplit > 1) return launch<false, false, float, true>(p, s);
    return a->c_is_f32 ? launch<false, false, float, false>(p, s) : launch<false, false, bf16, false>(p, s);
}
