// Geometry-parametrised core of the bf16 MFMA GEMM (see gemm.hip for the design notes).
// Shared by the production library (gemm.hip) and the tuning lab (tools/gemm_lab.hip).
#pragma once
#include "gv_common.h"

// Tuning-lab switches (ablation bits 20-22 of GemmP::epi, s_memtime stamps, experimental k-loop schedules 10 / 20)
// exist only in builds made with -DGV_GEMM_LAB (tools/gemm_lab.hip); the production library compiles them out and
// gv_linear rejects any epilogue bit outside the documented mask.
#ifdef GV_GEMM_LAB
#define GV_LAB_BIT(g, bit) (((g).epi & (1 << (bit))) != 0)
#else
#define GV_LAB_BIT(g, bit) false
#endif

namespace gvgemm {

static __device__ __attribute__((aligned(256))) unsigned short zero_page[128];   // 256 B of zeros (one copy per translation unit)

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
    float* slab;       // split-K: non-null -> partial tiles are stored here [slice][M][N] instead of atomics
    float* colsum_a;   // TA only: += column sums of A (bias gradient), computed as MFMAs against ones
    const float* row_scale;   // RESID only: v *= row_scale[m] before the residual add (stochastic depth); null = 1
    int order;   // 0: flat m-major items; 1: per-XCD M-panel ranges walked n-major (L2 reuse of A)
};

// BM x BN output tile, BK reduction depth per ring stage, WM x WN waves, NSTAGE ring stages
// SCHED: where the next stage's LDS-DMA pieces are issued inside a k-step --
//   0 before the fragment reads, 1 after the reads, 2 between the MFMA k-halves, 3 one piece per MFMA row group
template <int BM_, int BN_, int BK_, int WM_, int WN_, int NSTAGE_, int SCHED_ = 0>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, NSTAGE = NSTAGE_, SCHED = SCHED_;
    static constexpr int NW = WM * WN, THREADS = NW * 64;
    static constexpr int FM = BM / WM / 16, FN = BN / WN / 16;      // 16x16 fragments per wave
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    static constexpr int LDS = NSTAGE * STAGE;
    static constexpr int KS = BK / 32;                               // MFMA k-steps per stage
    static constexpr int A_PPW = A_BYTES / 1024 / NW, B_PPW = B_BYTES / 1024 / NW;   // pieces per wave
    static constexpr int GLDS = A_PPW + B_PPW;                       // LDS-DMA instrs per wave per step
    static constexpr int PD = NSTAGE - 1;
    static_assert(A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0, "pieces must divide over the waves");
    static_assert(BK == 32 || BK == 64, "BK must be 32 or 64");
};

__device__ __forceinline__ int swz_t(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
template <int BK> __device__ __forceinline__ int swz_n(int r) {
    if constexpr (BK == 32) return ((r >> 3) & 1) << 1; else return r & 7;
}

// LDS-DMA of 16 B per lane: LDS[dst + lane*16] <- *src (per-lane global address).  Issued from
// inline asm ON PURPOSE: with the builtin, hipcc's waitcnt pass treats the DMA as a pending LDS
// write and puts `s_waitcnt vmcnt(0)` in front of the next ds_read, i.e. right after the
// prefetch was issued -- the whole ring then runs load -> wait -> compute serially (measured:
// every tile geometry stuck at ~20 % of MFMA peak).  From asm the compiler's scoreboard never
// sees it; the k-loop's hand-counted s_waitcnt vmcnt(N) + s_barrier order the ds_reads.
// M0 (the DMA's LDS base) is compiler-reserved: save / set / restore inside one statement.
__device__ __forceinline__ void glds16(const void* src, GV_LDS char* dst) {
    unsigned keep;
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_addr)
                 : "memory");
}

// LDS-DMA of 16 B per lane with the global address split as  uniform 64-bit base (SGPRs) + per-lane 32-bit byte offset (VGPR)
// + immediate: the k-loop advances only the scalar base, the issuing wave spends no vector instruction on addresses (its SIMD
// partner is inside an MFMA cluster at raised priority: every VALU instruction of the load segment waits for a free issue slot)
// (the instruction's immediate offset is added to the LDS address as well as to the global one, as for MUBUF LDS loads:
// M0 is set IMM short of the destination)
template <int IMM, bool NT = false>      // NT: nontemporal hint for operands this step never reads again
__device__ __forceinline__ void glds16_s(unsigned long long sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    lds_dst -= IMM;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(IMM)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(IMM)
                     : "memory");
}

// Per-lane source pointers of one operand tile (one per LDS-DMA piece this wave issues), set
// up once per work item so the k-loop only adds the k offset.  natural: [R rows][BK k] image,
// pointer = row start + swizzled 16-B chunk; transposed: [BK k][R cols] image, pointer = column
// position in reduction row r (r = this lane's row inside a stage), `ok` = column in range.
template <bool T, int R, int BK, int NW>
struct TileSrc {
    static constexpr int PPW = R * BK * 2 / 1024 / NW;
    const bf16* ptr[PPW];
    bool ok[PPW];
    int r[PPW];
    bool all_ok;          // wave-uniform: every lane's column chunk of every piece lies inside the operand

    __device__ __forceinline__ void setup(const bf16* __restrict__ base, long ld, int o0, int lim, int wave, int lane) {
        if constexpr (!T) {
            constexpr int RB = BK * 2, RPP = 1024 / RB, LPR = RB / 16;
#pragma unroll
            for (int p = 0; p < PPW; ++p) {
                const int rr = (wave * PPW + p) * RPP + lane / LPR;
                const int c = (lane % LPR) ^ swz_n<BK>(rr);
                int row = o0 + rr;
                row = row < lim ? row : lim - 1;
                ptr[p] = base + (long)row * ld + c * 8;
                ok[p] = true; r[p] = 0;
            }
            all_ok = true;
        } else {
            constexpr int RB = R * 2, RPP = 1024 / RB, LPR = RB / 16;
#pragma unroll
            for (int p = 0; p < PPW; ++p) {
                const int rr = (wave * PPW + p) * RPP + lane / LPR;
                const int s16 = lane % LPR;
                const int logical = (s16 >> 1) ^ swz_t(rr);
                const int col = o0 + logical * 16 + (s16 & 1) * 8;
                ok[p] = col + 8 <= lim;
                r[p] = rr;
                ptr[p] = ok[p] ? base + (long)rr * ld + col : (const bf16*)zero_page + (s16 & 15) * 8;
            }
            bool every = true;
#pragma unroll
            for (int p = 0; p < PPW; ++p) every = every && ok[p];
            all_ok = __builtin_amdgcn_ballot_w64(!every) == 0ull;
        }
    }
    // issue piece p of this wave's share of the k-step starting at reduction index k0
    __device__ __forceinline__ void issue_one(int p, long ld, int k0, int klim, GV_LDS char* tile, int wave) const {
        const bf16* src;
        if constexpr (!T) src = ptr[p] + k0;
        else src = (ok[p] && k0 + r[p] < klim) ? ptr[p] + (long)k0 * ld : (const bf16*)zero_page;
        glds16(src, tile + (wave * PPW + p) * 1024);
    }
    __device__ __forceinline__ void issue(long ld, int k0, int klim, GV_LDS char* tile, int wave) const {
        if constexpr (T) {
            // whole step inside the reduction range and every column of this wave's pieces valid (the hot
            // shapes, always): no per-lane predicate -- the compare / select per piece are VALU work that
            // competes with the partner wave's MFMA issue
            if (all_ok && k0 + BK <= klim) {
                const long off = (long)k0 * ld;
#pragma unroll
                for (int p = 0; p < PPW; ++p) glds16(ptr[p] + off, tile + (wave * PPW + p) * 1024);
                return;
            }
        }
#pragma unroll
        for (int p = 0; p < PPW; ++p) issue_one(p, ld, k0, klim, tile, wave);
    }
};

// 8 bf16 along the reduction index for one output row/col of 16-wide block blk16, MFMA k-step ks
template <bool T, int R, int BK>
__device__ __forceinline__ bf16x8 read_frag(GV_LDS char* tile, int blk16, int ks, int lane) {
    if constexpr (!T) {
        constexpr int RB = BK * 2;
        const int row = blk16 * 16 + (lane & 15);
        const int chunk = ks * 4 + (lane >> 4);
        return *(GV_LDS bf16x8*)(tile + row * RB + ((chunk ^ swz_n<BK>(row)) << 4));
    } else {
        constexpr int RB = R * 2;
        const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
        const int kr = ks * 32 + g * 8 + q;
        GV_LDS char* a0 = tile + kr * RB + ((blk16 ^ swz_t(kr)) << 5) + p * 8;
        bf16x4 lo = GV_DS_READ_TR16(a0);
        bf16x4 hi = GV_DS_READ_TR16((a0 + 4 * RB));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <typename OutT> __device__ __forceinline__ void store4(OutT* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) { *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]}; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const float* v) {
    *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}

struct Item { int m0, n0, kbeg, kend, nt; };

// Work list of one workgroup.  order 0: item idx -> (slice, tile m-major), workgroups start at
// an XCD-contiguous `first` and stride by the grid.  order 1 (ksplit == 1, tiles_m >= 16): the
// 8 XCDs (blocks b, b+8, .. share one) each own a contiguous range of M panels, sized to sit in
// that XCD's 4 MiB L2, and walk it N-MAJOR: the first sweep over the range streams the A panels
// from HBM, every later sweep (next 128 output columns) re-reads them from L2.
struct Walk {
    int first, stride, count;      // local item indices first, first+stride, ..
    int mlo, mcnt;                 // order 1: this XCD's M-panel range
};

template <class C>
__device__ __forceinline__ Walk make_walk(const GemmP& g) {
    const int G = gridDim.x, xcd = blockIdx.x & 7, lw = blockIdx.x >> 3;
    Walk w;
    if (g.order == 1) {
        const int q = g.tiles_m >> 3, r = g.tiles_m & 7;
        w.mlo = xcd * q + (xcd < r ? xcd : r);
        w.mcnt = q + (xcd < r ? 1 : 0);
        const int gx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);     // workgroups on this XCD
        const int n_local = w.mcnt * g.tiles_n;
        w.first = lw; w.stride = gx;
        w.count = lw < n_local ? (n_local - lw + gx - 1) / gx : 0;
    } else {
        const int n_items = g.tiles_m * g.tiles_n * g.ksplit;
        const int q = G >> 3, r = G & 7;
        w.first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + lw;
        w.stride = G;
        w.count = w.first < n_items ? (n_items - w.first + G - 1) / G : 0;
        w.mlo = 0; w.mcnt = g.tiles_m;
    }
    return w;
}

template <class C>
__device__ __forceinline__ Item make_item(const GemmP& g, const Walk& w, int idx) {
    Item it;
    if (g.order == 1) {
        const int n = idx / w.mcnt, m = w.mlo + (idx - n * w.mcnt);
        it.m0 = m * C::BM; it.n0 = n * C::BN; it.kbeg = 0; it.kend = g.K;
    } else {
        const int tiles = g.tiles_m * g.tiles_n;
        const int slice = idx / tiles;
        const int t = idx - slice * tiles;
        it.m0 = (t / g.tiles_n) * C::BM;
        it.n0 = (t % g.tiles_n) * C::BN;
        it.kbeg = slice * g.k_per_split;
        it.kend = min(g.K, it.kbeg + g.k_per_split);
    }
    it.nt = (it.kend - it.kbeg + C::BK - 1) / C::BK;
    return it;
}

#if defined(GV_GEMM_STAMPS) && !defined(GV_GEMM_LAB)
#error "GV_GEMM_STAMPS is a tuning-lab option: build with -DGV_GEMM_LAB"
#endif
#ifdef GV_GEMM_STAMPS
__device__ __forceinline__ unsigned long long gv_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define GV_STAMP(var) const unsigned long long var = gv_stamp()
#else
#define GV_STAMP(var)
#endif

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename OutT> struct OutVec;
template <> struct OutVec<float> { static constexpr int W = 4; };   // 16 B per lane
template <> struct OutVec<bf16> { static constexpr int W = 8; };    // 16 B per lane

// EPI >= 0: compile-time epilogue mask (host guarantees N % 8 == 0); EPI < 0: runtime mask.
// `wk`: the work list of this workgroup (make_walk for a plain launch; a grouped launch hands every workgroup the one
// item of the problem it belongs to)
template <class C, bool TA, bool TB, typename OutT, bool ATOMIC, int EPI>
__device__ __forceinline__ void gemm_body_w(const GemmP& g, GV_LDS char* smem, const Walk wk) {
    constexpr int BM = C::BM, BN = C::BN, BK = C::BK, FM = C::FM, FN = C::FN, NSTAGE = C::NSTAGE, PD = C::PD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int li16 = lane & 15, gq = lane >> 4;
#ifdef GV_GEMM_STAMPS
    GV_STAMP(t_kernel0);
#endif

    // SCHED 20 (lab): a workgroup that walks several items issues the NEXT item's first ring stage before the
    // epilogue of the current one (the epilogue image then lives in the other stage only)
    #ifdef GV_GEMM_LAB
    constexpr bool XPF = C::SCHED == 20 && C::NSTAGE == 2 && !ATOMIC;
#else
    static_assert(C::SCHED != 10 && C::SCHED != 20, "k-loop schedules 10 / 20 are tuning-lab builds (-DGV_GEMM_LAB)");
    constexpr bool XPF = false;
#endif
    bool prefetched = false;
    for (int it_i = 0, idx = wk.first; it_i < wk.count; ++it_i, idx += wk.stride) {
        const Item it = make_item<C>(g, wk, idx);
        TileSrc<TA, BM, BK, C::NW> srcA;
        TileSrc<TB, BN, BK, C::NW> srcB;
        srcA.setup(g.A, g.lda, it.m0, g.M, wave, lane);
        srcB.setup(g.B, g.ldb, it.n0, g.N, wave, lane);
        // ---- the LDS-DMA ring runs PD k-steps ahead of the MFMAs
        int l_k = 0, l_stage = 0;
        auto issue = [&]() {
            if (l_k < it.nt && !GV_LAB_BIT(g, 21)) {
                GV_LDS char* st = smem + l_stage * C::STAGE;
                const int k0 = it.kbeg + l_k * BK;
                srcA.issue(g.lda, k0, it.kend, st, wave);
                srcB.issue(g.ldb, k0, it.kend, st + C::A_BYTES, wave);
            }
            ++l_k;
            l_stage = (l_stage + 1 == NSTAGE) ? 0 : l_stage + 1;
        };
        // the same, one piece at a time (SCHED >= 2 spreads a step's pieces between its MFMAs)
        bool p_live = false; GV_LDS char* p_st = smem; int p_k0 = 0;
        auto issue_begin = [&]() {
            p_live = l_k < it.nt && !GV_LAB_BIT(g, 21);
            p_st = smem + l_stage * C::STAGE; p_k0 = it.kbeg + l_k * BK;
            ++l_k;
            l_stage = (l_stage + 1 == NSTAGE) ? 0 : l_stage + 1;
        };
        auto issue_piece = [&](int p) {
            if (!p_live) return;
            if (p < C::A_PPW) srcA.issue_one(p, g.lda, p_k0, it.kend, p_st, wave);
            else srcB.issue_one(p - C::A_PPW, g.ldb, p_k0, it.kend, p_st + C::A_BYTES, wave);
        };
        if (XPF && prefetched) { l_k = 1; l_stage = 1; }      // stage 0 was issued during the previous item's epilogue
        else {
#pragma unroll
            for (int s = 0; s < PD; ++s) issue();
        }

        // ---- epilogue geometry (see below) and EARLY PREFETCH of its global operands: the
        // residual / pos / accumulate-into values and the saved pre-activation are loaded into
        // registers now, so their HBM latency hides under the whole k-loop instead of being
        // paid once per row block after it.
        const int m0 = it.m0 + wm * FM * 16, n0 = it.n0 + wn * FN * 16;
        constexpr int IW = FN * 16;                     // image width (columns of this wave)
        constexpr int STRIDE = IW + 4;                  // f32 row stride, +4 breaks bank conflicts
        constexpr int IMG_BYTES = (XPF ? C::STAGE : C::LDS) / C::NW;      // this wave's share of the epilogue image space
        constexpr int ROWS_FIT = IMG_BYTES / (STRIDE * 4);
        constexpr int IB = ROWS_FIT >= FM * 16 ? FM : (ROWS_FIT >= 32 && FM % 2 == 0 ? 2 : 1);   // 16-row blocks per pass
        static_assert(16 * STRIDE * 4 <= IMG_BYTES, "one 16-row block of the image must fit the wave's share");
        constexpr int W = ATOMIC ? 1 : OutVec<OutT>::W;     // columns per lane in the row pass
        constexpr int LPR = IW / W;                          // lanes per image row
        constexpr int RPI = LPR >= 64 ? 1 : 64 / LPR;        // image rows per wave-instruction
        constexpr int CPI = LPR > 64 ? LPR / 64 : 1;         // column chunks when a row needs > 64 lanes
        constexpr int NIT = FM * 16 / RPI * CPI;             // row-pass iterations per tile
        const int lrow = LPR >= 64 ? 0 : lane / LPR;
        const int lcol = (LPR >= 64 ? lane : lane % LPR) * W;
        const int epi = EPI >= 0 ? EPI : g.epi;
        const int N = g.N, M = g.M;
        constexpr bool PREFETCH = !ATOMIC && EPI >= 0 && (EPI & (GV_EPI_RESID | GV_EPI_POS | GV_EPI_ACCUM | GV_EPI_DGELU)) != 0 && FM <= 4 && C::SCHED != 10;
        constexpr bool BIAS_EARLY = !ATOMIC && EPI >= 0 && (EPI & GV_EPI_BIAS) != 0 && CPI == 1;
        f32x4 pre_b[W / 4 > 0 ? W / 4 : 1];
        if constexpr (BIAS_EARLY) {
            const int nb = n0 + lcol < N ? n0 + lcol : N - W;
#pragma unroll
            for (int q = 0; q < W; q += 4) pre_b[q / 4] = *(const f32x4*)(g.bias + nb + q);
        }
        f32x4 pre_r[PREFETCH ? NIT : 1][W / 4 > 0 ? W / 4 : 1];
        bf16x8 pre_a[PREFETCH && (EPI >= 0 && (EPI & GV_EPI_DGELU)) ? NIT : 1];
        if constexpr (PREFETCH) {
#pragma unroll
            for (int itn = 0; itn < NIT; ++itn) {
                const int row = (itn / CPI) * RPI + lrow, col = lcol + (itn % CPI) * 64 * W;
                const int m = m0 + row, n = n0 + col;
                const int mc = m < M ? m : M - 1, nc = n < N ? n : N - W;
                long orow = mc; int prow = 0;
                if (EPI & GV_EPI_POS) { orow = mc + mc / g.P + 1; prow = (mc % g.P) + 1; }
#pragma unroll
                for (int q = 0; q < W; q += 4) {
                    f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (EPI & GV_EPI_RESID) r += *(const f32x4*)(g.resid + orow * g.ldr + nc + q);
                    if (EPI & GV_EPI_POS) r += *(const f32x4*)(g.pos + (long)prow * N + nc + q);
                    if constexpr (sizeof(OutT) == 4) { if (EPI & GV_EPI_ACCUM) r += *(const f32x4*)((const float*)g.C + orow * g.ldc + nc + q); }
                    pre_r[itn][q / 4] = r;
                }
                if constexpr ((EPI & GV_EPI_DGELU) != 0) {
                    if constexpr (W == 8) pre_a[itn] = *(const bf16x8*)(g.aux_in + orow * g.ld_aux + nc);
                    else { const bf16x4 a = *(const bf16x4*)(g.aux_in + orow * g.ld_aux + nc); pre_a[itn] = bf16x8{a[0], a[1], a[2], a[3], a[0], a[1], a[2], a[3]}; }
                }
            }
        }

        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // bias gradient fused into the dW product: D[n][m] += ones[n][k] A[m][k] gives every row n the
        // column sum of A over this item's k range; one extra MFMA per A fragment, no VALU, and dY
        // is not read a second time.  Done by the wn == 0 waves of the n-tile-0 workgroups.
        const bool do_colsum = TA && g.colsum_a != nullptr && it.n0 == 0 && wn == 0;
        f32x4 csum[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) csum[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16x8 ones = bf16x8{(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};

#ifdef GV_GEMM_STAMPS
        unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_comp = 0;
        GV_STAMP(t_loop0);
#endif
        int c_stage = 0;
#ifdef GV_GEMM_LAB
        if constexpr (C::SCHED == 10) {
            // ---- PING-PONG k-loop (lab): the workgroup's two halves of wave rows (SIMD partners: waves w and w + NW/2)
            // alternate roles every half step: while one group issues the LDS-DMA of step t + PD and reads its
            // fragments of step t, the other runs the MFMAs of the step it read before.  Two barriers per step;
            // before the second one every wave has waited for its pieces of step t + 1 (counted vmcnt) and for
            // its fragment reads (lgkmcnt(0)), so the next reader finds the stage landed and the next LDS-DMA
            // finds its target stage (last read one step ago) free.
            static_assert(C::WM % 2 == 0 && C::KS == 1 && PD >= 2, "ping-pong: an even number of wave rows, BK = 32");
            const int grp = wm / (C::WM / 2);
            bf16x8 pa[FM], pb[FN];
            auto reads = [&](int stage) {
                GV_LDS char* cur = smem + stage * C::STAGE;
#pragma unroll
                for (int j = 0; j < FN; ++j) pb[j] = read_frag<TB, BN, BK>(cur + C::A_BYTES, wn * FN + j, 0, lane);
#pragma unroll
                for (int i = 0; i < FM; ++i) pa[i] = read_frag<TA, BM, BK>(cur, wm * FM + i, 0, lane);
            };
            auto mfmas = [&]() {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = GV_MFMA_16x16x32(pb[j], pa[i], acc[i][j]);
            };
            auto wait_young = [&](int young) {
                if (young >= 2) wait_vmcnt<C::GLDS * 2>();
                else if (young == 1) wait_vmcnt<C::GLDS>();
                else wait_vmcnt<0>();
            };
            wait_young(min(PD, it.nt) - 1);                  // step 0 landed (mine)
            __builtin_amdgcn_s_barrier();
            // one code path for both halves; the second half runs it one barrier late (and the first half
            // takes one extra barrier at the end), so its memory phase lines up with the other's MFMA phase
            if (grp == 1) __builtin_amdgcn_s_barrier();
            for (int t = 0; t < it.nt; ++t) {
                issue();
                reads(c_stage);
                __builtin_amdgcn_s_waitcnt(0xC07F);              // fragments in registers before the barrier
                wait_young(max(0, min(PD - 1, it.nt - 2 - t)));   // my pieces of step t + 1 landed; t + 2.. may fly
                // sched_barrier: hipcc moves register-only MFMAs across s_barrier / s_waitcnt otherwise, which
                // would put both halves' MFMAs into the same phase
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                mfmas();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                c_stage = (c_stage + 1 == NSTAGE) ? 0 : c_stage + 1;
            }
            if (grp == 0) __builtin_amdgcn_s_barrier();
        } else
#endif
        for (int t = 0; t < it.nt; ++t) {
            GV_STAMP(ts0);
            // this step's pieces (mine) landed: everything but the younger in-flight steps
            {
                const int young = min(PD - 1, it.nt - 1 - t);
                if (PD >= 4 && young >= 3) wait_vmcnt<C::GLDS * 3>();
                else if (PD >= 3 && young == 2) wait_vmcnt<C::GLDS * 2>();
                else if (PD >= 2 && young == 1) wait_vmcnt<C::GLDS * 1>();
                else wait_vmcnt<0>();
            }
            GV_STAMP(ts1);
            __builtin_amdgcn_s_barrier();     // everybody's pieces landed; last step's stage is free
            GV_STAMP(ts2);
            if constexpr (C::SCHED == 0 || C::SCHED == 20) issue();
            GV_STAMP(ts3);
            GV_LDS char* cur = smem + c_stage * C::STAGE;
            // all fragment reads of the stage are issued up front (KS * (FM + FN) ds_read_b128 /
            // tr reads); the MFMAs then run back-to-back behind the compiler's counted lgkmcnt(N)
            // instead of exposing the LDS latency once per small read group.
            bf16x8 fa[C::KS][FM], fb[C::KS][FN];
            if (GV_LAB_BIT(g, 22)) {     // lab ablation: fragments from registers, no LDS reads
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) {
#pragma unroll
                    for (int j = 0; j < FN; ++j) { fb[ks][j] = fb[0][0]; asm volatile("" : "+v"(fb[ks][j])); }
#pragma unroll
                    for (int i = 0; i < FM; ++i) { fa[ks][i] = fb[0][0]; asm volatile("" : "+v"(fa[ks][i])); }
                }
            } else
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
#pragma unroll
                for (int j = 0; j < FN; ++j) fb[ks][j] = read_frag<TB, BN, BK>(cur + C::A_BYTES, wn * FN + j, ks, lane);
#pragma unroll
                for (int i = 0; i < FM; ++i) fa[ks][i] = read_frag<TA, BM, BK>(cur, wm * FM + i, ks, lane);
            }
#ifdef GV_GEMM_PIN_READS
            __builtin_amdgcn_sched_barrier(0);   // keep every read ahead of the first MFMA
#endif
            if constexpr (C::SCHED == 1) issue();
            if constexpr (C::SCHED == 2 || C::SCHED == 3) issue_begin();
            constexpr int GROUPS = C::KS * FM;                     // MFMA row groups of FN MFMAs each
            constexpr int PER = (C::GLDS + GROUPS - 1) / GROUPS;   // SCHED 3: pieces per group
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
#pragma unroll
                for (int i = 0; i < FM; ++i) {
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = GV_MFMA_16x16x32(fb[ks][j], fa[ks][i], acc[i][j]);
                    if constexpr (C::SCHED == 3) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < PER; ++q) { const int pc = (ks * FM + i) * PER + q; if (pc < C::GLDS) issue_piece(pc); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (C::SCHED == 2) {
                    if (ks == 0) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int pc = 0; pc < C::GLDS; ++pc) issue_piece(pc);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (TA) {
                if (do_colsum) {
#pragma unroll
                    for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
                        for (int i = 0; i < FM; ++i) csum[i] = GV_MFMA_16x16x32(ones, fa[ks][i], csum[i]);
                }
            }
            c_stage = (c_stage + 1 == NSTAGE) ? 0 : c_stage + 1;
#ifdef GV_GEMM_STAMPS
            { GV_STAMP(ts4); c_wait += ts1 - ts0; c_bar += ts2 - ts1; c_issue += ts3 - ts2; c_comp += ts4 - ts3; }
#endif
        }
#ifdef GV_GEMM_STAMPS
        GV_STAMP(t_loop1);
#endif
        __builtin_amdgcn_s_barrier();   // every wave is done reading the ring: it is epilogue scratch now
        if constexpr (TA) {
            if (do_colsum && gq == 0) {       // csum[i][r]: column = lane&15 = m, every row identical
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int m = it.m0 + wm * FM * 16 + i * 16 + li16;
                    if (m < g.M) atomicAdd(g.colsum_a + m, csum[i][0]);
                }
            }
        }

        // ---- epilogue.  acc[i][j][r]: m = m0 + i*16 + (lane&15), n = n0 + j*16 + (lane>>4)*4 + r.
        // The accumulators go through a per-wave LDS image so that every lane ends up with
        // ROW-CONTIGUOUS columns: bias / residual / aux loads and all stores (and atomics) are
        // then whole 128..256-B row segments instead of 16 scattered 32-B pieces per instruction.
        if constexpr (XPF) {
            prefetched = false;
            if (it_i + 1 < wk.count) {       // every wave is past the ring (barrier above): stage 0 is free for the next item
                const Item nx = make_item<C>(g, wk, idx + wk.stride);
                TileSrc<TA, BM, BK, C::NW> nA;
                TileSrc<TB, BN, BK, C::NW> nB;
                nA.setup(g.A, g.lda, nx.m0, g.M, wave, lane);
                nB.setup(g.B, g.ldb, nx.n0, g.N, wave, lane);
                if (!GV_LAB_BIT(g, 21)) {
                    nA.issue(g.lda, nx.kbeg, nx.kend, smem, wave);
                    nB.issue(g.ldb, nx.kbeg, nx.kend, smem + C::A_BYTES, wave);
                }
                prefetched = true;
            }
        }
        GV_LDS float* img = (GV_LDS float*)(smem + (XPF ? C::STAGE : 0) + wave * IMG_BYTES);
        float* Cf = (float*)g.C;
        OutT* Cp = (OutT*)g.C;
        bool vec_path = true;
        if constexpr (EPI < 0 && !ATOMIC) vec_path = (N & 7) == 0;

        if (ATOMIC || vec_path) {
#pragma unroll
            for (int ib = 0; ib < FM; ib += IB) {
#pragma unroll
                for (int ii = 0; ii < IB; ++ii)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        f32x4 v = acc[ib + ii][j] * g.alpha;
                        *(GV_LDS f32x4*)(img + (ii * 16 + li16) * STRIDE + j * 16 + gq * 4) = v;
                    }
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): own writes visible to own reads
#pragma unroll
                for (int r0 = 0; r0 < IB * 16; r0 += RPI) {
#pragma unroll
                    for (int cc = 0; cc < CPI; ++cc) {
                        const int row = r0 + lrow, col = lcol + cc * 64 * W;
                        const int m = m0 + ib * 16 + row, n = n0 + col;
                        const int itn = ((ib * 16 + r0) / RPI) * CPI + cc;   // compile-time after unrolling
                        const bool ok = m < M && n < N;
                        if constexpr (ATOMIC) {
                            if (ok) {
                                if (g.slab) g.slab[((long)(it.kbeg / g.k_per_split) * M + m) * N + n] = img[row * STRIDE + col];
                                else atomicAdd(Cf + (long)m * g.ldc + n, img[row * STRIDE + col]);
                            }
                        } else {
                            const int mc = m < M ? m : M - 1, nc = n < N ? n : N - W;
                            long orow = mc; int prow = 0;
                            if (epi & GV_EPI_POS) { orow = mc + mc / g.P + 1; prow = (mc % g.P) + 1; }
                            float v[W];
#pragma unroll
                            for (int q = 0; q < W; q += 4) {
                                const f32x4 x = *(GV_LDS f32x4*)(img + row * STRIDE + col + q);
                                v[q] = x[0]; v[q + 1] = x[1]; v[q + 2] = x[2]; v[q + 3] = x[3];
                            }
                            if (epi & GV_EPI_BIAS) {
#pragma unroll
                                for (int q = 0; q < W; q += 4) {
                                    f32x4 b;
                                    if constexpr (BIAS_EARLY) b = pre_b[q / 4]; else b = *(const f32x4*)(g.bias + nc + q);
                                    v[q] += b[0]; v[q + 1] += b[1]; v[q + 2] += b[2]; v[q + 3] += b[3];
                                }
                            }
                            if (epi & GV_EPI_SAVE_PRE) {
                                if (ok) {
                                    if constexpr (W == 8) *(bf16x8*)(g.aux_out + orow * g.ld_aux + n) = bf16x8{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
                                    else store4<bf16>(g.aux_out + orow * g.ld_aux + n, v);
                                }
                            }
                            if (epi & GV_EPI_GELU) {
#pragma unroll
                                for (int q = 0; q < W; ++q) v[q] = gelu_f(v[q]);
                            }
                            if (epi & GV_EPI_DGELU) {
                                bf16x8 a;
                                if constexpr (PREFETCH) a = pre_a[(EPI >= 0 && (EPI & GV_EPI_DGELU)) ? itn : 0];
                                else if constexpr (W == 8) a = *(const bf16x8*)(g.aux_in + orow * g.ld_aux + nc);
                                else { const bf16x4 a4 = *(const bf16x4*)(g.aux_in + orow * g.ld_aux + nc); a = bf16x8{a4[0], a4[1], a4[2], a4[3], a4[0], a4[1], a4[2], a4[3]}; }
#pragma unroll
                                for (int q = 0; q < W; ++q) v[q] *= dgelu_f((float)a[q]);
                            }
                            if ((epi & GV_EPI_RESID) && g.row_scale) {
                                const float rs = g.row_scale[mc];
#pragma unroll
                                for (int q = 0; q < W; ++q) v[q] *= rs;
                            }
#pragma unroll
                            for (int q = 0; q < W; q += 4) {
                                f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
                                if constexpr (PREFETCH) r = pre_r[itn][q / 4];
                                else {
                                    if (epi & GV_EPI_RESID) r += *(const f32x4*)(g.resid + orow * g.ldr + nc + q);
                                    if (epi & GV_EPI_POS) r += *(const f32x4*)(g.pos + (long)prow * N + nc + q);
                                    if constexpr (sizeof(OutT) == 4) { if (epi & GV_EPI_ACCUM) r += *(const f32x4*)(Cf + orow * g.ldc + nc + q); }
                                }
                                v[q] += r[0]; v[q + 1] += r[1]; v[q + 2] += r[2]; v[q + 3] += r[3];
                            }
                            if (GV_LAB_BIT(g, 20)) { asm volatile("" ::"v"(v[0]), "v"(v[W - 1])); }   // lab ablation: no store
                            else if (ok) {
                                if constexpr (W == 8) *(bf16x8*)(Cp + orow * g.ldc + n) = bf16x8{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
                                else store4<OutT>(Cp + orow * g.ldc + n, v);
                            }
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);   // image reads done before the next row block overwrites
            }
        } else if constexpr (EPI < 0 && !ATOMIC) {
            // ragged N (tiny heads): scalar, guarded, runtime flags -- correctness path only,
            // compiled into the generic build alone (its dynamic acc indexing costs scratch)
            const int e2 = g.epi;
#pragma unroll 1
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + i * 16 + li16;
                if (m >= M) continue;
                long orow = m; int prow = 0;
                if (e2 & GV_EPI_POS) { orow = m + m / g.P + 1; prow = (m % g.P) + 1; }
#pragma unroll 1
                for (int j = 0; j < FN; ++j) {
                    const int n = n0 + j * 16 + gq * 4;
                    for (int r = 0; r < 4; ++r) {
                        if (n + r >= N) break;
                        float x = acc[i][j][r] * g.alpha;
                        if (e2 & GV_EPI_BIAS) x += g.bias[n + r];
                        if (e2 & GV_EPI_SAVE_PRE) g.aux_out[orow * g.ld_aux + n + r] = (bf16)x;
                        if (e2 & GV_EPI_GELU) x = gelu_f(x);
                        if (e2 & GV_EPI_DGELU) x *= dgelu_f((float)g.aux_in[orow * g.ld_aux + n + r]);
                        if (e2 & GV_EPI_RESID) x = (g.row_scale ? x * g.row_scale[m] : x) + g.resid[orow * g.ldr + n + r];
                        if (e2 & GV_EPI_POS) x += g.pos[(long)prow * N + n + r];
                        OutT* dst = Cp + orow * g.ldc + n + r;
                        if constexpr (sizeof(OutT) == 4) { if (e2 & GV_EPI_ACCUM) x += *dst; }
                        *dst = (OutT)x;
                    }
                }
            }
        }
        // Retire this item's stores with a wait the COMPILER can see (otherwise it guards the
        // store-data registers it reuses in the next k-loop with its own vmcnt(0) per k-step),
        // and let every wave leave its image before the next item's LDS-DMA overwrites it.
        if (it_i + 1 < wk.count) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0); a wave that exits next needs no wait
#ifdef GV_GEMM_STAMPS
        {   // stamps leave through a buffer of their own (g.pos is unused by the lab shapes)
            GV_STAMP(t_end);
            if (lane == 0 && g.pos) {
                float* o = (float*)g.pos + ((long)blockIdx.x * C::NW + wave) * 8;
                o[0] = (float)c_wait; o[1] = (float)c_bar; o[2] = (float)c_issue; o[3] = (float)c_comp;
                o[4] = (float)(t_loop1 - t_loop0); o[5] = (float)(t_end - t_loop1); o[6] = (float)(t_loop0 - t_kernel0); o[7] = (float)it.nt;
            }
        }
#endif
        if (it_i + 1 < wk.count) __builtin_amdgcn_s_barrier();
    }
}

template <class C, bool TA, bool TB, typename OutT, bool ATOMIC, int EPI>
__device__ __forceinline__ void gemm_body(const GemmP& g, GV_LDS char* smem) {
    gemm_body_w<C, TA, TB, OutT, ATOMIC, EPI>(g, smem, make_walk<C>(g));
}

}  // namespace gvgemm
