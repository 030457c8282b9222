// Full-row "panel" GEMM for the N = 384 products of ViT-S with the LayerNorm that follows them fused into the
// epilogue -- gv_linear_ln_fwd / gv_linear_ln_bwd (include/gipvit.h).
//
// Replaces, per transformer block (vit.pyc@L146-152 Block.forward and its autograd),
//   forward :  x' = x + Linear(a)  (attn.proj / mlp.fc2, vit.pyc@L98-104, L119-131)  +  the NEXT LayerNorm
//              (norm2 / the next block's norm1, vit.pyc@L138,142): y = LN(x'), mean, rstd;
//   backward:  dXn = dY W  (the dX product of mlp.fc1 / attn.qkv)  +  the LayerNorm backward it feeds:
//              g += dLN/dx, gb = bf16(g), column sums for dgamma / dbeta / the previous Linear's bias gradient.
// In round 1 these were a 128x128-tile GEMM plus a separate LayerNorm pass; measured there (profiles/r01_*):
//   * 345 x 3 tiles on 512 workgroup slots = 2.02 rounds -> a third, nearly empty round (423-548 TFLOP/s);
//   * a 128x128x64 step moves 64 FLOP per byte through the CU's L2 -> LDS path, the bound of that kernel;
//   * the LayerNorm pass re-reads the f32 row the GEMM epilogue has just written (68 MB per call).
// Here ONE workgroup (8 waves, 1 per CU) owns BM = 16 * FM consecutive token rows and ALL 384 output columns:
//   * FM is chosen per launch so that ceil(M / BM) <= 256 * rounds with the last round full (M = 44 160 tokens ->
//     BM = 176, 251 workgroups, one round);
//   * a k-step stages (BM + 384) x 64 bf16 = 70 KB for 2 * BM * 384 * 64 FLOP = 121 FLOP per byte (BM = 176);
//   * the whole output row lives in one workgroup, so the row statistics of the following LayerNorm (forward) or the
//     row dot products of its backward are taken in the epilogue and the row never makes a second HBM round trip.
// Wave w owns the 16-column fragments {w, w + 8, w + 16} of every row block (the interleaving keeps a wave's
// fragments inside one 128-column block of a transposed weight image).  Staging, swizzles, fragment reads and the
// swapped MFMA roles are those of gemm_core.h (LDS-DMA from inline asm behind hand-counted vmcnt + raw s_barrier).
// The epilogue transposes the accumulators through a per-pass LDS image and then works one wave per row (full 512-B row
// segments for every load and store, the arithmetic of layernorm.hip's row kernels, wave sums by DPP); the global rows
// a pass needs are all requested before the pass's image barrier.
#include "gemm_core.h"
#include "timing.h"
#include <type_traits>

namespace {

using namespace gvgemm;

constexpr int PN = 384;              // output columns = the model width this kernel is built for (ViT-S)
constexpr int IMG_STRIDE = PN + 4;          // f32 image row stride (+4: conflict-free transposed writes)

// Geometry: NW = 8 waves, each 3 column fragments x FM row fragments, BK = 64, ONE workgroup per CU (ring 2 x (A + 48 KB)).
// (A 4-wave, BK = 32 variant with two workgroups per CU -- one workgroup's HBM-bound epilogue under the other's k-loop -- is
// expressible with the same template and was measured: 103 vs 87 us on the K = 1536 forward; not instantiated.)
template <int FM, int NW_, int BK_>
struct PC {
    static constexpr int NW = NW_, BK = BK_, KS = BK / 32, NF = 24 / NW;      // waves, k depth per stage, MFMA k-steps, column fragments per wave
    static constexpr int BM = FM * 16;
    static constexpr int RPP = 1024 / (BK * 2);             // rows per 1-KB piece of the A image
    static constexpr int A_PPW = (FM * 16 + RPP * NW - 1) / (RPP * NW);      // pieces per wave
    static constexpr int A_ROWS = A_PPW * RPP * NW;         // staged rows (>= BM; surplus rows are the next rows, clamped at M)
    static constexpr int A_BYTES = A_ROWS * BK * 2;
    static constexpr int W_BLOCK = 128 * BK * 2;            // one 128-column block of the weight stage
    static constexpr int STAGE = A_BYTES + 3 * W_BLOCK;
    static constexpr int LDS = 2 * STAGE;
    static constexpr int WG_PER_CU = NW == 8 ? 1 : 2;
    static constexpr int IB_FIT = LDS / (IMG_STRIDE * 4) / 16;
    // 16-row blocks per epilogue pass: bounded by the LDS image and by the registers that hold a pass's prefetched
    // global rows (forward: the residual row, 6 f32 per lane and row; backward: x and g rows, 12)
    static constexpr int ib(int want) { return want < (IB_FIT < FM ? IB_FIT : FM) ? want : (IB_FIT < FM ? IB_FIT : FM); }
    // MODE_WIDE: the image sits behind ring stage 0 (which holds the next column block's first k-step during the epilogue)
    static constexpr int ib_wide() { return FM < 3 ? FM : 3; }
    static constexpr int LDS_WIDE = (STAGE + ib_wide() * 16 * IMG_STRIDE * 4) > LDS ? (STAGE + ib_wide() * 16 * IMG_STRIDE * 4) : LDS;
    static_assert(LDS_WIDE <= 160 * 1024, "LDS budget (wide)");
    static_assert(IB_FIT >= 1 && LDS * WG_PER_CU <= 160 * 1024, "LDS budget");
    static_assert(NW * 3 * PN * 4 <= LDS, "column-sum reduction scratch must fit the ring");
};

// k-loop variants measured and dropped (round 2, M = 44 160, K = 1536, fused forward; tools/panel_probe.py gives 88 us for
// the loop below, 71 us with every operand L2-resident, ~29 us of it the HBM-bound epilogue):
//   * BK = 32, 4-stage ring, the two waves of every SIMD half a step apart (one reads / issues LDS-DMA while its partner
//     runs MFMAs, two barriers per step): 91 us -- no gain;
//   * separate rings fed by separate waves (waves 0-3 stream A through 7 stages = 72 KB in flight per CU, waves 4-7 stream
//     W through 3; one vmcnt queue per operand): 98-101 us -- deeper A prefetch does not help, the 32-deep step costs.
// So neither the SIMD partners' phase alignment nor the depth of the A stream is what holds the loop at ~55 % of its MFMA
// time; the LDS fragment traffic (every wave reads all of A: 224 KB per 64-deep step and CU) is the next suspect.

template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (LO < HI) { f(std::integral_constant<int, LO>{}); static_for<LO + 1, HI>(f); }
}

struct PanelP {
    const bf16* A; const bf16* W; int M, K; long lda, ldw;
    // forward
    const float* bias; const float* resid; long ldr; float* out; long ldo;
    const float* gamma; const float* beta; float eps; bf16* y; float* mean; float* rstd;
    // backward
    const float* x; long ldx; float* g; long ldg; bf16* gb; long ldgb; float* partials; int g_init;
    // stochastic depth: forward out = resid + row_scale[m] (A W^T + bias); backward gb = bf16(g gb_scale[m]) (and the third
    // column sum, the bias gradient of the branch in front of the residual add, sums the scaled rows)
    const float* row_scale; const float* gb_scale;
    // wide (plain Linear with N = ncb x 384 output columns, bf16 out): out = epilogue(A W^T) one 384-column block after the other
    bf16* outb; long ldob; const bf16* aux_in; bf16* aux_out; long ld_aux; int ncb, n_total;
    // fused MLP (forward, EP = EP_MLP): A -> fc1 (W, bias1, `hidden` columns) -> GELU -> fc2 (W2, bias) -> the forward epilogue
    const bf16* W2; long ldw2; const float* bias1; int hidden;
};


enum { MODE_FWD = 0, MODE_BWD = 1, MODE_WIDE = 2 };
// MODE_WIDE epilogues (the GV_EPI_* combinations of the hot path's wide products)
// EP_BIAS_RESID: f32 output = (A W^T + bias) * row_scale + resid -- x + proj(a) / x + fc2(h) of the widths that have no fused
// LayerNorm kernel (ViT-B: N = 768)
enum { EP_NONE = 0, EP_BIAS = 1, EP_BIAS_GELU = 2, EP_BIAS_GELU_SAVE = 3, EP_DGELU = 4, EP_BIAS_RESID = 5 };
constexpr int EP_MLP = 1;           // (MODE_FWD only: the fused-MLP variant of the forward kernel)

// PP = 1: the k-loop of gemm_dw8.h (ping-pong halves, three phases per 64-deep K-tile, counted LDS-DMA stream) instead of the
// one-stage-ahead loop: waves (wm, wn) = (wave >> 2, wave & 3); the wm = 0 half owns the first FMH = ceil(FM / 2) row fragments,
// the wm = 1 half the rest; in weight block b (phase b of a K-tile) a wave owns column fragments {2 wn, 2 wn + 1}.
template <int FM, int NW, int BK, bool TB, int MODE, int EP = 0, int PP = 0>
__global__ __launch_bounds__(NW * 64, 2) void panel_kernel(const PanelP p) {
    using C = PC<FM, NW, BK>;
    constexpr int NF = C::NF;
    static_assert(!PP || (NW == 8 && BK == 64), "ping-pong loop: 8 waves, BK = 64");
    // MODE_FWD with EP = EP_MLP: mlp.fc1 -> GELU -> mlp.fc2 in ONE kernel (vit.pyc@L98-104) in front of the forward epilogue
    // (+ bias + residual + the next LayerNorm): the hidden activation never exists in HBM.  See the k-loop below.
    constexpr bool MLP = MODE == MODE_FWD && EP == EP_MLP;
    static_assert(!MLP || (!PP && NW == 8 && BK == 64 && !TB && C::A_BYTES <= C::W_BLOCK), "fused MLP: one-stage-ahead loop, 8 waves, natural weights, <= 128 rows");
    constexpr int FMH = (FM + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    GV_LDS char* smem = (GV_LDS char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li16 = lane & 15, gq = lane >> 4;
    const int M = p.M;
    // MODE_WIDE (N = ncb x 384 output columns): a workgroup owns ONE 384-column block of the output and walks up to ncb
    // consecutive row panels with it; the ncb workgroups that walk the SAME panels (one per column block) are neighbours on
    // one XCD (workgroup ids go round-robin over the 8 XCDs: id & 7 is the XCD, id >> 3 the slot on it), so an A panel is
    // fetched from HBM once and its other ncb - 1 readers hit that XCD's L2 -- the XCD's working set is 32 / ncb panels
    // + the weights (~2.3 MB), where the round-2 mapping (one workgroup = one panel x all column blocks) re-read every panel
    // ncb times out of an L2 that 32 different panels (4.3 MB) had already left: 1.41 x the algorithmic traffic on fc1.
    int m0 = blockIdx.x * C::BM;
    int cb = 0, nit = 1, first_panel = 0;                 // column block; row panels this workgroup walks, the first one's index
    if constexpr (MODE == MODE_WIDE) {
        // (integer division runs on the vector ALU: readfirstlane brings the uniform results back to scalar registers, or every
        //  row address derived from m0 / cb would occupy vector registers this kernel does not have)
        const int ncb_ = p.ncb, slot = blockIdx.x >> 3;
        const int lg = __builtin_amdgcn_readfirstlane(slot / ncb_);
        const int grp = lg * 8 + (blockIdx.x & 7);
        cb = slot - lg * ncb_;
        const int P = __builtin_amdgcn_readfirstlane((M + C::BM - 1) / C::BM), first = grp * ncb_;
        if (first >= P) return;                           // (whole workgroup: no barrier is pending)
        nit = P - first < ncb_ ? P - first : ncb_;
        first_panel = first;
        m0 = first * C::BM;
        m0 = m0 < M - C::BM ? m0 : M - C::BM;             // the last panel is shifted back to end at row M (see panel_origin below)
    }
    // MODE_WIDE row panels: panel i starts at min(i BM, M - BM) -- every panel holds BM valid rows, so the per-lane source offsets
    // of the A pieces are the same for all panels (no clamp at M to recompute when the stream moves on to the next panel: that
    // cost 6 vector registers this kernel does not have -- spills whose reloads drained the counted DMA stream).  The rows the
    // last two panels share are computed twice, by the same arithmetic in the same order: bit-identical stores.  (The wide
    // epilogues are pure functions of the row; gv_panel_wide refuses out == resid for the one that reads what it overwrites.)
    auto panel_origin = [&](int idx) { const int o = idx * C::BM; return o < M - C::BM ? o : M - C::BM; };

    // acc[i][j][r]: row m0 + 16 i + (lane & 15), column 16 (NW j + wave) + 4 (lane >> 4) + r
    f32x4 acc[PP ? 1 : FM][NF];
    // PP: acc8[b][i][j][r]: row m0 + 16 (wm FMH + i) + (lane & 15), column 128 b + 16 (2 wn + j) + 4 (lane >> 4) + r
    f32x4 acc8[PP ? 3 : 1][PP ? FMH : 1][2];
    const int wm = wave >> 2, wn = wave & 3;
    const int f0 = wm * FMH, nfr = wm == 0 ? FMH : FM - FMH;

    TileSrc<false, C::A_ROWS, BK, NW> srcA;
    TileSrc<TB, 128, BK, NW> srcW[3];
    const int n_total = MODE == MODE_WIDE ? p.n_total : PN;
    TileSrc<false, 128, BK, NW> srcW2[MLP ? 3 : 1];
    if constexpr (!PP) {
        srcA.setup(p.A, p.lda, m0, M, wave, lane);
#pragma unroll
        for (int j = 0; j < 3; ++j) srcW[j].setup(p.W, p.ldw, 128 * j, MLP ? p.hidden : n_total, wave, lane);
        if constexpr (MLP) {
#pragma unroll
            for (int j = 0; j < 3; ++j) srcW2[j].setup(p.W2, p.ldw2, 128 * j, PN, wave, lane);
        }
    }
    const int nt = p.K / BK;
    // ---- fused MLP: ring slots of 3 weight blocks (48 KB).  A fc1 stage = the A tile + 2 blocks of W1 (256 hidden columns x 64 k),
    // a fc2 stage = 3 blocks of W2 (384 output columns x 64 hidden columns); behind the two slots, the chunk's hidden activation
    // H [A_ROWS][256] bf16 as four 64-deep tiles in the A operand's LDS format
    constexpr int HC = 256, MSLOT = 3 * C::W_BLOCK, HT = C::A_ROWS * BK * 2;
    auto mlp_issue1 = [&](int c, int r, int slot) {             // fc1 stage r of chunk c
        GV_LDS char* st = smem + slot * MSLOT;
        srcA.issue(p.lda, r * BK, p.K, st, wave);
#pragma unroll
        for (int j = 0; j < 2; ++j) srcW[j].issue(p.ldw, c * HC * (int)p.ldw + r * BK, p.K, st + C::A_BYTES + j * C::W_BLOCK, wave);
    };
    auto mlp_issue2 = [&](int c, int kt, int slot) {            // fc2 stage kt of chunk c
        GV_LDS char* st = smem + slot * MSLOT;
#pragma unroll
        for (int j = 0; j < 3; ++j) srcW2[j].issue(p.ldw2, c * HC + kt * BK, p.hidden, st + j * C::W_BLOCK, wave);
    };
    // PP: LDS-DMA sources as uniform base (SGPRs) + per-lane 32-bit byte offset, as in gemm_dw8.h.  A image piece pc: rows
    // (wave a + pc) 8 .. + 7 (clamped at M), 128 B each, 16-B chunk XOR (row & 7); W block piece pc: natural -- weight rows
    // (wave 2 + pc) 8 .. + 7 of the block; transposed -- reduction rows (wave 2 + pc) 4 .. + 3, 256 B of the block's columns each
    unsigned voffA[PP ? C::A_PPW : 1], voffW[2];
    auto a_offsets = [&](unsigned (&vo)[PP ? C::A_PPW : 1], int origin) {      // per-lane byte offsets of a panel's A pieces from its first row
#pragma unroll
        for (int pc = 0; pc < C::A_PPW; ++pc) {
            const int rr = (wave * C::A_PPW + pc) * 8 + (lane >> 3);
            int row;
            if constexpr (MODE == MODE_WIDE) row = origin + (rr < C::BM ? rr : rr - C::BM);      // surplus staged rows (never stored): any row of the panel
            else { row = origin + rr; row = row < M ? row : M - 1; }
            vo[pc] = (unsigned)(((long)(row - origin) * p.lda + (((lane & 7) ^ (rr & 7)) << 3)) * 2);
        }
    };
    if constexpr (PP) {
        a_offsets(voffA, m0);
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            if constexpr (!TB) {
                const int rr = (wave * 2 + pc) * 8 + (lane >> 3);
                voffW[pc] = (unsigned)(((long)rr * p.ldw + (((lane & 7) ^ (rr & 7)) << 3)) * 2);
            } else {
                const int rr = (wave * 2 + pc) * 4 + (lane >> 4), s16 = lane & 15;
                voffW[pc] = (unsigned)(((long)rr * p.ldw + ((((s16 >> 1) ^ swz_t(rr)) << 4) + (s16 & 1) * 8)) * 2);
            }
        }
    }
    unsigned long long baseA = (unsigned long long)(p.A + (long)m0 * p.lda);         // (advances panel by panel in MODE_WIDE)
    const unsigned long long baseW = (unsigned long long)p.W
        + (unsigned long long)cb * (unsigned long long)(TB ? (long)PN * 2 : (long)PN * p.ldw * 2);     // this workgroup's column block
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    auto issue = [&](int t) {
        GV_LDS char* st = smem + (t & 1) * C::STAGE;
        const int k0 = t * BK;
        srcA.issue(p.lda, k0, p.K, st, wave);
#pragma unroll
        for (int j = 0; j < 3; ++j) srcW[j].issue(p.ldw, k0, p.K, st + C::A_BYTES + j * C::W_BLOCK, wave);
    };
    // ---- PP: pieces of a K-tile in DMA-stream order: A (a = A_PPW per wave), W0, W1, W2 (two per wave each); n = a + 6
    constexpr int PA = C::A_PPW, PN_T = PA + 6;
    // Tile nt (the one the stream reaches behind a panel's last): MODE_WIDE with another row panel to come -- that panel's
    // tile 0 (the next BM rows of A from k = 0, the same 384 weight rows), so the stream runs on across the panel boundary
    // and the panel's epilogue covers its latency (the epilogue's image lives in stage 1 and behind it: the k-loop ends in
    // stage 1, nt is even); otherwise a re-read of the last tile that is never consumed (keeps the counted waits uniform).
    // Tile nt + 1 is never issued: the last tile's phases run the TAIL variant below.
    // A phase resolves "past the end" once, with scalar selects, for the pieces it issues: (uc, ba) = the tile's k index and A
    // base; uc_past / baseA_past (set per panel) describe tile nt.  The per-lane A offsets are the same for every panel.
    int uc_past = nt - 1;
    unsigned long long baseA_past = baseA;
    auto issue_x = [&](auto Xc, int u, int uc, unsigned long long ba) {
        constexpr int X = decltype(Xc)::value;
        const unsigned st = lds0 + (u & 1) * C::STAGE;
        if constexpr (X < PA) {
            glds16_s<0>(ba + (unsigned long long)uc * (BK * 2), voffA[X], st + (wave * PA + X) * 1024);
        } else {
            constexpr int b = (X - PA) / 2, pc = (X - PA) % 2;
            const unsigned dst = st + C::A_BYTES + b * C::W_BLOCK + (wave * 2 + pc) * 1024;
            if constexpr (!TB) glds16_s<0>(baseW + ((unsigned long long)(128 * b) * p.ldw + (unsigned long long)uc * BK) * 2, voffW[pc], dst);
            else glds16_s<b * 256>(baseW + (unsigned long long)uc * BK * p.ldw * 2, voffW[pc], dst);
        }
    };
    if constexpr (PP) static_for<0, PN_T>([&](auto X) { issue_x(X, 0, 0, baseA); });
    else if constexpr (MLP) mlp_issue1(0, 0, 0);
    else issue(0);
    float s_dg[3][2], s_db[3][2], s_g[3][2];          // backward: column sums over this workgroup's rows
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e) { s_dg[c][e] = 0.f; s_db[c][e] = 0.f; s_g[c][e] = 0.f; }
    for (int it = 0; it < nit; ++it) {
    int m0_next = m0;
    if constexpr (PP) {
        const bool has_next = MODE == MODE_WIDE && it + 1 < nit;
        uc_past = has_next ? 0 : nt - 1;
        m0_next = has_next ? panel_origin(first_panel + it + 1) : m0;
        baseA_past = (unsigned long long)(p.A + (long)m0_next * p.lda);
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int i = 0; i < FMH; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc8[b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // prologue: tile 0 is out (start of the kernel / before the previous block's epilogue); A, W0, W1 of tile 1 follow
        static_for<0, PA + 4>([&](auto X) { issue_x(X, 1, 1, baseA); });
        wait_vmcnt<PN_T + 2>();                               // A and W0 of tile 0 landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();            // the second half runs one barrier late
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 fa[2][FMH], fb[2][2];
        // Phase (t, J), tile t in stage S = t & 1:
        //   load segment:  fragment reads of THIS phase (J = 0: the tile's A fragments and W0's; J = 1, 2: W1's, W2's), LDS-DMA
        //                  issue, counted vmcnt, lgkmcnt(0)   | s_barrier |   MFMA segment: 4 FMH MFMAs   | s_barrier
        // (fragments are single-buffered: 144 accumulator registers leave no room for a second set).
        // Stream position of piece (u, X) = n u + X (X: A 0 .. a-1, W0 a, a+1, W1 a+2, a+3, W2 a+4, a+5); issue: J = 0 -> W2 of
        // t + 1, J = 1 -> A of t + 2, J = 2 -> W0, W1 of t + 2; P0 = 2 a + 10 pieces are out before phase (0, 0).
        // RAW (staggered halves): an image read in load segment p was waited for in load segment p - 1 by its issuer:
        //   (t, 0) retires W1(t): n + 2 pieces may fly; (t, 1) retires W2(t): n + a; (t, 2) retires A, W0 of t + 1: n + 2.
        // WAR: the reads of a load segment are retired (lgkmcnt(0)) before its barrier, the image is restaged one phase
        // later at the earliest (A: read (t, 0), restaged (t, 1); W0: (t, 0) / (t, 2); W1: (t, 1) / (t, 2); W2: (t, 2) / (t + 1, 0)).
        // The block's last tile (t = nt - 1, always in stage 1): its W2 pieces are tile nt's, nothing of tile nt + 1 is issued,
        // and what may still fly when W2(t) / A, W0 of tile nt must have landed is then n / 4 pieces (uniform branches).
        auto phase = [&](auto Jc, auto Sc, int t) {
            constexpr int J = decltype(Jc)::value, S = decltype(Sc)::value;
            const bool tail = S == 1 && t == nt - 1;
            GV_LDS char* st = smem + S * C::STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[ks][j] = read_frag<TB, 128, BK>(st + C::A_BYTES + J * C::W_BLOCK, 2 * wn + j, ks, lane);
            if constexpr (J == 0) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < FMH; ++i) fa[ks][i] = read_frag<false, C::A_ROWS, BK>(st, f0 + i, ks, lane);
                const bool past = t + 1 >= nt;
                const int uc = past ? uc_past : t + 1;
                static_for<PA + 4, PN_T>([&](auto X) { issue_x(X, t + 1, uc, baseA); }); wait_vmcnt<PN_T + 2>();
            } else {
                const bool past = t + 2 >= nt;
                const int uc = past ? uc_past : t + 2;
                const unsigned long long ba = past ? baseA_past : baseA;
                if constexpr (J == 1) {
                    if (tail) wait_vmcnt<PN_T>();
                    else { static_for<0, PA>([&](auto X) { issue_x(X, t + 2, uc, ba); }); wait_vmcnt<PN_T + PA>(); }
                } else {
                    if (tail) wait_vmcnt<4>();
                    else { static_for<PA, PA + 4>([&](auto X) { issue_x(X, t + 2, uc, ba); }); wait_vmcnt<PN_T + 2>(); }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);               // this segment's reads are retired before its barrier
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            // (the wm = 1 half of an odd FM multiplies one row fragment past its own: staged rows, never stored -- keeps the
            //  cluster branch-free and the two halves equally long)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < FMH; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc8[J][i][j] = GV_MFMA_16x16x32(fb[ks][j], fa[ks][i], acc8[J][i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        for (int t = 0; t < nt; t += 2) {                     // nt is even (K % 128 == 0)
            phase(I0{}, I0{}, t); phase(I1{}, I0{}, t); phase(I2{}, I0{}, t);
            phase(I0{}, I1{}, t + 1); phase(I1{}, I1{}, t + 1); phase(I2{}, I1{}, t + 1);
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();            // pair the late half's last barrier
        // FWD / BWD: the epilogue's image overlays the whole ring, so the stream's never-consumed last pieces must have landed.
        // WIDE: the image lives in stage 1 and behind it and the pieces in flight are the NEXT panel's tile 0 into stage 0 --
        // they keep flying through the epilogue (the kernel's end waits for the last ones)
        if constexpr (MODE != MODE_WIDE) wait_vmcnt<0>();
    }
    if constexpr (!PP) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (MLP) {
        // Per chunk c of 256 hidden columns: fc1 over K (nt stages; wave w owns the chunk's column fragments {w, w + 8}) -> + bias1,
        // GELU, bf16 -> H in LDS -> fc2 over the chunk (4 stages; wave w owns output fragments {w, w + 8, w + 16} as in the plain
        // loop, A fragments from H).  One accumulator chain per output over the hidden index in increasing order: bit-identical to
        // gv_linear (BIAS | GELU) followed by gv_linear_ln_fwd.  The DMA stream runs one stage ahead across both products.
        GV_LDS char* Himg = smem + 2 * MSLOT;
        const int nchunk = p.hidden / HC;
        int slot = 0;
        for (int c = 0; c < nchunk; ++c) {
            f32x4 acc1[FM][2], b1v[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) b1v[j] = *(const f32x4*)(p.bias1 + c * HC + 16 * (NW * j + wave) + gq * 4);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < nt; ++r, slot ^= 1) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (r + 1 < nt) mlp_issue1(c, r + 1, slot ^ 1); else mlp_issue2(c, 0, slot ^ 1);
                GV_LDS char* cur = smem + slot * MSLOT;
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) {
                    bf16x8 fw[2], fa[FM];
#pragma unroll
                    for (int j = 0; j < 2; ++j) fw[j] = read_frag<false, 128, BK>(cur + C::A_BYTES + j * C::W_BLOCK, wave, ks, lane);
#pragma unroll
                    for (int i = 0; i < FM; ++i) fa[i] = read_frag<false, C::A_ROWS, BK>(cur, i, ks, lane);
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc1[i][j] = GV_MFMA_16x16x32(fw[j], fa[i], acc1[i][j]);
                }
            }
            // hidden column 16 (8 j + wave) + 4 gq + r of the chunk = k index of fc2: tile 2 j + (wave >> 2), 16-B chunk 2 (wave & 3) + (gq >> 1)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 v = acc1[i][j] + b1v[j];
                    const int row = 16 * i + li16;
                    *(GV_LDS bf16x4*)(Himg + (2 * j + (wave >> 2)) * HT + row * (BK * 2) + (((2 * (wave & 3) + (gq >> 1)) ^ swz_n<BK>(row)) << 4) + (gq & 1) * 8) =
                        bf16x4{(bf16)gelu_f(v[0]), (bf16)gelu_f(v[1]), (bf16)gelu_f(v[2]), (bf16)gelu_f(v[3])};
                }
            __builtin_amdgcn_s_waitcnt(0xC07F);       // H is written before the next barrier lets anybody read it
            for (int kt = 0; kt < HC / BK; ++kt, slot ^= 1) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (kt + 1 < HC / BK) mlp_issue2(c, kt + 1, slot ^ 1); else if (c + 1 < nchunk) mlp_issue1(c + 1, 0, slot ^ 1);
                GV_LDS char* cur = smem + slot * MSLOT;
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) {
                    bf16x8 fw[NF], fa[FM];
#pragma unroll
                    for (int j = 0; j < NF; ++j) fw[j] = read_frag<false, 128, BK>(cur + j * C::W_BLOCK, wave, ks, lane);
#pragma unroll
                    for (int i = 0; i < FM; ++i) fa[i] = read_frag<false, C::A_ROWS, BK>(Himg + kt * HT, i, ks, lane);
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < NF; ++j) acc[i][j] = GV_MFMA_16x16x32(fw[j], fa[i], acc[i][j]);
                }
            }
        }
    }
    if constexpr (!PP && !MLP) for (int t = 0; t < nt; ++t) {
        wait_vmcnt<0>();                      // my pieces of stage t have landed (nothing younger is in flight yet)
        __builtin_amdgcn_s_barrier();         // everybody's have; every wave is past its reads of stage t - 1
        if (t + 1 < nt) issue(t + 1);
        GV_LDS char* cur = smem + (t & 1) * C::STAGE;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            bf16x8 fw[NF], fa[FM];
#pragma unroll
            for (int j = 0; j < NF; ++j) {          // fragment f = NW j + wave lives in 128-column block f / 8 at local index f % 8
                const int f = NW * j + wave;
                fw[j] = read_frag<TB, 128, BK>(cur + C::A_BYTES + (f >> 3) * C::W_BLOCK, f & 7, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = read_frag<false, C::A_ROWS, BK>(cur, i, ks, lane);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = GV_MFMA_16x16x32(fw[j], fa[i], acc[i][j]);
        }
    }
    // ---- MODE_WIDE epilogue: PER WAVE, no workgroup barrier between its two ends.  A wave owns, of every 16-row fragment of
    // its half, the columns 128 b + 32 wn + [0, 32) of the three weight blocks b: it transposes one fragment row at a time
    // through a PRIVATE LDS image ([16 rows][3 x 32 columns] f32, stride 100 floats: b128 writes and b64 reads conflict-free)
    // and works on units of 4 rows x 32 columns (lane: row 4 q + (lane >> 4), columns 2 (lane & 15) + {0, 1}): every global access
    // of a unit is four 64-B row pieces (128-B for the f32 output), which the neighbouring wave (wn + 1) completes to full lines.
    // Round 2 ran the wide epilogues on the LayerNorm kernels' scheme -- a workgroup-wide image, one wave per 768-B row, two
    // barriers per 32 rows: 8 us per 176 x 384 panel with NO global store in it (a lab build without stores), as long as
    // the panel's k-loop.  Nothing here needs a whole row in one wave, so nothing needs the other waves.
    if constexpr (MODE == MODE_WIDE) {
        constexpr int RS = 100;
        static_assert(C::STAGE + NW * 16 * RS * 4 <= C::LDS_WIDE, "per-wave images must fit behind ring stage 0");
        constexpr bool WB = (EP >= EP_BIAS && EP <= EP_BIAS_GELU_SAVE) || EP == EP_BIAS_RESID;
        // (lane-derived addresses are recomputed per panel from an opaque copy of the lane id: hoisted out of the panel loop they
        //  would be held in registers across the k-loop, which has none to spare -- spills whose reloads drain the DMA stream)
        int le = lane;
        asm volatile("" : "+v"(le));
        const int urow = le >> 4, ucol = (le & 15) * 2, li16e = le & 15, gqe = le >> 4;
        GV_LDS float* wimg = (GV_LDS float*)(smem + C::STAGE) + wave * (16 * RS);
        f32x4 bw[3][2];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bw[b][j] = (WB && p.bias) ? *(const f32x4*)(p.bias + cb * PN + 128 * b + 16 * (2 * wn + j) + gqe * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                          // every wave is past ring stage 1: it is image space now
        static_for<0, FMH>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            if (i < nfr) {                                        // (wave-uniform; static but for a half's last fragment)
                const int mrow = m0 + 16 * (f0 + i) + urow;       // + 4 q: this lane's row of unit (b, q); < M (every panel holds BM rows)
                const int ccol = cb * PN + 32 * wn + ucol;        // + 128 b
                // row pointers: ONE 64-bit multiply per fragment and stream, then uniform strides (4 q rows, 128 b columns) -- the
                // epilogue is bound by its vector-ALU instruction count, and a 64-bit row * pitch product per unit was a fifth of it
                [[maybe_unused]] bf16* const o_bf = MODE == MODE_WIDE && EP != EP_BIAS_RESID ? p.outb + (long)mrow * p.ldob + ccol : nullptr;
                [[maybe_unused]] bf16* const o_aux = EP == EP_BIAS_GELU_SAVE ? p.aux_out + (long)mrow * p.ld_aux + ccol : nullptr;
                [[maybe_unused]] const bf16* const i_aux = EP == EP_DGELU ? p.aux_in + (long)mrow * p.ld_aux + ccol : nullptr;
                [[maybe_unused]] float* const o_f32 = EP == EP_BIAS_RESID ? p.out + (long)mrow * p.ldo + ccol : nullptr;
                [[maybe_unused]] const float* const i_res = EP == EP_BIAS_RESID ? p.resid + (long)mrow * p.ldr + ccol : nullptr;
                const long s_ob = 4 * p.ldob, s_aux = 4 * p.ld_aux, s_of = 4 * p.ldo, s_res = 4 * p.ldr;            // (scalar registers)
                bf16x2 ax[EP == EP_DGELU ? 12 : 1];
                f32x2 ar[EP == EP_BIAS_RESID ? 12 : 1];
                float rsc[EP == EP_BIAS_RESID ? 4 : 1];
                if constexpr (EP == EP_DGELU || EP == EP_BIAS_RESID) {        // the fragment's global operand rows, all requested up front
#pragma unroll
                    for (int b = 0; b < 3; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if constexpr (EP == EP_DGELU) ax[b * 4 + q] = *(const bf16x2*)(i_aux + q * s_aux + 128 * b);
                            else ar[b * 4 + q] = *(const f32x2*)(i_res + q * s_res + 128 * b);
                        }
                    if constexpr (EP == EP_BIAS_RESID) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) rsc[q] = p.row_scale ? p.row_scale[mrow + 4 * q] : 1.0f;
                    }
                }
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4 v = acc8[b][i][j];
                        if constexpr (WB) v += bw[b][j];
                        *(GV_LDS f32x4*)(wimg + li16e * RS + 32 * b + 16 * j + gqe * 4) = v;
                    }
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x2 t2 = *(GV_LDS f32x2*)(wimg + (4 * q + urow) * RS + 32 * b + ucol);       // (a wave's LDS accesses execute in order)
                        float v0 = t2[0], v1 = t2[1];
                        if constexpr (EP == EP_BIAS_RESID) {
                            *(f32x2*)(o_f32 + q * s_of + 128 * b) = f32x2{fmaf(v0, rsc[q], ar[b * 4 + q][0]), fmaf(v1, rsc[q], ar[b * 4 + q][1])};
                        } else {
                            if constexpr (EP == EP_BIAS_GELU_SAVE) {
                                // nontemporal: the saved pre-activation is not read again before the backward pass; kept out of the
                                // caches, more of h (the next kernel's A operand, written beside it) is still in L2 / Infinity Cache
                                // when fc2 reads it (round 2: fc2 + LayerNorm forward 99 -> 89 us)
                                __builtin_nontemporal_store(bf16x2{(bf16)v0, (bf16)v1}, (bf16x2*)(o_aux + q * s_aux + 128 * b));
                            }
                            if constexpr (EP == EP_BIAS_GELU || EP == EP_BIAS_GELU_SAVE) { v0 = gelu_f(v0); v1 = gelu_f(v1); }
                            if constexpr (EP == EP_DGELU) { v0 *= dgelu_f((float)ax[b * 4 + q][0]); v1 *= dgelu_f((float)ax[b * 4 + q][1]); }
                            bf16x2* dst = (bf16x2*)(o_bf + q * s_ob + 128 * b);
                            *dst = bf16x2{(bf16)v0, (bf16)v1};
                        }
                    }
            }
        });
        __syncthreads();                                          // images are read: stage 1 may receive the next panel's tile 1
    }
    // ---- epilogue (FWD / BWD).  The accumulators pass through an LDS image so that the row phase works on whole rows, ONE WAVE PER
    // ROW (wave w: rows w, w + 8, .. of the pass): every global access is a full 512-B row segment and the row reductions
    // of the LayerNorm stay inside a wave.  The global rows a pass needs (forward: the residual; backward: x and g, mean,
    // rstd) are all requested BEFORE the pass's image barrier, so their HBM latency overlaps the image traffic and the
    // other rows' arithmetic instead of being paid once per row.
    // PP: a pass takes the SAME wave-relative row fragments [q0, q0 + IBH) of both halves (image row blocks h IBH + ii), so which
    // accumulators a pass consumes is static and their registers are released pass by pass
    constexpr int IBH = !PP ? 1 : (MODE == MODE_FWD ? (FMH < 2 ? FMH : 2) : 1);
    constexpr int IB = PP ? 2 * IBH : (MODE == MODE_WIDE ? C::ib_wide() : C::ib(NW == 8 ? (MODE == MODE_FWD ? 5 : 3) : (MODE == MODE_FWD ? 2 : 1)));
    static_assert(!PP || IB * 16 * IMG_STRIDE * 4 + (MODE == MODE_WIDE ? C::STAGE : 0) <= (MODE == MODE_WIDE ? C::LDS_WIDE : C::LDS), "PP image");
    constexpr int RPW = 16 * IB / NW;                             // rows per wave and pass
    GV_LDS float* img = (GV_LDS float*)(smem + (MODE == MODE_WIDE ? C::STAGE : 0));
    constexpr bool HAS_BIAS = MODE == MODE_FWD || (MODE == MODE_WIDE && ((EP >= EP_BIAS && EP <= EP_BIAS_GELU_SAVE) || EP == EP_BIAS_RESID));
    f32x4 bias4[NF], bias8[3][2];
    if constexpr (HAS_BIAS) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
            bias4[j] = p.bias ? *(const f32x4*)(p.bias + cb * PN + 16 * (NW * j + wave) + gq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bias8[b][j] = (PP && p.bias) ? *(const f32x4*)(p.bias + cb * PN + 128 * b + 16 * (2 * wn + j) + gq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    static_assert(MODE != MODE_WIDE || PP, "MODE_WIDE runs on the ping-pong k-loop only (gv_panel_wide requires K % 128 == 0)");
    // per-lane column constants of the row phase: lane owns columns (c * 64 + lane) * 2 + {0, 1}, c = 0..2
    float gm[3][2], bt[3][2];
    const bool ln = MODE == MODE_BWD || (MODE == MODE_FWD && p.gamma != nullptr);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int col = (c * 64 + lane) * 2;
        if (ln) { const f32x2 gv = *(const f32x2*)(p.gamma + col); gm[c][0] = gv[0]; gm[c][1] = gv[1]; }
        else { gm[c][0] = gm[c][1] = 1.f; }
        if (MODE == MODE_FWD && ln) { const f32x2 bv = *(const f32x2*)(p.beta + col); bt[c][0] = bv[0]; bt[c][1] = bv[1]; }
        else { bt[c][0] = bt[c][1] = 0.f; }
    }

    // rows of pass i0: image row r = wave + NW rr of the pass  ->  valid (compile-time per rr) and global row offset from m0
    //   plain: row blocks i0 .. i0 + ni - 1 in order;   PP: slot = rr >> 1, half h = slot / IBH, fragment h FMH + i0 + slot % IBH
    auto row_valid = [&](int i0, int ni, int rr) constexpr {
        if (!PP) return rr < 16 * ni / NW;
        const int slot = rr >> 1, h = slot / IBH, ii = slot % IBH;
        return ii < ni && (h == 0 || i0 + ii < FM - FMH);
    };
    auto row_off = [&](int i0, int rr) constexpr {
        if (!PP) return i0 * 16 + NW * rr;
        const int slot = rr >> 1, h = slot / IBH, ii = slot % IBH;
        return 16 * (h * FMH + i0 + ii) + 8 * (rr & 1);
    };
    if constexpr (MODE != MODE_WIDE) {
    // The row phase is HBM-bound (s_memtime stamps of a lab build: forward 170 MB, backward 237 MB per launch at 5.6 - 5.9 TB/s, as long
    // as the k-loop).  Requesting pass p + 1's rows ahead of pass p's stores was measured and changes nothing (LAB_NOTES.md).
    constexpr int PSTEP = PP ? IBH : IB, NPASS = ((PP ? FMH : FM) + PSTEP - 1) / PSTEP;
    f32x2 pre_a[2][RPW][3], pre_b[2][MODE == MODE_BWD ? RPW : 1][3];
    float pre_mean[2][MODE == MODE_BWD ? RPW : 1], pre_rstd[2][MODE == MODE_BWD ? RPW : 1];
    auto prefetch = [&](auto Pc) {
        constexpr int ps = decltype(Pc)::value, i0 = ps * PSTEP, S = ps & 1;
        constexpr int ni = ((PP ? FMH : FM) - i0) < PSTEP ? ((PP ? FMH : FM) - i0) : PSTEP;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            if (row_valid(i0, ni, rr)) {
                int m = m0 + row_off(i0, rr) + wave;
                m = m < M ? m : M - 1;                            // clamped at M: surplus rows load valid memory and are never stored
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int col = (c * 64 + lane) * 2;
                    if constexpr (MODE == MODE_FWD) {
                        pre_a[S][rr][c] = p.resid ? *(const f32x2*)(p.resid + (long)m * p.ldr + col) : f32x2{0.f, 0.f};
                    } else {
                        pre_a[S][rr][c] = *(const f32x2*)(p.x + (long)m * p.ldx + col);
                        pre_b[S][rr][c] = p.g_init ? f32x2{0.f, 0.f} : *(const f32x2*)(p.g + (long)m * p.ldg + col);
                    }
                }
                if constexpr (MODE == MODE_BWD) { pre_mean[S][rr] = p.mean[m]; pre_rstd[S][rr] = p.rstd[m]; }
            }
        }
    };
    prefetch(std::integral_constant<int, 0>{});
    static_for<0, NPASS>([&](auto Pc) {
        constexpr int ps = decltype(Pc)::value, i0 = ps * PSTEP, S = ps & 1;
        constexpr int ni = ((PP ? FMH : FM) - i0) < PSTEP ? ((PP ? FMH : FM) - i0) : PSTEP;
        if constexpr (ps >= 1) prefetch(Pc);
        if constexpr (i0 == 0) __syncthreads();                   // every wave is past the ring: it is image space now
        // ---- accumulators -> LDS image [rows of this pass][384 columns] f32
        if constexpr (PP) {
#pragma unroll
            for (int i = 0; i < FMH; ++i) {
                if (i >= i0 && i < i0 + ni && i < nfr) {          // static but for the last (a half's own fragment count)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            f32x4 v = acc8[b][i][j];
                            if constexpr (HAS_BIAS) v += bias8[b][j];
                            *(GV_LDS f32x4*)(img + ((wm * IBH + i - i0) * 16 + li16) * IMG_STRIDE + 128 * b + 16 * (2 * wn + j) + gq * 4) = v;
                        }
                }
            }
        } else {
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) {
                if (ii < ni) {
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        f32x4 v = acc[i0 + ii][j];
                        if constexpr (HAS_BIAS) v += bias4[j];
                        *(GV_LDS f32x4*)(img + (ii * 16 + li16) * IMG_STRIDE + 16 * (NW * j + wave) + gq * 4) = v;
                    }
                }
            }
        }
        __syncthreads();
        // (pass 0 still holds most of the accumulators: its successor's rows are requested at the top of pass 1, as before --
        //  requesting them here overflows the register file by ~20 registers at FM = 11 / 12)
        if constexpr (false) {}

        // ---- one wave per row
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            if (row_valid(i0, ni, rr)) {
                const int r = wave + NW * rr;
                const int m = m0 + row_off(i0, rr) + wave;
                if (m < M) {                                      // wave-uniform
                    float v[3][2];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const f32x2 t2 = *(GV_LDS f32x2*)(img + r * IMG_STRIDE + (c * 64 + lane) * 2);
                        v[c][0] = t2[0]; v[c][1] = t2[1];
                    }
                    if constexpr (MODE == MODE_FWD) {
                        float* orow = p.out + (long)m * p.ldo;
                        const float rs = p.row_scale ? p.row_scale[m] : 1.0f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            v[c][0] = fmaf(v[c][0], rs, pre_a[S][rr][c][0]); v[c][1] = fmaf(v[c][1], rs, pre_a[S][rr][c][1]);
                            *(f32x2*)(orow + (c * 64 + lane) * 2) = f32x2{v[c][0], v[c][1]};
                        }
                        if (ln) {
                            float sm = 0.f;
#pragma unroll
                            for (int c = 0; c < 3; ++c) sm += v[c][0] + v[c][1];
                            const float mean = wave_sum_dpp(sm) * (1.0f / PN);
                            float q = 0.f;
#pragma unroll
                            for (int c = 0; c < 3; ++c) { const float d0 = v[c][0] - mean, d1 = v[c][1] - mean; q += d0 * d0 + d1 * d1; }
                            const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) * (1.0f / PN) + p.eps);
                            bf16* yrow = p.y + (long)m * PN;
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const bf16x2 yv = bf16x2{(bf16)((v[c][0] - mean) * rstd * gm[c][0] + bt[c][0]), (bf16)((v[c][1] - mean) * rstd * gm[c][1] + bt[c][1])};
                                *(bf16x2*)(yrow + (c * 64 + lane) * 2) = yv;
                            }
                            if (lane == 0) { p.mean[m] = mean; p.rstd[m] = rstd; }
                        }
                    } else {
                        // LayerNorm backward of this row: v = dy (the dX product, still f32)
                        float* grow = p.g + (long)m * p.ldg;
                        const float mean = pre_mean[S][rr], rstd = pre_rstd[S][rr];
                        float xh[3][2], wdy[3][2], gv[3][2];
                        float c1 = 0.f, c2 = 0.f;
#pragma unroll
                        for (int c = 0; c < 3; ++c)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                gv[c][e] = pre_b[S][rr][c][e];
                                xh[c][e] = (pre_a[S][rr][c][e] - mean) * rstd;
                                wdy[c][e] = v[c][e] * gm[c][e];
                                c1 += wdy[c][e];
                                c2 += wdy[c][e] * xh[c][e];
                                s_dg[c][e] += v[c][e] * xh[c][e];
                                s_db[c][e] += v[c][e];
                            }
                        c1 = wave_sum_dpp(c1) * (1.0f / PN);
                        c2 = wave_sum_dpp(c2) * (1.0f / PN);
                        const float gs = p.gb_scale ? p.gb_scale[m] : 1.0f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const int col = (c * 64 + lane) * 2;
                            float gbv[2];
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                gv[c][e] += rstd * (wdy[c][e] - c1 - xh[c][e] * c2);
                                gbv[e] = gv[c][e] * gs;
                                s_g[c][e] += gbv[e];
                            }
                            // nontemporal: neither row is read again by this launch, and 100 MB of them would pass through the L2
                            // that holds the weights every workgroup re-reads (fc1-dX + LayerNorm backward 88.3 -> 85.7 us)
                            __builtin_nontemporal_store(f32x2{gv[c][0], gv[c][1]}, (f32x2*)(grow + col));
                            if (p.gb) __builtin_nontemporal_store(bf16x2{(bf16)gbv[0], (bf16)gbv[1]}, (bf16x2*)(p.gb + (long)m * p.ldgb + col));
                        }
                    }
                }
            }
        }
        __syncthreads();                                          // image free for the next pass / the reduction below
    });
    }
    if constexpr (MODE == MODE_WIDE) {                            // next row panel (its tile 0 went out as tile nt of this panel's stream)
        m0 = m0_next; baseA = baseA_past;
    }
    }   // row panels (MODE_WIDE)
    if constexpr (MODE == MODE_WIDE) wait_vmcnt<0>();             // the stream's last, never-consumed pieces land before the LDS is released

    if constexpr (MODE == MODE_BWD) {
        // column sums over this workgroup's rows -> partials[blockIdx][3][384] (gv_ln_finalize folds them)
        GV_LDS float* red = (GV_LDS float*)smem;                  // [NW waves][3][384]
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int col = (c * 64 + lane) * 2 + e;
                red[(wave * 3 + 0) * PN + col] = s_dg[c][e];
                red[(wave * 3 + 1) * PN + col] = s_db[c][e];
                red[(wave * 3 + 2) * PN + col] = s_g[c][e];
            }
        __syncthreads();
        float* outp = p.partials + (long)blockIdx.x * 3 * PN;
        for (int idx = tid; idx < 3 * PN; idx += NW * 64) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red[w * 3 * PN + idx];
            outp[idx] = s;
        }
    }
}

// rows per workgroup for M rows: the smallest supported FM that covers M in as few full rounds of 256 workgroups as possible
constexpr int FM_SET8[] = {4, 7, 9, 11, 12};
int cu_budget() { return gv_cu_budget(); }       // 256, or 256 - C under a data-parallel run that leaves C CUs to RCCL (gv_common.h)
int pick_fm(int M) {
    const int m16 = (M + 15) / 16;
    const int rounds = (m16 + cu_budget() * 12 - 1) / (cu_budget() * 12);
    const int need = (m16 + cu_budget() * rounds - 1) / (cu_budget() * rounds);
    for (int fm : FM_SET8) if (fm >= need) return fm;
    return 12;
}

// MODE_WIDE launch geometry (see the kernel's id mapping): groups of ncb sibling workgroups, ceil(groups / 8) per XCD
int wide_grid(int M, int BM, int ncb) {
    const int P = (M + BM - 1) / BM, groups = (P + ncb - 1) / ncb;
    return 8 * ncb * ((groups + 7) / 8);
}
// ... and the smallest supported FM whose WORKING workgroups (ncb per group of panels; the grid's padding to 8 ncb exits at once)
// fit one round of the CU budget; 12 (several rounds) for larger M
int pick_fm_wide(int M, int ncb) {
    for (int fm : FM_SET8) {
        const int P = (M + 16 * fm - 1) / (16 * fm), groups = (P + ncb - 1) / ncb;
        if (groups * ncb <= cu_budget() && wide_grid(M, 16 * fm, ncb) <= 256) return fm;
    }
    return 12;
}

template <int FM, int NW, int BK, bool TB, int MODE, int EP = 0, int PP = 0>
int launch_panel(const PanelP& p, hipStream_t s, const char* name) {
    auto kern = panel_kernel<FM, NW, BK, TB, MODE, EP, PP>;
    using C = PC<FM, NW, BK>;
    constexpr bool MLP = MODE == MODE_FWD && EP == EP_MLP;
    constexpr int LDS_MLP = 2 * 3 * C::W_BLOCK + 4 * C::A_ROWS * BK * 2;      // two ring slots + the chunk's hidden activation
    static_assert(!MLP || (LDS_MLP <= 160 * 1024 && LDS_MLP >= C::LDS - 2 * C::A_BYTES), "fused MLP: LDS");
    constexpr int LDS_BYTES = MLP ? LDS_MLP : (MODE == MODE_WIDE ? C::LDS_WIDE : C::LDS);
    static GvLdsOptIn opt_in;
    if (int rc = gv_lds_opt_in(opt_in, (const void*)kern, LDS_BYTES, name)) return rc;
    struct Name { char s[96]; Name() { snprintf(s, sizeof(s), "panel_kernel<%d, %d, %d, %s, %d, %d, %d>", FM, NW, BK, TB ? "true" : "false", MODE, EP, PP); } };
    static const Name kn;                                           // as rocprofv3 prints it (initialised once, thread-safe)
    const char* kname = kn.s;
    const int grid = MODE == MODE_WIDE ? wide_grid(p.M, C::BM, p.ncb) : (p.M + C::BM - 1) / C::BM;
    // algorithmic bytes: forward  A row + f32 residual in + f32 row out + bf16 normalised row out + stats;
    // backward  dY row + f32 x row + f32 g in / out + bf16 g out; + the weight once
    // wide: A row + the bf16 output row (+ the saved / re-read pre-activation row)
    const int nout = MODE == MODE_WIDE ? p.n_total : PN;
    const double row_bytes = MODE == MODE_WIDE ? 2.0 * p.K + (EP == EP_BIAS_RESID ? 8.0 * nout : 2.0 * nout * ((EP == EP_BIAS_GELU_SAVE || EP == EP_DGELU) ? 2 : 1))
                             : MODE == MODE_FWD ? 2.0 * p.K + PN * 4 * 2 + PN * 2 + 8 : 2.0 * p.K + PN * 4 * 3 + PN * 2 + 8;
    const double flops = MLP ? 4.0 * p.M * p.hidden * p.K : 2.0 * p.M * nout * p.K;
    const double bytes = p.M * row_bytes + (MLP ? 4.0 * p.hidden * p.K : 2.0 * nout * p.K);
    const int th = gvtime::enabled() ? gvtime::begin(kname, flops, bytes, s) : -1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS_BYTES, s, p);
    gvtime::end(th, s);
    GV_LAUNCH_CHECK(name);
    return GV_OK;
}

template <bool TB, int MODE, int EP = 0>
int dispatch_fm(const PanelP& p, hipStream_t s, const char* name) {
    // the ping-pong k-loop walks K-tiles in pairs (static ring stage): K % 128 == 0; other depths keep the one-stage-ahead loop
    if (p.K % 128 == 0) {
        switch (MODE == MODE_WIDE ? pick_fm_wide(p.M, p.ncb) : pick_fm(p.M)) {
            case 4: return launch_panel<4, 8, 64, TB, MODE, EP, 1>(p, s, name);
            case 7: return launch_panel<7, 8, 64, TB, MODE, EP, 1>(p, s, name);
            case 9: return launch_panel<9, 8, 64, TB, MODE, EP, 1>(p, s, name);
            case 11: return launch_panel<11, 8, 64, TB, MODE, EP, 1>(p, s, name);
            default: return launch_panel<12, 8, 64, TB, MODE, EP, 1>(p, s, name);
        }
    }
    if constexpr (MODE == MODE_WIDE) return -1;                     // (gv_panel_wide admits K % 128 == 0 only)
    else
    switch (pick_fm(p.M)) {
        case 4: return launch_panel<4, 8, 64, TB, MODE, EP>(p, s, name);
        case 7: return launch_panel<7, 8, 64, TB, MODE, EP>(p, s, name);
        case 9: return launch_panel<9, 8, 64, TB, MODE, EP>(p, s, name);
        case 11: return launch_panel<11, 8, 64, TB, MODE, EP>(p, s, name);
        default: return launch_panel<12, 8, 64, TB, MODE, EP>(p, s, name);
    }
}

}  // namespace

// gv_linear's wide products (timing.h): C[M, N] bf16 = epilogue(A W^T) with N a multiple of 384, K a multiple of 128 and one of
// the hot path's epilogues, on the full-row kernel -- row panels sized for ONE round of workgroups, 768-B row segments out.
// Returns -1 when the call is not one of these (the caller then runs the 128x128-tile kernel).
int gv_panel_wide(const gv_linear_args* a, hipStream_t s) {
    if (a->trans_a || a->N % PN != 0 || a->K % 128 != 0 || a->M < 2048 || a->ldc % 2 != 0) return -1;
    if (a->alpha != 0.f && a->alpha != 1.f) return -1;
    const bool plan = s == (hipStream_t)(intptr_t)-1;       // gv_workspace_bytes: "would this call run here?" (these kernels take no scratch)
    if (a->c_is_f32) {      // x + Linear(a) with an f32 residual stream, where no fused LayerNorm kernel exists for the width
        if (a->trans_b || a->epilogue != (GV_EPI_BIAS | GV_EPI_RESID) || a->ldr % 2 != 0) return -1;
        if ((const void*)a->resid == (const void*)a->C) return -1;      // in place: the rows the last two panels share would be updated twice
        if (plan) return GV_OK;
        PanelP q{};
        q.A = (const bf16*)a->A; q.W = (const bf16*)a->B; q.M = a->M; q.K = a->K; q.lda = a->lda; q.ldw = a->ldb;
        q.bias = a->bias; q.out = (float*)a->C; q.ldo = a->ldc; q.resid = a->resid; q.ldr = a->ldr; q.row_scale = a->row_scale;
        q.ncb = a->N / PN; q.n_total = a->N;
        return dispatch_fm<false, MODE_WIDE, EP_BIAS_RESID>(q, s, "gv_linear(wide)");
    }
    PanelP p{};
    p.A = (const bf16*)a->A; p.W = (const bf16*)a->B; p.M = a->M; p.K = a->K; p.lda = a->lda; p.ldw = a->ldb;
    p.bias = a->bias; p.outb = (bf16*)a->C; p.ldob = a->ldc; p.aux_in = (const bf16*)a->aux_in; p.aux_out = (bf16*)a->aux_out; p.ld_aux = a->ld_aux;
    p.ncb = a->N / PN; p.n_total = a->N;
    if (a->ld_aux % 2 != 0) return -1;
    const int e = a->epilogue;
    if (plan) {
        const bool fwd = !a->trans_b && (e == GV_EPI_BIAS || e == (GV_EPI_BIAS | GV_EPI_GELU) || e == (GV_EPI_BIAS | GV_EPI_GELU | GV_EPI_SAVE_PRE));
        return (fwd || (a->trans_b && (e == 0 || e == GV_EPI_DGELU))) ? GV_OK : -1;
    }
    if (!a->trans_b) {
        if (e == GV_EPI_BIAS) return dispatch_fm<false, MODE_WIDE, EP_BIAS>(p, s, "gv_linear(wide)");
        if (e == (GV_EPI_BIAS | GV_EPI_GELU)) return dispatch_fm<false, MODE_WIDE, EP_BIAS_GELU>(p, s, "gv_linear(wide)");
        if (e == (GV_EPI_BIAS | GV_EPI_GELU | GV_EPI_SAVE_PRE)) return dispatch_fm<false, MODE_WIDE, EP_BIAS_GELU_SAVE>(p, s, "gv_linear(wide)");
    } else {
        if (e == 0) return dispatch_fm<true, MODE_WIDE, EP_NONE>(p, s, "gv_linear(wide)");
        if (e == GV_EPI_DGELU) return dispatch_fm<true, MODE_WIDE, EP_DGELU>(p, s, "gv_linear(wide)");
    }
    return -1;
}


extern "C" int gv_linear_ln_blocks(int32_t M) {
    if (M <= 0) return 0;
    return (M + 16 * pick_fm(M) - 1) / (16 * pick_fm(M));
}

extern "C" int gv_linear_ln_fwd(const gv_linear_ln_fwd_args* a, void* stream) {
    GV_REQUIRE(a && a->A && a->W && a->out, GV_E_NULL, "gv_linear_ln_fwd: null operand");
    GV_REQUIRE(a->N == PN, GV_E_UNSUPPORTED, "gv_linear_ln_fwd: built for N = %d output columns (ViT-S), got %d", PN, a->N);
    GV_REQUIRE(a->M > 0 && a->K > 0 && a->K % 64 == 0, GV_E_SHAPE, "gv_linear_ln_fwd: need M > 0 and K %% 64 == 0 (got M=%d K=%d)", a->M, a->K);
    GV_REQUIRE(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->ldo % 2 == 0 && a->ldr % 4 == 0, GV_E_ALIGN, "gv_linear_ln_fwd: leading dimensions misaligned");
    GV_REQUIRE(gv_aligned(a->A, 16) && gv_aligned(a->W, 16) && gv_aligned(a->out, 16), GV_E_ALIGN, "gv_linear_ln_fwd: A/W/out must be 16-byte aligned");
    if (a->bias) GV_REQUIRE(gv_aligned(a->bias, 16), GV_E_ALIGN, "gv_linear_ln_fwd: bias misaligned");
    if (a->resid) GV_REQUIRE(gv_aligned(a->resid, 16), GV_E_ALIGN, "gv_linear_ln_fwd: resid misaligned");
    if (a->gamma) GV_REQUIRE(a->beta && a->y && a->mean && a->rstd, GV_E_NULL, "gv_linear_ln_fwd: gamma given, so beta / y / mean / rstd are required");
    PanelP p{};
    p.A = (const bf16*)a->A; p.W = (const bf16*)a->W; p.M = a->M; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw;
    p.bias = a->bias; p.resid = a->resid; p.ldr = a->ldr; p.out = a->out; p.ldo = a->ldo;
    p.gamma = a->gamma; p.beta = a->beta; p.eps = a->eps; p.y = (bf16*)a->y; p.mean = a->mean; p.rstd = a->rstd;
    p.row_scale = a->row_scale;
    return dispatch_fm<false, MODE_FWD>(p, (hipStream_t)stream, "gv_linear_ln_fwd");
}

// mlp.fc1 -> GELU -> mlp.fc2 -> + residual -> the next LayerNorm in ONE launch (vit.pyc@L98-104, L146-152) for passes that keep no
// activations (the DINO teacher, inference): the [M, hidden] activation never exists in HBM.  112-row panels (64 for short M).
extern "C" int gv_mlp_ln_fwd(const gv_mlp_ln_fwd_args* a, void* stream) {
    GV_REQUIRE(a && a->A && a->W1 && a->W2 && a->bias1 && a->out, GV_E_NULL, "gv_mlp_ln_fwd: null operand");
    GV_REQUIRE(a->N == PN, GV_E_UNSUPPORTED, "gv_mlp_ln_fwd: built for N = %d output columns (ViT-S), got %d", PN, a->N);
    GV_REQUIRE(a->M > 0 && a->K > 0 && a->K % 64 == 0 && a->hidden > 0 && a->hidden % 256 == 0, GV_E_SHAPE,
               "gv_mlp_ln_fwd: need M > 0, K %% 64 == 0, hidden %% 256 == 0 (got M=%d K=%d hidden=%d)", a->M, a->K, a->hidden);
    GV_REQUIRE((long)a->hidden * a->ldw1 < (1L << 31), GV_E_SHAPE, "gv_mlp_ln_fwd: fc1 weight too large for 32-bit element offsets");
    GV_REQUIRE(a->lda % 8 == 0 && a->ldw1 % 8 == 0 && a->ldw2 % 8 == 0 && a->ldo % 2 == 0 && a->ldr % 4 == 0, GV_E_ALIGN, "gv_mlp_ln_fwd: leading dimensions misaligned");
    GV_REQUIRE(gv_aligned(a->A, 16) && gv_aligned(a->W1, 16) && gv_aligned(a->W2, 16) && gv_aligned(a->out, 16) && gv_aligned(a->bias1, 16), GV_E_ALIGN,
               "gv_mlp_ln_fwd: A / W1 / W2 / out / bias1 must be 16-byte aligned");
    if (a->bias2) GV_REQUIRE(gv_aligned(a->bias2, 16), GV_E_ALIGN, "gv_mlp_ln_fwd: bias2 misaligned");
    if (a->resid) GV_REQUIRE(gv_aligned(a->resid, 16), GV_E_ALIGN, "gv_mlp_ln_fwd: resid misaligned");
    if (a->gamma) GV_REQUIRE(a->beta && a->y && a->mean && a->rstd, GV_E_NULL, "gv_mlp_ln_fwd: gamma given, so beta / y / mean / rstd are required");
    PanelP p{};
    p.A = (const bf16*)a->A; p.W = (const bf16*)a->W1; p.M = a->M; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw1;
    p.W2 = (const bf16*)a->W2; p.ldw2 = a->ldw2; p.bias1 = a->bias1; p.hidden = a->hidden;
    p.bias = a->bias2; p.resid = a->resid; p.ldr = a->ldr; p.out = a->out; p.ldo = a->ldo;
    p.gamma = a->gamma; p.beta = a->beta; p.eps = a->eps; p.y = (bf16*)a->y; p.mean = a->mean; p.rstd = a->rstd;
    p.row_scale = a->row_scale;
    hipStream_t s = (hipStream_t)stream;
    if (a->M <= gv_cu_budget() * 64) return launch_panel<4, 8, 64, false, MODE_FWD, EP_MLP>(p, s, "gv_mlp_ln_fwd");
    return launch_panel<7, 8, 64, false, MODE_FWD, EP_MLP>(p, s, "gv_mlp_ln_fwd");
}

extern "C" int gv_linear_ln_bwd(const gv_linear_ln_bwd_args* a, void* stream) {
    GV_REQUIRE(a && a->A && a->W && a->x && a->mean && a->rstd && a->gamma && a->g && a->partials, GV_E_NULL, "gv_linear_ln_bwd: null operand");
    GV_REQUIRE(a->N == PN, GV_E_UNSUPPORTED, "gv_linear_ln_bwd: built for N = %d output columns (ViT-S), got %d", PN, a->N);
    GV_REQUIRE(a->M > 0 && a->K > 0 && a->K % 64 == 0, GV_E_SHAPE, "gv_linear_ln_bwd: need M > 0 and K %% 64 == 0 (got M=%d K=%d)", a->M, a->K);
    GV_REQUIRE(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->ldx % 2 == 0 && a->ldg % 2 == 0 && a->ldgb % 2 == 0, GV_E_ALIGN,
               "gv_linear_ln_bwd: leading dimensions misaligned");
    GV_REQUIRE(gv_aligned(a->A, 16) && gv_aligned(a->W, 16) && gv_aligned(a->x, 8) && gv_aligned(a->g, 8), GV_E_ALIGN, "gv_linear_ln_bwd: misaligned pointer");
    GV_REQUIRE(gv_linear_ln_blocks(a->M) <= a->partial_blocks, GV_E_SHAPE, "gv_linear_ln_bwd: partials holds %d blocks, %d needed (gv_linear_ln_blocks)",
               a->partial_blocks, gv_linear_ln_blocks(a->M));
    PanelP p{};
    p.A = (const bf16*)a->A; p.W = (const bf16*)a->W; p.M = a->M; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw;
    p.x = a->x; p.ldx = a->ldx; p.mean = (float*)a->mean; p.rstd = (float*)a->rstd; p.gamma = a->gamma;
    p.g = a->g; p.ldg = a->ldg; p.gb = (bf16*)a->gb; p.ldgb = a->ldgb; p.partials = a->partials; p.g_init = a->g_init;
    p.gb_scale = a->gb_scale;
    return dispatch_fm<true, MODE_BWD>(p, (hipStream_t)stream, "gv_linear_ln_bwd");
}
