// Weight-gradient product dW = dY^T X over tens of thousands of token rows: 8-wave ping-pong kernel (included by gemm.hip).
//
// Replaces, for the hot shapes, the 128x128-tile split-K kernel of gemm_core.h behind gv_linear(trans_a, trans_b, ACCUM):
// the reference's `loss.backward()` weight gradients of attn.qkv / attn.proj / mlp.fc1 / mlp.fc2 (train.py:1071).
//
// Both operands are row-major with the REDUCTION index (token) as the row: P [K x Pn] is cut in 128-column tiles, Q [K x Qn]
// in 384-column tiles; one 512-thread workgroup per CU owns one (128 x 384) product tile over one k-slice and leaves it in
// the split-K slab (gemm.hip's reduce kernel adds the slices into C).  With the k-strided images of gemm_core.h
// ([64 k][128 cols] bf16, 32-B chunk XOR f(k), fragments by ds_read_b64_tr_b16) a 64-deep K-tile is four 16-KiB images
// (P | Q0 | Q1 | Q2), two stages = 128 KiB of LDS.
//
// Schedule (cdna_hip_programming.md "256^2 8-phase template", re-derived for a 3-image Q operand):
//   * waves (wm, wn) = (wave >> 2, wave & 3): wave tile 64 (P) x 96 (Q) = 4 x 6 fragments; in Q image b the wave owns
//     fragments {2 wn, 2 wn + 1}, so a K-tile is THREE phases b = 0, 1, 2 of 4 x 2 x 2 = 16 MFMAs each;
//   * a phase is  [fragment reads of this phase | LDS-DMA of a later tile | counted vmcnt]  s_barrier  [16 MFMAs]  s_barrier;
//     the wm = 1 half runs ONE barrier behind the wm = 0 half (SIMD partners are waves w and w + 4), so on every SIMD one
//     wave's MFMA cluster runs beside its partner's LDS reads and DMA issue;
//   * the DMA stream is simply tile 0 pieces 0..7, tile 1 pieces 0..7, ... (piece = 1 KiB per wave: P P Q0 Q0 Q1 Q1 Q2 Q2),
//     11 pieces ahead after the prologue and 3 / 2 / 3 more per phase; vmcnt(8) / (8) / (7) -- never 0 -- retires exactly
//     the image the NEXT phase reads, every piece has been in flight for three phases by then;
//   * hazards (the guide's rules for staggered halves): an image is read one phase AFTER the wait that retired it, and is
//     restaged no earlier than two phases after its last read (derivation next to the phase code).
// Out-of-range reduction rows (the last K-tile of a slice, and the dummy tiles that keep the counted waits uniform at the
// end of the slice) are sourced from a zero page.
//
// The bias gradient (column sums of dY) rides along as MFMAs against a ones fragment: normal orientation -- dY is P, every
// workgroup owns its 128 columns, wave wn sums fragment wn of its half; swapped orientation (C = [Qn x Pn], dY is Q, every
// workgroup sees all 384 columns) -- P-tile t sums Q fragments {2 (t / 3 % 4), +1} of image t % 3 (needs >= 12 P tiles).
#pragma once
#include "gemm_core.h"
#include <type_traits>

namespace gvgemm {

struct Dw8P {
    const bf16* P; const bf16* Q;
    long ldp, ldq;
    int K, Pn, Qn;
    int tiles_p, tiles_q, ksplit, k_per_split;
    float* slab;         // [ksplit][rows][cols] f32, rows x cols = Pn x Qn (normal) or Qn x Pn (swapped)
    float* colsum;       // null, or += column sums of P (normal) / of Q (swapped)
#ifdef GV_DW8_STAMPS
    unsigned long long* dbg;
#endif
};

constexpr int DW8_IMG = 64 * 128 * 2;
constexpr int DW8_STAGE = 4 * DW8_IMG;
constexpr int DW8_LDS = 2 * DW8_STAGE;

#ifdef GV_DW8_STAMPS
#define DW8_STAMP(v) const unsigned long long v = dw8_stamp()
__device__ __forceinline__ unsigned long long dw8_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#else
#define DW8_STAMP(v)
#endif

// one (tile, k-slice) item of problem g: tile tt of g.tiles_p x g.tiles_q, slice `slice`
template <bool SWAP, int VAR>      // VAR: tuning-lab ablation bits (GV_DW8_LAB builds only; the product instantiates 0)
__device__ __forceinline__ void dw8_body(const Dw8P& g, const int slice, const int tt, GV_LDS char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li16 = lane & 15, gq = lane >> 4;
    const int tp = tt / g.tiles_q, tq = tt - tp * g.tiles_q;
    const int p0 = tp * 128, q0 = tq * 384;
    const int kbeg = slice * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nt = (kend - kbeg + 63) >> 6;
    const int nfull = (kend - kbeg) >> 6;               // K-tiles that lie inside the slice entirely

    // ---- DMA sources.  Piece pc (0, 1) of an image = reduction rows rr .. of the K-tile, 256 B each; per lane one 16-B chunk
    // whose column is the image's swizzle applied to the source (rule 21).
    unsigned voffP[2], voffQ[2]; int rrow[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
        const int rr = (wave * 2 + pc) * 4 + gq;
        const int col = (((li16 >> 1) ^ swz_t(rr)) << 4) + (li16 & 1) * 8;
        rrow[pc] = rr;
        voffP[pc] = (unsigned)(((long)rr * g.ldp + col) * 2);
        voffQ[pc] = (unsigned)(((long)rr * g.ldq + col) * 2);
    }
    const unsigned long long baseP = (unsigned long long)(g.P + (long)kbeg * g.ldp + p0);
    const unsigned long long baseQ = (unsigned long long)(g.Q + (long)kbeg * g.ldq + q0);
    const unsigned long long stepP = 128ull * (unsigned long long)g.ldp, stepQ = 128ull * (unsigned long long)g.ldq;   // bytes per K-tile
    const bf16* zsrc = (const bf16*)zero_page + li16 * 8;
    const unsigned lds0 = (unsigned)(uintptr_t)smem + wave * 2048;
    // piece X (0,1: P; 2,3: Q0; 4,5: Q1; 6,7: Q2) of K-tile u into stage STG (= u & 1)
    auto issue_fast = [&](auto Xc, auto Sc, int u) {
        constexpr int X = decltype(Xc)::value, STG = decltype(Sc)::value, img = X >> 1, pc = X & 1;
        const unsigned dst = lds0 + STG * DW8_STAGE + img * DW8_IMG + pc * 1024;
        if constexpr (img == 0) glds16_s<0>(baseP + (unsigned long long)u * stepP, voffP[pc], dst);
#ifdef GV_NT_DWX    // lab: the 384-wide operand is the saved forward activation (h, xn2, o, xn1): last use in the step
        else glds16_s<(img - 1) * 256, true>(baseQ + (unsigned long long)u * stepQ, voffQ[pc], dst);
#else
        else glds16_s<(img - 1) * 256>(baseQ + (unsigned long long)u * stepQ, voffQ[pc], dst);
#endif
    };
    // ragged last K-tile of the slice / the dummy tiles behind it: rows past the slice read zeros
    auto issue_slow = [&](auto Xc, auto Sc, int u) {
        constexpr int X = decltype(Xc)::value, STG = decltype(Sc)::value, img = X >> 1, pc = X & 1;
        const unsigned dst = lds0 + STG * DW8_STAGE + img * DW8_IMG + pc * 1024;
        unsigned vo = img == 0 ? voffP[pc] : voffQ[pc]; int rr = rrow[pc];
        asm volatile("" : "+v"(vo), "+v"(rr));        // keeps this arm's address arithmetic out of the steady-state path (hipcc hoists it otherwise)
        const char* src = img == 0 ? (const char*)(baseP + (unsigned long long)u * stepP) + vo
                                   : (const char*)(baseQ + (unsigned long long)u * stepQ) + vo + (img - 1) * 256;
        if (!(kbeg + u * 64 + rr < kend)) src = (const char*)zsrc;
        glds16(src, (GV_LDS char*)(uintptr_t)dst);
    };
    // up to three pieces X0 .. of tile u behind ONE uniform branch
    auto issue = [&](auto Sc, int u, auto X0, auto X1, auto X2) {
        constexpr bool three = decltype(X2)::value >= 0;
        if (u < nfull) { issue_fast(X0, Sc, u); issue_fast(X1, Sc, u); if constexpr (three) issue_fast(X2, Sc, u); }
        else { issue_slow(X0, Sc, u); issue_slow(X1, Sc, u); if constexpr (three) issue_slow(X2, Sc, u); }
    };
    using IN = std::integral_constant<int, -1>;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;

    // ---- fragment read addresses (loop-invariant: stage, image, MFMA k-step and the +4-row half are immediates)
    //      read_frag<T>: row kr = ks*32 + gq*8 + q, chunk (blk16 ^ swz_t(kr)) << 5, + p*8   with q = li16 >> 2, p = li16 & 3
    GV_LDS char* ra[2][4]; GV_LDS char* rb[2][2];
    {
        const int kr = gq * 8 + (li16 >> 2), sw = swz_t(kr);
#pragma unroll
        for (int stg = 0; stg < 2; ++stg) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[stg][i] = smem + stg * DW8_STAGE + kr * 256 + (((wm * 4 + i) ^ sw) << 5) + (li16 & 3) * 8;
#pragma unroll
            for (int j = 0; j < 2; ++j) rb[stg][j] = smem + stg * DW8_STAGE + kr * 256 + (((2 * wn + j) ^ sw) << 5) + (li16 & 3) * 8;
        }
    }
    auto rd = [&](GV_LDS char* a0) {
        const bf16x4 lo = GV_DS_READ_TR16(a0);
        const bf16x4 hi = GV_DS_READ_TR16((a0 + 1024));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x4 acc[3][4][2];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 csum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bf16x8 ones = bf16x8{(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
    const bool cs_on = g.colsum != nullptr && (SWAP ? (tp < 12 && wave == (tp / 3) % 4) : tq == 0);
    const int cs_img = SWAP ? tp % 3 : 1;

    // ---- prologue: tile 0 whole and tile 1 pieces 0..5 (14 pieces); P, Q0, Q1 of tile 0 (the six oldest) landed
    issue(I0{}, 0, I0{}, I1{}, I2{}); issue(I0{}, 0, I3{}, I4{}, I5{}); issue(I0{}, 0, I6{}, I7{}, IN{});
    issue(I1{}, 1, I0{}, I1{}, I2{}); issue(I1{}, 1, I3{}, I4{}, I5{});
    wait_vmcnt<8>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wm == 1 && !(VAR & 4)) __builtin_amdgcn_s_barrier();       // the second half runs one barrier late
    __builtin_amdgcn_sched_barrier(0);

    // fragments are double-buffered in registers: phase p's MFMA cluster carries the LDS reads of phase p + 1
    bf16x8 fa[2][2][4], fb[2][2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][ks][j] = rd(rb[0][j] + DW8_IMG + ks * 8192);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[0][ks][i] = rd(ra[0][i] + ks * 8192);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);              // retired here, so that hipcc's counter model enters the loop clean
#ifdef GV_DW8_STAMPS
    unsigned long long c_rd = 0, c_is = 0, c_b1 = 0, c_mf = 0, c_b2 = 0;
    DW8_STAMP(t_start);
#endif
    // Phase (t, J): tile t in stage S = t & 1, fragments in fa[FA], fb[FB].
    //   load segment:  DMA issue, counted wait          | s_barrier |
    //   MFMA segment:  16 MFMAs + the fragment reads of phase (t, J) + 1 into the other register set, lgkmcnt(0) | s_barrier
    // DMA stream position of piece (u, X) is 8 u + X.  Issue: J = 0 -> (t + 1; 6, 7), J = 1 -> (t + 2; 0, 1, 2),
    // J = 2 -> (t + 2; 3, 4, 5); after the issue 8 t + {16, 19, 22}[J] pieces are out.
    // RAW (staggered halves): an image read inside MFMA segment p must have been waited for in load segment p - 1 by the
    // wave that issued it.  Segment (t, J) + 1 reads Q1(t) / Q2(t) / Q0(t + 1), P(t + 1) for J = 2 / 0 / 1 as the waiting
    // phase, last pieces 8 t + {13, 7, 11}[J = 2, 0, 1]  ->  vmcnt 8 / 8 / 7 for J = 0 / 1 / 2.. (table below).
    // WAR: reads are retired (lgkmcnt(0)) before the segment's closing barrier; an image is restaged no earlier than two
    // phases after the segment that read it: P, Q0 (read in (u - 3, 2)) at (u - 2, 1) / (u - 2, 2), Q1 (read in (u - 2, 0))
    // at (u - 2, 2), Q2 (read in (u - 2, 1)) at (u - 1, 0) -- each exactly two phases.
    auto phase = [&](auto Jc, auto Sc, auto FAc, auto FBc, int t) {
        constexpr int J = decltype(Jc)::value, S = decltype(Sc)::value, FA = decltype(FAc)::value, FB = decltype(FBc)::value;
        using SN = std::integral_constant<int, S ^ 1>;
        DW8_STAMP(s1);
        // waits: J = 0 needs Q2(t) (8 t + 7) of 8 t + 16 out; J = 1 needs Q0(t + 1) (8 t + 11) of 8 t + 19; J = 2 needs
        // Q1(t + 1) (8 t + 13) of 8 t + 22
        if constexpr ((VAR & 8) != 0) {}
        else if constexpr (J == 0) { issue(SN{}, t + 1, I6{}, I7{}, IN{}); wait_vmcnt<8>(); }
        else if constexpr (J == 1) { issue(Sc, t + 2, I0{}, I1{}, I2{}); wait_vmcnt<7>(); }
        else { issue(Sc, t + 2, I3{}, I4{}, I5{}); wait_vmcnt<8>(); }
        DW8_STAMP(s2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        DW8_STAMP(s3);
        if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(1);
        // next phase's fragments
        if constexpr ((VAR & 16) != 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { fb[FB ^ 1][ks][j] = fb[FB][ks][j]; asm volatile("" : "+v"(fb[FB ^ 1][ks][j])); }
                if constexpr (J == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { fa[FA ^ 1][ks][i] = fa[FA][ks][i]; asm volatile("" : "+v"(fa[FA ^ 1][ks][i])); }
                }
            }
        } else if constexpr (J < 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[FB ^ 1][ks][j] = rd(rb[S][j] + (2 + J) * DW8_IMG + ks * 8192);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[FB ^ 1][ks][j] = rd(rb[S ^ 1][j] + DW8_IMG + ks * 8192);
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[FA ^ 1][ks][i] = rd(ra[S ^ 1][i] + ks * 8192);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (SWAP) acc[J][i][j] = GV_MFMA_16x16x32(fa[FA][ks][i], fb[FB][ks][j], acc[J][i][j]);
                    else acc[J][i][j] = GV_MFMA_16x16x32(fb[FB][ks][j], fa[FA][ks][i], acc[J][i][j]);
                }
        if (cs_on && cs_img == J) {
            if constexpr (SWAP) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 2; ++j) csum[j] = GV_MFMA_16x16x32(ones, fb[FB][ks][j], csum[j]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (wn == i) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) csum[0] = GV_MFMA_16x16x32(ones, fa[FA][ks][i], csum[0]);
                    }
            }
        }
        // the reads go out EARLY in the cluster, one (J = 2: two) per MFMA gap -- hipcc sinks them behind the last MFMA otherwise
        // and the closing lgkmcnt(0) then exposes the whole LDS latency
        if constexpr ((VAR & 2) != 0) {
            __builtin_amdgcn_sched_group_barrier(0x100, J == 2 ? 24 : 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        } else if (!(cs_on && cs_img == J)) {
#pragma unroll
            for (int k = 0; k < (J == 2 ? 12 : 8); ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, J == 2 ? 2 : 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, J == 2 ? 4 : 8, 0);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this segment's reads are retired before its closing barrier (WAR)
        if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(0);
        DW8_STAMP(s4);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef GV_DW8_STAMPS
        { DW8_STAMP(s5); c_is += s2 - s1; c_b1 += s3 - s2; c_mf += s4 - s3; c_b2 += s5 - s4; }
#endif
    };
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        phase(I0{}, I0{}, I0{}, I0{}, t); phase(I1{}, I0{}, I0{}, I1{}, t); phase(I2{}, I0{}, I0{}, I0{}, t);
        phase(I0{}, I1{}, I1{}, I1{}, t + 1); phase(I1{}, I1{}, I1{}, I0{}, t + 1); phase(I2{}, I1{}, I1{}, I1{}, t + 1);
    }
    if (t < nt) { phase(I0{}, I0{}, I0{}, I0{}, t); phase(I1{}, I0{}, I0{}, I1{}, t); phase(I2{}, I0{}, I0{}, I0{}, t); }
    if (wm == 0 && !(VAR & 4)) __builtin_amdgcn_s_barrier();       // pair the late half's last barrier
    wait_vmcnt<0>();                                  // the dummy tiles' DMA must not outlive the workgroup
#ifdef GV_DW8_STAMPS
    if (g.dbg && lane == 0) {
        unsigned long long* o = g.dbg + ((long)blockIdx.x * 8 + wave) * 8;
        o[0] = c_rd; o[1] = c_is; o[2] = c_b1; o[3] = c_mf; o[4] = c_b2; o[5] = dw8_stamp() - t_start; o[6] = nt;
    }
#endif

    // ---- partial tile -> slab (row-major like C; 64-B segments per 4 lanes, neighbouring fragments complete the lines)
    if constexpr (!SWAP) {
        float* sl = g.slab + (long)slice * g.Pn * g.Qn;
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int m = p0 + wm * 64 + i * 16 + li16, n = q0 + b * 128 + (2 * wn + j) * 16 + gq * 4;
                    if constexpr ((VAR & 32) != 0) asm volatile("" ::"v"(acc[b][i][j])); else
                    *(f32x4*)(sl + (long)m * g.Qn + n) = acc[b][i][j];
                }
        if (cs_on && gq == 0) atomicAdd(g.colsum + p0 + wm * 64 + wn * 16 + li16, csum[0][0]);
    } else {
        float* sl = g.slab + (long)slice * g.Pn * g.Qn;
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nq = q0 + b * 128 + (2 * wn + j) * 16 + li16, mp = p0 + wm * 64 + i * 16 + gq * 4;
                    *(f32x4*)(sl + (long)nq * g.Pn + mp) = acc[b][i][j];
                }
        if (cs_on && wm == 0 && gq == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) atomicAdd(g.colsum + q0 + cs_img * 128 + (2 * wn + j) * 16 + li16, csum[j][0]);
        }
    }
}

// workgroup b -> item: XCD-contiguous (workgroups b, b + 8, .. share an XCD and take neighbouring items, i.e. tiles of the
// same k-slice that re-read the same Q rows from that XCD's L2)
__device__ __forceinline__ int dw8_item() {
    const int nb = gridDim.x, xcd = blockIdx.x & 7, lw = blockIdx.x >> 3, qd = nb >> 3, rm = nb & 7;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + lw;
}

template <bool SWAP, int VAR = 0>
__global__ __launch_bounds__(512, 2) void dw8_kernel(const Dw8P g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int item = dw8_item(), tiles = g.tiles_p * g.tiles_q;
    const int slice = item / tiles;
    dw8_body<SWAP, VAR>(g, slice, item - slice * tiles, (GV_LDS char*)smem_raw);
}

// the weight-gradient products of one transformer block (all reduce over the same token rows) as ONE launch: items are
// slice-major over the concatenated tile lists, so a workgroup's k-slice is 1 / ksplit of the rows with ksplit = 256 / (all
// tiles) -- four times longer k-loops and a quarter of the slab traffic of four separate launches
struct Dw8GroupP { Dw8P prob[GV_DW_GROUP_MAX]; int n, total_tiles; int tile_base[GV_DW_GROUP_MAX + 1]; };

__global__ __launch_bounds__(512, 2) void dw8_group_kernel(const Dw8GroupP G) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int item = dw8_item();
    const int slice = item / G.total_tiles, tg = item - slice * G.total_tiles;
    int q = 0;
#pragma unroll
    for (int i = 1; i < GV_DW_GROUP_MAX; ++i) q += (i < G.n && tg >= G.tile_base[i]) ? 1 : 0;
    dw8_body<false, 0>(G.prob[q], slice, tg - G.tile_base[q], (GV_LDS char*)smem_raw);
}

}  // namespace gvgemm
