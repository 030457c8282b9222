// Fused multi-head attention forward / backward for short sequences (N <= 288,
// head_dim 64) -- replaces Attention.forward's q@k^T, softmax, attn@v ATen calls
// (vit.pyc@L119-131) and their autograd backward; the N x N score matrix never
// reaches HBM.  A whole sequence fits one workgroup's LDS.
//
// Common LDS image: [rows = tokens][64 d] bf16, 128-B rows, 16-B chunk index XOR
// (row & 7).  Filled by LDS-DMA with the swizzle on the per-lane SOURCE address.  The
// same image serves k-contiguous fragment reads (ds_read_b128) and hardware-transposed
// reads (ds_read_b64_tr_b16: 4 rows x 16 columns per 16-lane group), both conflict-free.
//
// forward : a wave owns 32 queries.  S^T = K Q^T (keys on MFMA rows, queries on lanes) so
//           the row softmax is an in-lane reduction + 2 wavefront shuffles, and P^T is
//           already the B operand of O^T = V^T P^T (accumulator-as-operand, no LDS trip).
// backward: a wave owns 32 keys.  S, dP with keys on lanes; P, dS are then already the B
//           operands of dV^T += dO^T P and dK^T += Q^T dS, so dK/dV need no cross-wave
//           sum.  Only dS crosses LDS (transposed image) for dQ = dS K.
#include "gv_common.h"
#include <type_traits>

#ifdef GV_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[64 * 16];
#define AST(i) do { if (blockIdx.x < 64 && threadIdx.x == 0) g_attn_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int gv_lab_attn_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_stamps), sizeof(g_attn_stamps)); }
#else
#define AST(i)
#endif
namespace {

__device__ __attribute__((aligned(256))) unsigned short attn_zero_page[128];

__device__ __forceinline__ void glds16(const void* src, GV_LDS char* dst) {
    __builtin_amdgcn_global_load_lds((const GV_GLOBAL void*)src, (GV_LDS void*)dst, 16, 0, 0);
}

// stage rows [0, nrows_pad) of a [token][64] slice; rows >= nvalid are zero filled
__device__ __forceinline__ void stage_rows(const bf16* __restrict__ gbase, long ld, int nvalid, int nrows_pad,
                                           GV_LDS char* img, int wave, int nwaves, int lane) {
    const int pieces = nrows_pad >> 3;
    for (int piece = wave; piece < pieces; piece += nwaves) {
        const int r = piece * 8 + (lane >> 3);
        const int slot = lane & 7;
        const int c = slot ^ (r & 7);
        const bf16* src = r < nvalid ? gbase + (long)r * ld + c * 8 : (const bf16*)attn_zero_page + slot * 8;
        glds16(src, img + __builtin_amdgcn_readfirstlane(piece * 1024));
    }
}

__device__ __forceinline__ bf16x8 read_nat(GV_LDS char* img, int row, int chunk) {
    return *(GV_LDS bf16x8*)(img + row * 128 + ((chunk ^ (row & 7)) << 4));
}
// transposed read: this lane addresses `row`, 16-column block dt, quarter p (0..3)
__device__ __forceinline__ bf16x4 read_tr(GV_LDS char* img, int row, int dt, int p) {
    const int c16 = 2 * dt + (p >> 1);
    return GV_DS_READ_TR16(img + row * 128 + ((c16 ^ (row & 7)) << 4) + 8 * (p & 1));
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
    return bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}
template <int LO, int HI, class F>
__device__ __forceinline__ void attn_static_for(F&& f) {
    if constexpr (LO < HI) { f(std::integral_constant<int, LO>{}); attn_static_for<LO + 1, HI>(f); }
}
#define MFMA16(a, b, c) GV_MFMA_16x16x32((a), (b), (c))

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
template <int NKT>
struct FwdCfg {
    // long sequences: 16 queries per wave and round (the score registers halve, two 8-wave workgroups
    // fit a CU and cover each other's load phases); short ones: 32 queries per wave
#ifdef GV_LAB_ATTN_QT2       // lab: 32 queries per wave at every length (half the K / V fragment reads per query, one workgroup per CU)
    static constexpr int QT = 2;
#else
    static constexpr int QT = NKT >= 14 ? 1 : 2;                  // 16-query tiles per wave and round
#endif
    static constexpr int NQB = (NKT + QT - 1) / QT;               // query blocks of 16 QT rows
    static constexpr int NW = NQB >= 5 ? 8 : 4;                   // waves per workgroup
    static constexpr int PAIRS = NQB >= NW ? 1 : (NW / NQB >= 1 ? NW / NQB : 1);
    static constexpr int WPP = NW / PAIRS;                        // waves per (image, head) pair
    static constexpr int ROUNDS = (NQB + WPP - 1) / WPP;          // query blocks per wave
};

// One workgroup's worth of forward work: `block` = index among the workgroups of this length class, `wave` = wave index inside
// the NW waves that run it, `smem` = their LDS.  (A stand-alone launch passes blockIdx.x and the whole workgroup; the varlen
// launch runs two 4-wave instances of the short class side by side in one 8-wave workgroup.)
template <int NKT>
__device__ __forceinline__ void attn_fwd_body(const gv_attention_fwd_args a, const int n_pairs, const int block, const int wave, const int lane, GV_LDS char* const smem) {
    using F = FwdCfg<NKT>;
    constexpr int NQB = F::NQB, PAIRS = F::PAIRS, WPP = F::WPP, NW = F::NW, ROUNDS = F::ROUNDS, QT = F::QT, QB = 16 * QT;
    constexpr int NP = NKT * 16;
    constexpr int IMG = NP * 128;
    const int N = a.N, H = a.H;
    const long ld = 3L * H * 64;
    const bf16* qkv = (const bf16*)a.qkv;

    for (int pr = 0; pr < PAIRS; ++pr) {
        int pair = block * PAIRS + pr;
        pair = pair < n_pairs ? pair : n_pairs - 1;
        const int img = pair / H, h = pair - img * H;
        const bf16* base = qkv + (long)img * N * ld + h * 64;
        stage_rows(base + H * 64, ld, N, NP, smem + (pr * 2 + 0) * IMG, wave, NW, lane);
        stage_rows(base + 2 * H * 64, ld, N, NP, smem + (pr * 2 + 1) * IMG, wave, NW, lane);
    }
    const int lp = wave / WPP, wq = wave % WPP;
    const int pair_raw = block * PAIRS + lp;
    const bool valid = pair_raw < n_pairs;
    const int pair = valid ? pair_raw : n_pairs - 1;
    const int img = pair / H, h = pair - img * H;
    const bf16* qbase = qkv + (long)img * N * ld + h * 64;
    GV_LDS char* Kimg = smem + (lp * 2 + 0) * IMG;
    GV_LDS char* Vimg = smem + (lp * 2 + 1) * IMG;
    const int li = lane & 15, g = lane >> 4, q4 = li >> 2, p4 = li & 3;
    const float c = a.scale * 1.4426950408889634f;
    // q_limit: only the first query rows are wanted -- whole 32-row groups, so that the backward with the same limit finds o / lse
    // valid for every row its first half-step touches
    const int QE = (a.q_limit > 0 && ((a.q_limit + 31) & ~31) < N) ? ((a.q_limit + 31) & ~31) : N;

    // this wave's Q fragments for all its query blocks: issued BEFORE the staging wait so the
    // global-load latencies of Q, K and V overlap
    bf16x8 qf[ROUNDS][QT][2];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int qb = wq + rd * WPP;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            int qrow = qb * QB + qt * 16 + li;
            qrow = qrow < N ? qrow : N - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[rd][qt][ks] = *(const bf16x8*)(qbase + (long)qrow * ld + ks * 32 + g * 8);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // (rounds unrolled by hand: left to `#pragma unroll` + break, the loop survives whenever this body is inlined into a second
    //  kernel, and the register allocation of the N = 197 class jumps from 112 to 227)
    auto round = [&](auto RDc) {
        constexpr int rd = decltype(RDc)::value;
        const int qb = wq + rd * WPP;
        if (qb >= NQB || qb * QB >= QE) return;
        f32x4 s[NKT][QT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 kf = read_nat(Kimg, kt * 16 + li, ks * 4 + g);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) s[kt][qt] = MFMA16(kf, qf[rd][qt][ks], s[kt][qt]);
            }
        }
        // softmax over keys: key = kt*16 + 4g + r lives in (kt, r) of lanes {li, li+16, li+32, li+48}
        float mx[QT], sum[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) { mx[qt] = -INFINITY; sum[qt] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = kt * 16 + 4 * g + r < N;
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const float v = ok ? s[kt][qt][r] : -INFINITY;
                    s[kt][qt][r] = v;
                    mx[qt] = fmaxf(mx[qt], v);
                }
            }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 16, 64));
            mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 32, 64));
        }
        float mxc[QT];                                    // exp2((s - mx) c) as one fma + exp2 per element
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mxc[qt] = -mx[qt] * c;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][qt][r], c, mxc[qt]));
                    s[kt][qt][r] = p;
                    sum[qt] += p;
                }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            sum[qt] += __shfl_xor(sum[qt], 16, 64);
            sum[qt] += __shfl_xor(sum[qt], 32, 64);
        }
        // O^T[d][q] = sum_key V[key][d] P^T[key][q]
        f32x4 o[4][QT];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NKT / 2; ++u) {
            bf16x8 pf[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) pf[qt] = pack8(s[2 * u][qt], s[2 * u + 1][qt]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 vf = cat8(read_tr(Vimg, (2 * u) * 16 + 4 * g + q4, dt, p4),
                                       read_tr(Vimg, (2 * u + 1) * 16 + 4 * g + q4, dt, p4));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[dt][qt] = MFMA16(vf, pf[qt], o[dt][qt]);
            }
        }
        if (valid) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int q = qb * QB + qt * 16 + li;
                if (q < N) {
                    const float inv = 1.0f / sum[qt];
                    bf16* dst = (bf16*)a.o + ((long)img * N + q) * (H * 64) + h * 64 + 4 * g;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        *(bf16x4*)(dst + dt * 16) = bf16x4{(bf16)(o[dt][qt][0] * inv), (bf16)(o[dt][qt][1] * inv),
                                                           (bf16)(o[dt][qt][2] * inv), (bf16)(o[dt][qt][3] * inv)};
                    if (g == 0) a.lse[((long)img * H + h) * N + q] = mx[qt] * a.scale + __logf(sum[qt]);
                }
            }
        }
    };
    attn_static_for<0, ROUNDS>(round);
}

// (second launch-bounds argument = waves per SIMD the register allocation must leave room for: two 8-wave workgroups per CU = 4)
template <int NKT>
__global__ __launch_bounds__(FwdCfg<NKT>::NW * 64, FwdCfg<NKT>::QT == 1 && NKT <= 14 ? 4 : (FwdCfg<NKT>::QT == 1 ? 2 : 1)) void attn_fwd_kernel(gv_attention_fwd_args a, int n_pairs) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    attn_fwd_body<NKT>(a, n_pairs, blockIdx.x, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63, (GV_LDS char*)smem_raw);
}

// Varlen forward: the token-concatenated row space of a multi-crop pass holds a long segment (N <= 224) and a short one
// (N <= 64) -- ONE launch serves both: workgroups [0, nblk_long) run the long class (one pair, 8 waves), the rest run TWO
// short-class instances side by side (waves 0-3 and 4-7, two pairs each, their own LDS halves).  64 KB of LDS and <= 128
// registers either way: two workgroups per CU.  For the step's student pass (768 long + 3 072 short pairs) that is 768 + 768
// workgroups = exactly three rounds of the 512 slots, where the long launch alone filled one and a half.
// (Every instance passes the same workgroup barriers: the bodies' barrier count does not depend on the data.)
constexpr int VL_LDS_LONG = FwdCfg<14>::PAIRS * 2 * 14 * 16 * 128, VL_LDS_SHORT = FwdCfg<4>::PAIRS * 2 * 4 * 16 * 128;
constexpr int VL_LDS = VL_LDS_LONG > 2 * VL_LDS_SHORT ? VL_LDS_LONG : 2 * VL_LDS_SHORT;
static_assert(FwdCfg<14>::NW == 8 && FwdCfg<4>::NW == 4, "varlen launch: 8-wave long class, two 4-wave short instances");
__global__ __launch_bounds__(512, 4) void attn_fwd_varlen_kernel(gv_attention_fwd_args al, int np_long, int nblk_long, gv_attention_fwd_args as, int np_short) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    GV_LDS char* smem = (GV_LDS char*)smem_raw;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if ((int)blockIdx.x < nblk_long) {
        attn_fwd_body<14>(al, np_long, blockIdx.x, wave, lane, smem);
    } else {
        const int sub = wave >> 2;
        attn_fwd_body<4>(as, np_short, ((int)blockIdx.x - nblk_long) * 2 + sub, wave & 3, lane, smem + sub * VL_LDS_SHORT);
    }
}

// ---------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------
// KT = 16-key tiles per wave.  KT = 2: 32 keys per wave, ~190 registers, 2 waves per SIMD.  KT = 1: 16 keys per wave, the same
// chain on half the keys in < 128 registers, 4 waves per SIMD -- but every wave still reads all of Q and dO from LDS, so the
// workgroup's LDS read traffic of phase A doubles (measured: the short crops gain 20 %, N = 197 gains nothing; see gv_attention_bwd).
template <int NKT, int KT>
struct BwdCfg {
    static constexpr int NKB = NKT / KT;               // key blocks of 16 KT keys = waves per pair
    static constexpr int PAIRS = NKB >= 4 ? 1 : 4 / NKB;
    static constexpr int NW = NKB * PAIRS;             // waves per workgroup
    // dS^T images.  2 where a (image, head) pair holds its CU alone anyway (N = 197: 142 KB instead of 114): the image alternates per
    // 64-query step and the step's second barrier goes (72.9 -> 70.4 us per launch; the short classes, several workgroups per CU, keep 1)
    static constexpr int NDS = NKT == 14 ? 2 : 1;
};
template <int NKT, int KT>
// short sequences run two 4-wave workgroups per CU: the second launch-bounds argument keeps them within 256 registers
__global__ __launch_bounds__((BwdCfg<NKT, KT>::NW * 64), ((NKT >= 14 || BwdCfg<NKT, KT>::NW > 8) ? 1 : 2)) void attn_bwd_kernel(gv_attention_bwd_args a, int n_pairs) {
    constexpr int NKB = BwdCfg<NKT, KT>::NKB, PAIRS = BwdCfg<NKT, KT>::PAIRS, NW = BwdCfg<NKT, KT>::NW;
    constexpr int KW = 16 * KT;                        // keys per wave
    constexpr int NK32 = NKT / 2;                      // 32-key slices of the dQ reduction
    constexpr int NP = NKT * 16;
    constexpr int IMG = NP * 128;
    constexpr int QH = NKT >= 8 ? 2 : 1;               // 32-query halves per barrier pair (short sequences keep 1)
    constexpr int DROW = 64 * QH;                      // dS^T image: [key][32 QH q] bf16
    constexpr int DST = NP * DROW;
    constexpr int NDS = BwdCfg<NKT, KT>::NDS;          // dS^T images (2: the image alternates per step and the step's second barrier goes)
    constexpr int PER_PAIR = 3 * IMG + NDS * DST + 2 * NP * 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    GV_LDS char* smem = (GV_LDS char*)smem_raw;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N, H = a.H;
    const long ld = 3L * H * 64, ldo = (long)H * 64;
    const bf16* qkv = (const bf16*)a.qkv;
    AST(0);

    // ---- stage Q, K, dO of every pair; delta and lse into LDS
    for (int pr = 0; pr < PAIRS; ++pr) {
        int pair = blockIdx.x * PAIRS + pr;
        pair = pair < n_pairs ? pair : n_pairs - 1;
        const int img = pair / H, h = pair - img * H;
        const bf16* base = qkv + (long)img * N * ld + h * 64;
        GV_LDS char* P0 = smem + pr * PER_PAIR;
        stage_rows(base, ld, N, NP, P0, wave, NW, lane);
        stage_rows(base + H * 64, ld, N, NP, P0 + IMG, wave, NW, lane);
        stage_rows((const bf16*)a.d_o + (long)img * N * ldo + h * 64, ldo, N, NP, P0 + 2 * IMG, wave, NW, lane);
    }
    // delta[q] = sum_d O[q][d] dO[q][d] and lse[q], one query per thread (NW * 64 >= PAIRS * NP for every NKT): the O row is
    // fetched into registers now, the dO row is taken from the LDS image once it has landed -- dO makes ONE trip from HBM
    static_assert(NW * 64 >= PAIRS * NP, "one (pair, query) per thread");
    const int dq_pr = threadIdx.x / NP, dq_q = threadIdx.x - dq_pr * NP;
    const bool dq_has = (int)threadIdx.x < PAIRS * NP, dq_live = dq_has && dq_q < N;
    // (the 9-wave N <= 288 build has 168 registers per lane: there the row is multiplied against dO from global straight away)
    constexpr bool DELTA_LDS = NKT < 18;
    bf16x8 orow8[DELTA_LDS ? 8 : 1];
    float lse_q = 0.f, delta_q = 0.f;
    if (dq_live) {
        int pair = blockIdx.x * PAIRS + dq_pr;
        pair = pair < n_pairs ? pair : n_pairs - 1;
        const int img = pair / H, h = pair - img * H;
        const bf16* orow = (const bf16*)a.o + ((long)img * N + dq_q) * ldo + h * 64;
        if constexpr (DELTA_LDS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) orow8[i] = *(const bf16x8*)(orow + i * 8);
        } else {
            const bf16* drow = (const bf16*)a.d_o + ((long)img * N + dq_q) * ldo + h * 64;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf16x8 x = *(const bf16x8*)(orow + i * 8), y = *(const bf16x8*)(drow + i * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) delta_q += (float)x[j] * (float)y[j];
            }
        }
        lse_q = a.lse[((long)img * H + h) * N + dq_q];
    }
    const int lp = wave / NKB, kb = wave % NKB;        // local pair, this wave's key block
    const int pair_raw = blockIdx.x * PAIRS + lp;
    const bool valid = pair_raw < n_pairs;
    const int pair = valid ? pair_raw : n_pairs - 1;
    const int img = pair / H, h = pair - img * H;
    GV_LDS char* Qimg = smem + lp * PER_PAIR;
    GV_LDS char* Kimg = Qimg + IMG;
    GV_LDS char* Dimg = Qimg + 2 * IMG;
    GV_LDS char* const dsT0 = Qimg + 3 * IMG;
    GV_LDS float* delta = (GV_LDS float*)(dsT0 + NDS * DST);
    GV_LDS float* lse = delta + NP;
    const int li = lane & 15, g = lane >> 4, q4 = li >> 2, p4 = li & 3;
    const float c = a.scale * 1.4426950408889634f;

    // this wave's V fragments come straight from global: issued before the staging wait so
    // their latency overlaps the LDS-DMA of Q, K, dO and the delta pass
    bf16x8 kf[KT][2], vf[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int key = kb * KW + kt * 16 + li;
        const int keyc = key < N ? key : N - 1;
        const bf16* vrow = qkv + ((long)img * N + keyc) * ld + 2 * H * 64 + h * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) vf[kt][ks] = *(const bf16x8*)(vrow + ks * 32 + g * 8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    AST(1);
    if (dq_has) {
        GV_LDS char* P0 = smem + dq_pr * PER_PAIR;
        GV_LDS float* dl = (GV_LDS float*)(P0 + 3 * IMG + NDS * DST);
        float d = delta_q;
        if constexpr (DELTA_LDS) {
            if (dq_live) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bf16x8 x = orow8[i], y = read_nat(P0 + 2 * IMG, dq_q, i);
#pragma unroll
                    for (int j = 0; j < 8; ++j) d += (float)x[j] * (float)y[j];
                }
            }
        }
        dl[dq_q] = d;
        dl[NP + dq_q] = lse_q;
    }
    __syncthreads();
    AST(2);

    // this wave's K fragments (B operands: lane = key, 8 consecutive d)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int key = kb * KW + kt * 16 + li;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[kt][ks] = read_nat(Kimg, key, ks * 4 + g);
    }
    f32x4 dv[4][KT], dk[4][KT];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) { dv[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // 64 queries per barrier pair: phase A runs twice (two 32-query halves, registers as for one) and
    // fills both halves of the dS^T image, phase B then has 16 dQ tiles to spread over the waves
    // q_limit: d_o is zero behind the first q_limit query rows -> whole 32-query halves behind them contribute nothing to dK / dV
    // and have dQ = 0: they are skipped (their dQ rows are zero-filled at the end)
    const int QE = (a.q_limit > 0 && ((a.q_limit + 31) & ~31) < N) ? ((a.q_limit + 31) & ~31) : N;
    const int nqc2 = (QE + 32 * QH - 1) / (32 * QH);
    for (int qc2 = 0; qc2 < nqc2; ++qc2) {
        // NDS = 2: step i writes image i & 1.  Its readers (phase B of step i) come before phase A of step i + 1 in every wave's
        // program order, hence before the barrier of step i + 1 -- and image i & 1 is next written in step i + 2, behind that barrier
        GV_LDS char* const dsT = dsT0 + (NDS == 2 ? (qc2 & 1) * DST : 0);
#pragma unroll
        for (int half = 0; half < QH; ++half) {
            const int qc = qc2 * QH + half;
            if (qc * 32 >= QE) break;
            // ---- phase A: S, dP for [32 q] x [this wave's keys]
            f32x4 s[2][KT], dp[2][KT];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int qrow = qc * 32 + qt * 16 + li;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) { s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 qa = read_nat(Qimg, qrow, ks * 4 + g);
                    const bf16x8 da = read_nat(Dimg, qrow, ks * 4 + g);
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        s[qt][kt] = MFMA16(qa, kf[kt][ks], s[qt][kt]);
                        dp[qt][kt] = MFMA16(da, vf[kt][ks], dp[qt][kt]);
                    }
                }
            }
            // P = exp(scale*S - lse[q]); dS = P * (dP - delta[q]) * scale.  rows q = 4g + r, col key = li
            f32x4 pv[2][KT], ds[2][KT];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int q0 = qc * 32 + qt * 16 + 4 * g;
                // per row, once for both key tiles: -lse * log2(e) and -delta * scale  (one fma / exp2 / select / fma / mul per element)
                f32x4 l4 = *(GV_LDS f32x4*)(lse + q0), d4 = *(GV_LDS f32x4*)(delta + q0);
                l4 *= -1.4426950408889634f;
                d4 *= -a.scale;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const bool kok = kb * KW + kt * 16 + li < N;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = kok && (q0 + r < N);
                        const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[qt][kt][r], c, l4[r])) : 0.f;
                        pv[qt][kt][r] = p;
                        ds[qt][kt][r] = p * fmaf(dp[qt][kt][r], a.scale, d4[r]);
                    }
                }
            }
            // dV^T += dO^T P ; dK^T += Q^T dS   (reduction over the chunk's 32 queries)
            {
                bf16x8 pb[KT], sb[KT];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) { pb[kt] = pack8(pv[0][kt], pv[1][kt]); sb[kt] = pack8(ds[0][kt], ds[1][kt]); }
                const int r0 = qc * 32 + 4 * g + q4;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 dot = cat8(read_tr(Dimg, r0, dt, p4), read_tr(Dimg, r0 + 16, dt, p4));
                    const bf16x8 qtt = cat8(read_tr(Qimg, r0, dt, p4), read_tr(Qimg, r0 + 16, dt, p4));
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        dv[dt][kt] = MFMA16(dot, pb[kt], dv[dt][kt]);
                        dk[dt][kt] = MFMA16(qtt, sb[kt], dk[dt][kt]);
                    }
                }
            }
            // dS^T image [key][32 QH q]: 32-B piece index XOR f(key) (conflict-free transposed reads)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int key = kb * KW + kt * 16 + li;
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    *(GV_LDS bf16x4*)(dsT + key * DROW + (((half * 2 + qt) ^ (QH == 2 ? (key >> 1) & 3 : (key >> 2) & 1)) << 5) + g * 8) =
                        bf16x4{(bf16)ds[qt][kt][0], (bf16)ds[qt][kt][1], (bf16)ds[qt][kt][2], (bf16)ds[qt][kt][3]};
            }
        }
        AST(3 + 3 * qc2);
        __syncthreads();
        AST(4 + 3 * qc2);
        // ---- phase B: dQ^T[d][q] = sum_key K[key][d] dS[q][key]; 16 (qt, dt) tiles over the pair's waves
        for (int tile = kb; tile < 8 * QH; tile += NKB) {
            const int qt = tile >> 2, dt = tile & 3;
            if (qc2 * 32 * QH + qt * 16 >= QE) continue;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NK32; ++ks) {
                // k-slot (g, j): key = 32 ks + 16 (j >> 2) + 4 g + (j & 3)
                const int k0 = ks * 32 + 4 * g + q4;
                const bf16x8 ka = cat8(read_tr(Kimg, k0, dt, p4), read_tr(Kimg, k0 + 16, dt, p4));
                const int hq = qt ^ (QH == 2 ? (k0 >> 1) & 3 : (k0 >> 2) & 1);          // same for key k0 + 16
                const bf16x4 lo = GV_DS_READ_TR16((dsT + k0 * DROW + (hq << 5) + p4 * 8));
                const bf16x4 hi = GV_DS_READ_TR16((dsT + (k0 + 16) * DROW + (hq << 5) + p4 * 8));
                acc = MFMA16(ka, cat8(lo, hi), acc);
            }
            const int q = qc2 * 32 * QH + qt * 16 + li;
            if (valid && q < N)
                *(bf16x4*)((bf16*)a.dqkv + ((long)img * N + q) * ld + h * 64 + dt * 16 + 4 * g) =
                    bf16x4{(bf16)acc[0], (bf16)acc[1], (bf16)acc[2], (bf16)acc[3]};
        }
        AST(5 + 3 * qc2);
        if constexpr (NDS == 1) __syncthreads();
    }
    if (valid) {        // skipped queries: dQ = 0 (4 rows x 128 B per wave-instruction)
        for (int q = QE + kb * 4 + g; q < N; q += NKB * 4)
            *(bf16x4*)((bf16*)a.dqkv + ((long)img * N + q) * ld + h * 64 + li * 4) = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    }
    // ---- dK, dV: lane = key, rows d = 16 dt + 4 g + r
    if (valid) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int key = kb * KW + kt * 16 + li;
            if (key < N) {
                bf16* dst = (bf16*)a.dqkv + ((long)img * N + key) * ld + h * 64 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    *(bf16x4*)(dst + H * 64 + dt * 16) = bf16x4{(bf16)dk[dt][kt][0], (bf16)dk[dt][kt][1], (bf16)dk[dt][kt][2], (bf16)dk[dt][kt][3]};
                    *(bf16x4*)(dst + 2 * H * 64 + dt * 16) = bf16x4{(bf16)dv[dt][kt][0], (bf16)dv[dt][kt][1], (bf16)dv[dt][kt][2], (bf16)dv[dt][kt][3]};
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AST(15);
}

template <int NKT> int launch_fwd(const gv_attention_fwd_args* a, hipStream_t s) {
    constexpr int PAIRS = FwdCfg<NKT>::PAIRS;
    constexpr int LDS = PAIRS * 2 * NKT * 16 * 128;
    auto kern = attn_fwd_kernel<NKT>;
    static GvLdsOptIn opt_in;
    if (int rc = gv_lds_opt_in(opt_in, (const void*)kern, LDS, "gv_attention_fwd")) return rc;
    const int n_pairs = a->n_img * a->H;
    hipLaunchKernelGGL(kern, dim3((n_pairs + PAIRS - 1) / PAIRS), dim3(FwdCfg<NKT>::NW * 64), LDS, s, *a, n_pairs);
    GV_LAUNCH_CHECK("gv_attention_fwd");
    return GV_OK;
}

template <int NKT, int KT> int launch_bwd(const gv_attention_bwd_args* a, hipStream_t s) {
    using B = BwdCfg<NKT, KT>;
    constexpr int PAIRS = B::PAIRS, NW = B::NW, NP = NKT * 16;
    constexpr int LDS = PAIRS * (3 * NP * 128 + B::NDS * NP * (NKT >= 8 ? 128 : 64) + 2 * NP * 4);
    static_assert(NW * 64 <= 1024 && LDS <= 160 * 1024, "workgroup size / LDS");
    auto kern = attn_bwd_kernel<NKT, KT>;
    static GvLdsOptIn opt_in;
    if (int rc = gv_lds_opt_in(opt_in, (const void*)kern, LDS, "gv_attention_bwd")) return rc;
    const int n_pairs = a->n_img * a->H;
    hipLaunchKernelGGL(kern, dim3((n_pairs + PAIRS - 1) / PAIRS), dim3(NW * 64), LDS, s, *a, n_pairs);
    GV_LAUNCH_CHECK("gv_attention_bwd");
    return GV_OK;
}

int launch_fwd_any(const gv_attention_fwd_args* a, hipStream_t s) {
    if (a->N <= 32) return launch_fwd<2>(a, s);
    if (a->N <= 64) return launch_fwd<4>(a, s);
    if (a->N <= 128) return launch_fwd<8>(a, s);
    if (a->N <= 224) return launch_fwd<14>(a, s);
    return launch_fwd<18>(a, s);
}

}  // namespace

extern "C" int gv_attention_fwd_varlen(const gv_attention_fwd_varlen_args* v, void* stream) {
    GV_REQUIRE(v && v->qkv && v->o, GV_E_NULL, "gv_attention_fwd_varlen: null pointer");
    GV_REQUIRE(v->n_seg >= 1 && v->n_seg <= GV_ATTN_MAX_SEG && v->H > 0, GV_E_SHAPE, "gv_attention_fwd_varlen: 1..%d segments, H > 0", GV_ATTN_MAX_SEG);
    GV_REQUIRE(gv_aligned(v->qkv, 16) && gv_aligned(v->o, 16), GV_E_ALIGN, "gv_attention_fwd_varlen: qkv/o must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    gv_attention_fwd_args seg[GV_ATTN_MAX_SEG];
    long row = 0;
    for (int i = 0; i < v->n_seg; ++i) {
        GV_REQUIRE(v->n_img[i] > 0 && v->N[i] > 0 && v->N[i] <= 288 && v->lse[i], GV_E_SHAPE, "gv_attention_fwd_varlen: segment %d: need n_img > 0, 0 < N <= 288, lse", i);
        seg[i].qkv = (const char*)v->qkv + row * 3 * v->H * 64 * 2;
        seg[i].o = (char*)v->o + row * v->H * 64 * 2;
        seg[i].lse = v->lse[i]; seg[i].n_img = v->n_img[i]; seg[i].N = v->N[i]; seg[i].H = v->H; seg[i].scale = v->scale;
        seg[i].q_limit = v->q_limit;
        row += (long)v->n_img[i] * v->N[i];
    }
    // one launch for a long + a short segment (either order); any other mix runs one launch per segment
    if (v->n_seg == 2) {
        const int il = (seg[0].N > 128 && seg[0].N <= 224) ? 0 : ((seg[1].N > 128 && seg[1].N <= 224) ? 1 : -1);
        const int is = il < 0 ? -1 : 1 - il;
        if (il >= 0 && seg[is].N > 32 && seg[is].N <= 64) {
            static GvLdsOptIn opt_in;
            if (int rc = gv_lds_opt_in(opt_in, (const void*)attn_fwd_varlen_kernel, VL_LDS, "gv_attention_fwd_varlen")) return rc;
            const int np_long = seg[il].n_img * v->H, np_short = seg[is].n_img * v->H;
            const int nblk_long = (np_long + FwdCfg<14>::PAIRS - 1) / FwdCfg<14>::PAIRS;
            const int nblk_short = (np_short + 2 * FwdCfg<4>::PAIRS - 1) / (2 * FwdCfg<4>::PAIRS);
            hipLaunchKernelGGL(attn_fwd_varlen_kernel, dim3(nblk_long + nblk_short), dim3(512), VL_LDS, s, seg[il], np_long, nblk_long, seg[is], np_short);
            GV_LAUNCH_CHECK("gv_attention_fwd_varlen");
            return GV_OK;
        }
    }
    for (int i = 0; i < v->n_seg; ++i)
        if (int rc = launch_fwd_any(&seg[i], s)) return rc;
    return GV_OK;
}

extern "C" int gv_attention_fwd(const gv_attention_fwd_args* a, void* stream) {
    GV_REQUIRE(a && a->qkv && a->o && a->lse, GV_E_NULL, "gv_attention_fwd: null pointer");
    GV_REQUIRE(a->n_img > 0 && a->H > 0 && a->N > 0 && a->N <= 288, GV_E_SHAPE, "gv_attention_fwd: need 0 < N <= 288 (got %d)", a->N);
    GV_REQUIRE(gv_aligned(a->qkv, 16) && gv_aligned(a->o, 16), GV_E_ALIGN, "gv_attention_fwd: qkv/o must be 16-byte aligned");
    return launch_fwd_any(a, (hipStream_t)stream);
}

extern "C" int gv_attention_bwd_varlen(const gv_attention_bwd_varlen_args* v, void* stream) {
    GV_REQUIRE(v && v->qkv && v->o && v->d_o && v->dqkv, GV_E_NULL, "gv_attention_bwd_varlen: null pointer");
    GV_REQUIRE(v->n_seg >= 1 && v->n_seg <= GV_ATTN_MAX_SEG && v->H > 0, GV_E_SHAPE, "gv_attention_bwd_varlen: 1..%d segments, H > 0", GV_ATTN_MAX_SEG);
    long row = 0;
    for (int i = 0; i < v->n_seg; ++i) {
        GV_REQUIRE(v->n_img[i] > 0 && v->N[i] > 0 && v->N[i] <= 288 && v->lse[i], GV_E_SHAPE, "gv_attention_bwd_varlen: segment %d: need n_img > 0, 0 < N <= 288, lse", i);
        gv_attention_bwd_args seg;
        seg.qkv = (const char*)v->qkv + row * 3 * v->H * 64 * 2;
        seg.o = (const char*)v->o + row * v->H * 64 * 2;
        seg.d_o = (const char*)v->d_o + row * v->H * 64 * 2;
        seg.dqkv = (char*)v->dqkv + row * 3 * v->H * 64 * 2;
        seg.lse = v->lse[i]; seg.n_img = v->n_img[i]; seg.N = v->N[i]; seg.H = v->H; seg.scale = v->scale;
        seg.q_limit = v->q_limit;
        if (int rc = gv_attention_bwd(&seg, stream)) return rc;
        row += (long)v->n_img[i] * v->N[i];
    }
    return GV_OK;
}

extern "C" int gv_attention_bwd(const gv_attention_bwd_args* a, void* stream) {
    GV_REQUIRE(a && a->qkv && a->o && a->d_o && a->lse && a->dqkv, GV_E_NULL, "gv_attention_bwd: null pointer");
    GV_REQUIRE(a->n_img > 0 && a->H > 0 && a->N > 0 && a->N <= 288, GV_E_SHAPE, "gv_attention_bwd: need 0 < N <= 288 (got %d)", a->N);
    GV_REQUIRE(gv_aligned(a->qkv, 16) && gv_aligned(a->o, 16) && gv_aligned(a->d_o, 16) && gv_aligned(a->dqkv, 16), GV_E_ALIGN,
               "gv_attention_bwd: buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // keys per wave: 16 for the short crops (N = 37: 39.2 -> 31.4 us per launch of 512 images x 6 heads), 32 from N = 65 on (N = 197:
    // 72.6 us against 73.8 with 16 -- twice the waves buy nothing there: per (image, head) pair the kernel is the SUM of ~8 us of
    // exp / dS arithmetic, ~4 us of MFMA and ~8 us (32 keys) or ~12 us (16 keys) of LDS reads, phase-aligned by its barriers)
    if (a->N <= 32) return launch_bwd<2, 2>(a, s);
    if (a->N <= 64) return launch_bwd<4, 1>(a, s);
    if (a->N <= 128) return launch_bwd<8, 2>(a, s);
    if (a->N <= 224) return launch_bwd<14, 2>(a, s);
    return launch_bwd<18, 2>(a, s);
}
