// Two-context form of gv_linear's wide bf16 products (N = ncb x 384 output columns, K % 128 == 0: attn.qkv, mlp.fc1 (+ GELU,
// + saved pre-activation), the GELU' . dX product of mlp.fc2, the dX product of attn.proj -- vit.pyc@L98-104, L119-131 and
// their autograd).  Same arithmetic, summation order and results as panel.hip's MODE_WIDE (bit-identical: one accumulator
// chain per output over k in increasing order on v_mfma_f32_16x16x32), different occupancy:
//
//   panel.hip MODE_WIDE: ONE 8-wave workgroup per CU owns a 176-row panel: an MFMA-bound k-loop, then a vector-ALU-bound
//   epilogue (GELU polynomial, conversions, the LDS transit), one after the other, 252 workgroups in lockstep -- the matrix
//   pipe idles through every epilogue and the vector ALU through every k-loop (17.8 k + 17.7 k cycles
//   per panel of fc1).
//
//   here: TWO 4-wave workgroups ("contexts") per CU, one wave per SIMD each (256 registers, 64 KB of LDS), each walking
//   several 16 FM-row panels (FM <= 6: 144 accumulator registers) of ONE 384-column block, persistently.  The context that
//   arrives second on a CU starts a fraction of a panel late, so that one context's epilogue (vector ALU, stores) runs under the
//   other's k-loop (MFMA, LDS-DMA): the hardware arbitrates between the two wave sets per SIMD, nothing in the code pairs
//   them.  A wave owns 96 consecutive output columns of all the panel's rows (6 x FM fragments), so an epilogue row piece
//   is 192 contiguous bytes and a k-step reads FM + 6 fragments for 6 FM MFMAs.
//
// LAB FILE, not part of the product library (tools/lab/build_wide2.sh builds it into a lab library).  Measured in round 4 and
// dropped: slower than panel.hip's MODE_WIDE on every shape (fc1 117 vs 97 us, qkv 66 vs 50), with or without the stagger --
// the s_memtime stamps (tools/wide2_stamps.py, profiles/r04_wide2_two_context_stamps.txt) show why: see LAB_NOTES.md.
//
// k-loop: BK = 32, two ring stages of (128 A rows + 3 x 128 weight columns) x 32 = 32 KB, LDS-DMA one step ahead behind
// vmcnt(0) + one raw s_barrier per step (the co-resident context covers the wait; the single-context loop is NOT meant to be
// MFMA-bound on its own).  The next panel's first stage is requested before the epilogue.
#include "gemm_core.h"       // (-I gipmed-project-self-supervised-vit_amd/csrc)
#include "timing.h"
#include <type_traits>
#include <stdlib.h>

namespace {

using namespace gvgemm;

constexpr int PN = 384;
constexpr int NW2 = 4, BK2 = 32;
enum { W2_NONE = 0, W2_BIAS = 1, W2_BIAS_GELU = 2, W2_BIAS_GELU_SAVE = 3, W2_DGELU = 4 };

// contexts resident per compute unit, keyed by (XCC id, SE / SH / CU id): the first workgroup to arrive on a CU reads 0 and
// starts at once, the second reads 1 and starts `stagger` sleeps late; every workgroup gives its count back on exit, so the
// table is all zeros between launches.  Placement only decides a DELAY: results never depend on it.
__device__ unsigned g_occupancy[2048];

struct Wide2P {
    const bf16* A; const bf16* W; int M, K; long lda, ldw;
    const float* bias; bf16* out; long ldo; const bf16* aux_in; bf16* aux_out; long ld_aux;
    int ncb, n_total;
    int npan, n_panels;       // panels per workgroup walk, panels in all
    int stagger;              // late context: s_sleep(127) repetitions (~8 k cycles each) before its first k-loop
#ifdef GV_WIDE2_LAB             // lab: ablation bits (1 no global stores, 2 no GELU arithmetic, 4 no MFMAs) + s_memtime stamps
    int lab; unsigned long long* dbg;      // dbg[workgroup][8 panels][4]: k-loop start, k-loop end, epilogue end, (cu key << 8 | role)
#endif
};
#ifdef GV_WIDE2_LAB
__device__ __forceinline__ unsigned long long w2_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define W2_STAMP(k) do { if (tid == 0 && it < 8) p.dbg[((long)blockIdx.x * 8 + it) * 4 + (k)] = w2_stamp(); } while (0)
#define W2_LAB(bit) ((p.lab & (bit)) != 0)
#else
#define W2_STAMP(k)
#define W2_LAB(bit) false
#endif

template <int FM, bool TB, int EP>
__global__ __launch_bounds__(NW2 * 64, 2) void wide2_kernel(const Wide2P p) {
    constexpr int NW = NW2, BK = BK2, BM = FM * 16;
    constexpr int A_ROWS = 128;                         // staged A rows: two 16-row pieces per wave (>= BM; surplus rows are never used)
    constexpr int A_BYTES = A_ROWS * BK * 2;
    constexpr int W_BLOCK = 128 * BK * 2;
    constexpr int STAGE = A_BYTES + 3 * W_BLOCK;        // 32 KB
    constexpr int RS = 100;                             // per-wave epilogue image [16 rows][96 columns] f32, row stride 100
    static_assert(FM * 16 <= A_ROWS && NW * 16 * RS * 4 <= STAGE, "geometry");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    GV_LDS char* smem = (GV_LDS char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M;

    // workgroup -> (column block, first panel): ids go round-robin over the 8 XCDs (id & 7 labels the XCD, id >> 3 the slot on
    // it); the ncb workgroups that walk the SAME panels sit in consecutive slots of one XCD, so a panel leaves HBM once
    const int ncb = p.ncb, slot = blockIdx.x >> 3;
    const int lg = __builtin_amdgcn_readfirstlane(slot / ncb);
    const int grp = lg * 8 + (blockIdx.x & 7);
    const int cb = slot - lg * ncb;
    const int first = grp * p.npan;
    if (first >= p.n_panels) return;                    // (whole workgroup, before any barrier)
    const int nit = p.n_panels - first < p.npan ? p.n_panels - first : p.npan;

    // second context on this CU?  (HW_REG_HW_ID bits 8..15: CU, SH, SE id; HW_REG_XCC_ID bits 0..3)
    const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const unsigned cu_key = ((xcc_id & 7u) << 8) | ((hw_id >> 8) & 0xFFu);
    {
        GV_LDS unsigned* role_s = (GV_LDS unsigned*)smem;
        if (tid == 0) *role_s = atomicAdd(&g_occupancy[cu_key], 1u);
        __syncthreads();
        const unsigned role = *role_s;
        __syncthreads();                                // (the word is ring space from here on)
        if (role != 0) for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
#ifdef GV_WIDE2_LAB
        if (tid == 0) p.dbg[((long)blockIdx.x * 8 + 0) * 4 + 3] = ((unsigned long long)cu_key << 8) | role;
#endif
    }

    // panel i starts at min(i BM, M - BM): every panel holds BM valid rows (the rows the last two panels share are computed
    // twice, bit-identically: the epilogues are pure functions of the row)
    auto panel_origin = [&](int idx) { const int o = idx * BM; return o < M - BM ? o : M - BM; };

    TileSrc<false, A_ROWS, BK, NW> srcA;
    TileSrc<TB, 128, BK, NW> srcW[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) srcW[j].setup(p.W, p.ldw, cb * PN + 128 * j, p.n_total, wave, lane);
    const int nt = p.K / BK;                            // even (K % 128 == 0)
    auto issue = [&](int t) {
        GV_LDS char* st = smem + (t & 1) * STAGE;
        const int k0 = t * BK;
        srcA.issue(p.lda, k0, p.K, st, wave);
#pragma unroll
        for (int j = 0; j < 3; ++j) srcW[j].issue(p.ldw, k0, p.K, st + A_BYTES + j * W_BLOCK, wave);
    };
    int m0 = panel_origin(first);
    srcA.setup(p.A, p.lda, m0, M, wave, lane);
    issue(0);

    // this wave's six column fragments f = 6 wave + j: 128-column weight block f >> 3, local fragment f & 7
    int wblk[6], wfr[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { const int f = 6 * wave + j; wblk[j] = (f >> 3) * W_BLOCK; wfr[j] = f & 7; }

    for (int it = 0; it < nit; ++it) {
        // acc[i][j][r]: row m0 + 16 i + (lane & 15), column 384 cb + 96 wave + 16 j + 4 (lane >> 4) + r
        f32x4 acc[FM][6];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        W2_STAMP(0);
        for (int t = 0; t < nt; ++t) {
            wait_vmcnt<0>();                            // my pieces of stage t have landed (and, at t = 0, the last epilogue's stores)
            __builtin_amdgcn_s_barrier();               // everybody's have; every wave is past its reads of stage t - 1 / its image
            if (t + 1 < nt) issue(t + 1);
            GV_LDS char* cur = smem + (t & 1) * STAGE;
            bf16x8 fw[6], fa[FM];
#pragma unroll
            for (int j = 0; j < 6; ++j) fw[j] = read_frag<TB, 128, BK>(cur + A_BYTES + wblk[j], wfr[j], 0, lane);
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = read_frag<false, A_ROWS, BK>(cur, i, 0, lane);
            if (W2_LAB(4)) {
#pragma unroll
                for (int j = 0; j < 6; ++j) asm volatile("" ::"v"(fw[j]));
#pragma unroll
                for (int i = 0; i < FM; ++i) asm volatile("" ::"v"(fa[i]));
            } else
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[i][j] = GV_MFMA_16x16x32(fw[j], fa[i], acc[i][j]);
        }
        W2_STAMP(1);
        // ---- the k-loop ended in stage 1 (nt even).  Behind one barrier the whole ring is free: the next panel's stage 0 goes out
        // now and flies through the epilogue, whose per-wave images live in stage 1 (the next k-loop's first barrier keeps stage 1
        // untouched until every wave has left its image)
        const int m0_cur = m0;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        if (it + 1 < nit) {
            m0 = panel_origin(first + it + 1);
            srcA.setup(p.A, p.lda, m0, M, wave, lane);
            issue(0);
        }
        // ---- epilogue, per wave: one 16-row fragment at a time through a private image, then units of 4 rows x 32 columns
        // (lane: row 4 q + (lane >> 4), columns 32 b + 2 (lane & 15) + {0, 1}): every global access of a unit is four 64-B row pieces,
        // three units side by side make the wave's 192-B row segment
        constexpr bool WB = EP >= W2_BIAS && EP <= W2_BIAS_GELU_SAVE;
        int le = lane;
        asm volatile("" : "+v"(le));                    // (lane-derived addresses are recomputed per panel: not held across the k-loop)
        const int urow = le >> 4, ucol = (le & 15) * 2, li16e = le & 15, gqe = le >> 4;
        GV_LDS float* wimg = (GV_LDS float*)(smem + STAGE) + wave * (16 * RS);
        f32x4 bw[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            bw[j] = (WB && p.bias) ? *(const f32x4*)(p.bias + cb * PN + 96 * wave + 16 * j + gqe * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int mrow = m0_cur + 16 * i + urow;    // + 4 q: this lane's row of unit (b, q); < M (every panel holds BM rows)
            const int ccol = cb * PN + 96 * wave + ucol;    // + 32 b
            bf16* const o_bf = p.out + (long)mrow * p.ldo + ccol;
            [[maybe_unused]] bf16* const o_aux = EP == W2_BIAS_GELU_SAVE ? p.aux_out + (long)mrow * p.ld_aux + ccol : nullptr;
            [[maybe_unused]] const bf16* const i_aux = EP == W2_DGELU ? p.aux_in + (long)mrow * p.ld_aux + ccol : nullptr;
            const long s_ob = 4 * p.ldo, s_aux = 4 * p.ld_aux;
            bf16x2 ax[EP == W2_DGELU ? 12 : 1];
            if constexpr (EP == W2_DGELU) {             // the fragment's pre-activation rows, all requested up front
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) ax[b * 4 + q] = *(const bf16x2*)(i_aux + q * s_aux + 32 * b);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                f32x4 v = acc[i][j];
                if constexpr (WB) v += bw[j];
                *(GV_LDS f32x4*)(wimg + li16e * RS + 16 * j + gqe * 4) = v;
            }
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 t2 = *(GV_LDS f32x2*)(wimg + (4 * q + urow) * RS + 32 * b + ucol);       // (a wave's LDS accesses execute in order)
                    float v0 = t2[0], v1 = t2[1];
                    if constexpr (EP == W2_BIAS_GELU_SAVE) {
                        // nontemporal: the saved pre-activation is not read again before the backward pass (panel.hip)
                        if (!W2_LAB(1)) __builtin_nontemporal_store(bf16x2{(bf16)v0, (bf16)v1}, (bf16x2*)(o_aux + q * s_aux + 32 * b));
                    }
                    if (!W2_LAB(2)) {
                    if constexpr (EP == W2_BIAS_GELU || EP == W2_BIAS_GELU_SAVE) { v0 = gelu_f(v0); v1 = gelu_f(v1); }
                    if constexpr (EP == W2_DGELU) { v0 *= dgelu_f((float)ax[b * 4 + q][0]); v1 *= dgelu_f((float)ax[b * 4 + q][1]); }
                    }
                    if (W2_LAB(1)) asm volatile("" ::"v"(v0), "v"(v1));
                    else *(bf16x2*)(o_bf + q * s_ob + 32 * b) = bf16x2{(bf16)v0, (bf16)v1};
                }
        }
        W2_STAMP(2);
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (tid == 0) atomicSub(&g_occupancy[cu_key], 1u);
}

int slots2() { return 2 * gv_cu_budget(); }      // workgroup slots of a launch: two contexts per CU of the budget
int stagger2() {
    static const int s = [] { const char* e = getenv("GIPVIT_WIDE2_STAGGER"); return e ? atoi(e) : 1; }();
    return s;
}

#ifdef GV_WIDE2_LAB
unsigned long long* g_w2_dbg = nullptr;
constexpr size_t W2_DBG_BYTES = 1024 * 8 * 4 * 8;
#endif
struct Geo { int fm, npan, n_panels, grid; };
// panel height and walk length: the FM in {6, 5, 4} whose walk covers the fewest rows per workgroup (= time), ties to the taller
Geo pick_geo(int M, int ncb) {
    const int gmax = (slots2() / ncb) / 8 * 8;
    Geo best{0, 0, 0, 0};
    long best_rows = 1L << 60;
    for (int fm = 6; fm >= 4; --fm) {
        const int P = (M + 16 * fm - 1) / (16 * fm);
        const int npan = (P + gmax - 1) / gmax;
        const long rows = (long)npan * 16 * fm;
        if (rows < best_rows) {
            const int groups = (P + npan - 1) / npan;
            best = Geo{fm, npan, P, 8 * ncb * ((groups + 7) / 8)};
            best_rows = rows;
        }
    }
    return best;
}

template <int FM, bool TB, int EP>
int launch2(const Wide2P& p, int grid, hipStream_t s) {
    auto kern = wide2_kernel<FM, TB, EP>;
    constexpr int LDS_BYTES = 2 * (128 * BK2 * 2 + 3 * 128 * BK2 * 2);
    static GvLdsOptIn opt_in;
    if (int rc = gv_lds_opt_in(opt_in, (const void*)kern, LDS_BYTES, "gv_linear(wide2)")) return rc;
    struct Name { char s[64]; Name() { snprintf(s, sizeof(s), "wide2_kernel<%d, %s, %d>", FM, TB ? "true" : "false", EP); } };
    static const Name kn;
    const double row_bytes = 2.0 * p.K + 2.0 * p.n_total * ((EP == W2_BIAS_GELU_SAVE || EP == W2_DGELU) ? 2 : 1);
    const int th = gvtime::enabled() ? gvtime::begin(kn.s, 2.0 * p.M * p.n_total * p.K, p.M * row_bytes + 2.0 * p.n_total * p.K, s) : -1;
#ifdef GV_WIDE2_LAB
    if (!g_w2_dbg) (void)hipMalloc(&g_w2_dbg, W2_DBG_BYTES);
    (void)hipMemsetAsync(g_w2_dbg, 0, W2_DBG_BYTES, s);
    const_cast<Wide2P&>(p).dbg = g_w2_dbg;
    { const char* e = getenv("GIPVIT_WIDE2_LAB"); const_cast<Wide2P&>(p).lab = e ? atoi(e) : 0; }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW2 * 64), LDS_BYTES, s, p);
    gvtime::end(th, s);
    GV_LAUNCH_CHECK("gv_linear(wide2)");
    return GV_OK;
}

template <bool TB, int EP>
int dispatch2(Wide2P& p, hipStream_t s) {
    const Geo g = pick_geo(p.M, p.ncb);
    p.npan = g.npan; p.n_panels = g.n_panels; p.stagger = stagger2();
    switch (g.fm) {
        case 4: return launch2<4, TB, EP>(p, g.grid, s);
        case 5: return launch2<5, TB, EP>(p, g.grid, s);
        default: return launch2<6, TB, EP>(p, g.grid, s);
    }
}

}  // namespace

// -1: not one of this kernel's calls (the caller goes on to panel.hip's MODE_WIDE / the 128x128-tile kernel)
int gv_panel_wide2(const gv_linear_args* a, hipStream_t s);
int gv_panel_wide2(const gv_linear_args* a, hipStream_t s) {
    static const bool on = [] { const char* e = getenv("GIPVIT_WIDE2"); return e && atoi(e) != 0; }();
    if (!on) return -1;
    if (a->trans_a || a->c_is_f32 || a->N % PN != 0 || a->K % 128 != 0 || a->M < 2048 || a->ldc % 2 != 0 || a->ld_aux % 2 != 0) return -1;
    if (a->alpha != 0.f && a->alpha != 1.f) return -1;
    const int e = a->epilogue;
    const bool fwd = !a->trans_b && (e == GV_EPI_BIAS || e == (GV_EPI_BIAS | GV_EPI_GELU) || e == (GV_EPI_BIAS | GV_EPI_GELU | GV_EPI_SAVE_PRE));
    const bool bwd = a->trans_b && (e == 0 || e == GV_EPI_DGELU);
    if (!fwd && !bwd) return -1;
    if (s == (hipStream_t)(intptr_t)-1) return GV_OK;       // gv_workspace_bytes: runs here, takes no scratch
    Wide2P p{};
    p.A = (const bf16*)a->A; p.W = (const bf16*)a->B; p.M = a->M; p.K = a->K; p.lda = a->lda; p.ldw = a->ldb;
    p.bias = a->bias; p.out = (bf16*)a->C; p.ldo = a->ldc; p.aux_in = (const bf16*)a->aux_in; p.aux_out = (bf16*)a->aux_out; p.ld_aux = a->ld_aux;
    p.ncb = a->N / PN; p.n_total = a->N;
    if (!a->trans_b) {
        if (e == GV_EPI_BIAS) return dispatch2<false, W2_BIAS>(p, s);
        if (e == (GV_EPI_BIAS | GV_EPI_GELU)) return dispatch2<false, W2_BIAS_GELU>(p, s);
        return dispatch2<false, W2_BIAS_GELU_SAVE>(p, s);
    }
    if (e == 0) return dispatch2<true, W2_NONE>(p, s);
    return dispatch2<true, W2_DGELU>(p, s);
}

#ifdef GV_WIDE2_LAB      // tuning-lab build only (tools/wide2_stamps.py)
extern "C" int gv_wide2_dbg_read(unsigned long long* host) { if (!g_w2_dbg) return -1; return (int)hipMemcpy(host, g_w2_dbg, W2_DBG_BYTES, hipMemcpyDeviceToHost); }
#endif
