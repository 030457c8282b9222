// (region sizes are powers of two: the wrap is a mask -- a 64-bit modulo per piece costs more VALU time than the load)
// Micro-benchmark: how fast can ONE workgroup per CU pull bytes into LDS with global_load_lds_dwordx4 (1 KiB per wave
// instruction), nothing else in the loop?  Region sizes: L2-resident and shared by all CUs / private per CU in L2-MALL / HBM.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/glds_rate.hip -o /tmp/glds_rate && /tmp/glds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <signal.h>
#define LDSP __attribute__((address_space(3)))

__device__ __forceinline__ void glds16(const void* src, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
}

template <int INFLIGHT>
__global__ __launch_bounds__(512, 2) void fill(const char* src, long region_bytes, long per_wg_stride, int iters, int row_bytes) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(LDSP char*)smem + wave * 16384;
    const char* base = src + (long)blockIdx.x * per_wg_stride;
    // piece p of this wave: 1 KiB; with row_bytes = 128 a piece is 8 rows of 128 B at a row pitch of 768 B (a [rows][384]-bf16 slice)
    const long lane_off = row_bytes >= 1024 ? lane * 16 : (long)(lane / (row_bytes / 16)) * 768 + (lane % (row_bytes / 16)) * 16;
    long pos = (long)wave * 8192;
    for (int i = 0; i < iters; ++i) {
        glds16(base + (pos & (region_bytes - 1)) + lane_off, lds0 + (i & 15) * 1024);
        pos += 8 * 8192;
        if (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (INFLIGHT == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The same stream through the VGPR path (global_load_dwordx4 into registers, nothing consumes them until the end), and a
// 50 / 50 mix of both instructions: does a CU's ingest cap belong to the LDS-DMA path or to the L2 -> CU return path?
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>    // 0: VGPR loads only, 1: alternate LDS-DMA / VGPR
__global__ __launch_bounds__(512, 2) void fill_vgpr(const char* src, long region_bytes, long per_wg_stride, int iters, int row_bytes, float* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(LDSP char*)smem + wave * 16384;
    const char* base = src + (long)blockIdx.x * per_wg_stride;
    const long lane_off = row_bytes >= 1024 ? lane * 16 : (long)(lane / (row_bytes / 16)) * 768 + (lane % (row_bytes / 16)) * 16;
    long pos = (long)wave * 8192;
    f32x4 r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const char* p = base + (pos & (region_bytes - 1)) + lane_off;
            if (MODE == 1 && (j & 1)) glds16(p, lds0 + ((i + j) & 15) * 1024);
            // "+v": the destination stays allocated between loads -- with "=v" the compiler may reuse it for the next
            // address while the load is still in flight, and the returning data then corrupts the pointer
            else asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(r[j]) : "v"(p) : "memory");
            pos += 8 * 8192;
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += r[j][0] + r[j][3];
    if (acc == 123.456f) sink[threadIdx.x] = acc;
}

// What a k-loop sees: ONE 8-wave workgroup per CU, every wave streaming 1-KB pieces in order with at most INFLIGHT of
// its own outstanding -- MIX = 0: all from HBM; MIX = 1: every third piece from HBM, two from an L2-resident region
// (the A : W byte ratio of the full-row kernels' stages); MIX = 2: all L2.  Rate against bytes in flight per CU.
template <int INFLIGHT> __device__ __forceinline__ void wait_inflight() {
    if constexpr (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (INFLIGHT == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}
template <int INFLIGHT, int MIX, int READS = 0>      // READS: ds_read_b128 wave instructions (1 KB each) per piece issued -- a k-loop reads 2.7 KB of fragments per staged KB
__global__ __launch_bounds__(512, 1) void fill_mix(const char* src, int iters) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(LDSP char*)smem + wave * 16384;
    const char* hbm = src + (1L << 20) + (long)blockIdx.x * (6L << 20);      // private 4 MB window per CU behind the shared 1 MB
    const char* l2 = src;
    long ph = (long)wave * 1024, pl = (long)wave * 1024;
    for (int i = 0; i < iters; ++i) {
        const bool from_hbm = MIX == 0 || MIX == 3 || ((MIX == 1 || MIX == 4) && i % 3 == 0);
        // MIX 3 / 4: the "HBM" window is 512 KB per CU (128 MB over the chip: past the 8 x 4 MB of L2, inside the 256-MB Infinity
        // Cache) and is walked many times -- Infinity-Cache hits after the first pass
        const long hwin = (MIX >= 3 ? (512L << 10) : (4L << 20)) - 1;
        const char* p = from_hbm ? hbm + (ph & hwin) : l2 + (pl & ((1L << 20) - 1));
        if (from_hbm) ph += 8 * 1024; else pl += 8 * 1024;
        glds16(p + lane * 16, lds0 + (i & 15) * 1024);
#pragma unroll
        for (int r = 0; r < READS; ++r) {
            f32x4 v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds0 + ((i + 8 + r) & 15) * 1024 + lane * 16) : "memory");
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(v)::"memory");
        }
        wait_inflight<INFLIGHT>();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// The A operand of a full-row kernel as the k-loop actually walks it: a [176 rows][1536] bf16 panel (540 KB, contiguous in
// memory) read as 24 sweeps of 128 B per row (row pitch 3072 B) -- against the same bytes read front to back.
template <int STRIDED>
__global__ __launch_bounds__(512, 1) void fill_panel(const char* src, int panels) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(LDSP char*)smem + wave * 16384;
    const char* base = src + (1L << 20) + (long)blockIdx.x * (6L << 20);
    int n = 0;
    for (int p = 0; p < panels; ++p) {
        const char* pb = base + (long)p * (176 * 3072);
        for (int kk = 0; kk < 24; ++kk)
            for (int piece = wave; piece < 22; piece += 8) {
                const char* a = STRIDED ? pb + (long)(piece * 8 + (lane >> 3)) * 3072 + kk * 128 + (lane & 7) * 16
                                        : pb + ((long)kk * 22 + piece) * 1024 + lane * 16;
                glds16(a, lds0 + (n++ & 15) * 1024);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main(int argc, char** argv) {
    signal(SIGPIPE, SIG_IGN);
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    int idx = 0;
    const long total = 2L << 30;
    char* buf; hipMalloc(&buf, total + (1 << 20)); hipMemset(buf, 1, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, long region, long stride, int row_bytes) {
        if (only >= 0 && only != idx++) return;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        const int iters = 4096;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, buf, region, stride, iters, row_bytes);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 256.0 * 8 * iters * 1024;
        printf("%-58s %7.1f us  %6.2f TB/s  %5.1f KB/us/CU  (%4.1f B/clk/CU at 2.1 GHz)\n", name, ms * 1e3, bytes / ms / 1e9, bytes / 256 / ms / 1e3 / 1e3 * 1e3 / 1e3,
               bytes / 256 / (ms * 1e-3 * 2.1e9));
    };
    run("shared 1 MB region (L2 hits), 1-KB contiguous pieces, 8 in flight", fill<8>, 1 << 20, 0, 1024);
    run("shared 1 MB region (L2 hits), 1-KB contiguous pieces, 16 in flight", fill<16>, 1 << 20, 0, 1024);
    run("shared 1 MB region (L2 hits), 1-KB contiguous pieces, 32 in flight", fill<32>, 1 << 20, 0, 1024);
    run("shared 1 MB region, 128-B rows at 768-B pitch, 16 in flight", fill<16>, 1 << 20, 0, 128);
    run("private 512 KB per CU (L2 / MALL), contiguous, 16 in flight", fill<16>, 512 << 10, 512 << 10, 1024);
    run("private 8 MB per CU = 2 GB (HBM), contiguous, 16 in flight", fill<16>, 8 << 20, 8 << 20, 1024);
    run("private 8 MB per CU = 2 GB (HBM), contiguous, 32 in flight", fill<32>, 8 << 20, 8 << 20, 1024);
    auto runm = [&](const char* name, auto kern, int inflight, int mix) {
        if (only >= 0 && only != idx++) return;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        const int iters = mix == 0 ? 1024 : (mix >= 3 ? 6144 : 3072);       // (a pure-HBM wave walks its 4 MB window at most twice: no MALL re-use)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, buf, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 256.0 * 8 * iters * 1024;
        printf("%-34s %2d KB in flight per wave, %3d KB per CU: %7.1f us  %6.2f TB/s  %5.1f KB/us/CU  -> %4.2f us per piece in flight\n", name, inflight + 1, 8 * (inflight + 1),
               ms * 1e3, bytes / ms / 1e9, bytes / 256 / (ms * 1e3) / 1e3, 8.0 * (inflight + 1) * 1024 / (bytes / 256 / (ms * 1e3)));
        fflush(stdout);
    };
#define RUNM(I) runm("one WG per CU, all HBM", fill_mix<I, 0>, I, 0); runm("one WG per CU, 1/3 HBM + 2/3 L2", fill_mix<I, 1>, I, 1); runm("one WG per CU, all L2", fill_mix<I, 2>, I, 2);
    RUNM(1) RUNM(2) RUNM(4) RUNM(8) RUNM(12) RUNM(16)
    runm("one WG per CU, all Infinity Cache", fill_mix<1, 3>, 1, 3); runm("one WG per CU, all Infinity Cache", fill_mix<2, 3>, 2, 3);
    runm("one WG per CU, all Infinity Cache", fill_mix<4, 3>, 4, 3); runm("one WG per CU, all Infinity Cache", fill_mix<8, 3>, 8, 3);
    runm("one WG per CU, all Infinity Cache", fill_mix<16, 3>, 16, 3);
    runm("one WG per CU, 1/3 Inf. Cache + 2/3 L2", fill_mix<4, 4>, 4, 4); runm("one WG per CU, 1/3 Inf. Cache + 2/3 L2", fill_mix<8, 4>, 8, 4);
    runm("... 1/3 Inf. Cache + 2/3 L2, 3 ds_read_b128 per piece", fill_mix<8, 4, 3>, 8, 4);
    runm("... 1/3 HBM + 2/3 L2, 3 ds_read_b128 per piece", fill_mix<8, 1, 3>, 8, 1);
    runm("... 1/3 HBM + 2/3 L2, 6 ds_read_b128 per piece", fill_mix<8, 1, 6>, 8, 1);
    runm("... all L2, 3 ds_read_b128 per piece", fill_mix<8, 2, 3>, 8, 2);
    auto runp = [&](const char* name, auto kern) {
        if (only >= 0 && only != idx++) return;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        const int panels = 10;                           // 5.4 MB of each CU's 6 MB window, read once
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, buf, panels);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 256.0 * panels * 176 * 3072;
        printf("%-70s %7.1f us  %6.2f TB/s  %5.1f KB/us/CU\n", name, ms * 1e3, bytes / ms / 1e9, bytes / 256 / (ms * 1e3) / 1e3);
        fflush(stdout);
    };
    runp("A panels from HBM, front to back (1-KB contiguous pieces)", fill_panel<0>);
    runp("A panels from HBM, as the k-loop walks them (128 B per row, pitch 3072)", fill_panel<1>);
    float* sink; hipMalloc(&sink, 4096);
    auto runv = [&](const char* name, auto kern, long region, long stride, int row_bytes) {
        if (only >= 0 && only != idx++) return;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        const int iters = 4096;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, buf, region, stride, iters, row_bytes, sink);
            hipError_t le = hipGetLastError();
            hipEventRecord(e1);
            hipError_t se = hipEventSynchronize(e1);
            if (le != hipSuccess || se != hipSuccess) { printf("%s: launch %s, sync %s\n", name, hipGetErrorString(le), hipGetErrorString(se)); fflush(stdout); }
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 256.0 * 8 * iters * 1024;
        printf("%-58s %7.1f us  %6.2f TB/s  %5.1f KB/us/CU  (%4.1f B/clk/CU at 2.1 GHz)\n", name, ms * 1e3, bytes / ms / 1e9, bytes / 256 / ms / 1e3 / 1e3 * 1e3 / 1e3,
               bytes / 256 / (ms * 1e-3 * 2.1e9));
        fflush(stdout);
    };
    runv("VGPR loads, shared 1 MB region (L2 hits), contiguous", fill_vgpr<0>, 1 << 20, 0, 1024);
    runv("VGPR loads, shared 1 MB region, 128-B rows at 768-B pitch", fill_vgpr<0>, 1 << 20, 0, 128);
    runv("VGPR loads, private 8 MB per CU = 2 GB (HBM)", fill_vgpr<0>, 8 << 20, 8 << 20, 1024);
    runv("LDS-DMA / VGPR alternating, shared 1 MB region (L2 hits)", fill_vgpr<1>, 1 << 20, 0, 1024);
    runv("LDS-DMA / VGPR alternating, private 8 MB per CU (HBM)", fill_vgpr<1>, 8 << 20, 8 << 20, 1024);
    return 0;
}
