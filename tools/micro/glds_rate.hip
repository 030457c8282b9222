// Micro-benchmark: how fast can ONE workgroup per CU pull bytes into LDS with global_load_lds_dwordx4 (1 KiB per wave
// instruction), nothing else in the loop?  Region sizes: L2-resident and shared by all CUs / private per CU in L2-MALL / HBM.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/glds_rate.hip -o /tmp/glds_rate && /tmp/glds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
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
        glds16(base + (pos % region_bytes) + lane_off, lds0 + (i & 15) * 1024);
        pos += 8 * 8192;
        if (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (INFLIGHT == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
    const long total = 2L << 30;
    char* buf; hipMalloc(&buf, total + (1 << 20)); hipMemset(buf, 1, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, long region, long stride, int row_bytes) {
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
    return 0;
}
