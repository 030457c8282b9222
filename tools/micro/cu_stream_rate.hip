// Micro-benchmark: what does ONE CU stream from HBM when only G of the 256 CUs stream at all?  One 8-wave workgroup per CU
// (128 KB of LDS requested so that no two share a CU), each reading its own 64-MB region with 16-B loads, 8 per lane in flight
// (64 KB per workgroup), and with a read + write mix like a LayerNorm epilogue's.  Answers whether shifting workgroups or kernels
// against each other in time could make an HBM-bound epilogue faster.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cu_stream_rate.hip -o tools/lab_build/cu_stream_rate && tools/lab_build/cu_stream_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool WRITE>
__global__ __launch_bounds__(512) void stream(const f32x4* src, f32x4* dst, long per_wg_vec, int iters, float* sink) {
    extern __shared__ char smem[];
    const f32x4* p = src + (long)blockIdx.x * per_wg_vec + threadIdx.x;
    f32x4* q = dst + (long)blockIdx.x * per_wg_vec + threadIdx.x;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
        f32x4 r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __builtin_nontemporal_load(p + (long)(i * 8 + j) * 512);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (WRITE) __builtin_nontemporal_store(r[j] * 1.5f, q + (long)(i * 8 + j) * 512);
            else acc += r[j];
        }
    }
    if (!WRITE && acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e30f) sink[0] = acc[0];
    if (threadIdx.x == 100000) smem[0] = 1;
}

int main() {
    const long per_wg = 64L << 20, per_wg_vec = per_wg / 16;
    const int iters = (int)(per_wg_vec / (512 * 8));
    f32x4 *src, *dst; float* sink;
    if (hipMalloc(&src, per_wg * 256) != hipSuccess || hipMalloc(&dst, per_wg * 256) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(src, 1, per_wg * 256); (void)hipMemset(dst, 0, per_wg * 256);
    (void)hipFuncSetAttribute((const void*)stream<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)hipFuncSetAttribute((const void*)stream<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wr = 0; wr < 2; ++wr)
        for (int G : {8, 16, 32, 64, 128, 192, 256}) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (wr) hipLaunchKernelGGL(stream<true>, dim3(G), dim3(512), 131072, 0, src, dst, per_wg_vec, iters, sink);
                else hipLaunchKernelGGL(stream<false>, dim3(G), dim3(512), 131072, 0, src, dst, per_wg_vec, iters, sink);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double bytes = (double)per_wg * (wr ? 2 : 1);
            printf("%s  %3d CUs streaming: %6.1f KB/us per CU, %5.2f TB/s in all\n", wr ? "read+write" : "read      ", G, bytes / (best * 1e-3) / 1e9 * 1e3 / 1e0 / 1e3 * 1e0, bytes * G / (best * 1e-3) / 1e12);
        }
    return 0;
}
