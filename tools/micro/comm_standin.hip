// Stand-in for RCCL's channel workgroups on a ONE-GPU box (tools/comm_standin.py): C persistent workgroups of 256 threads copy
// `bytes` from src to dst with 16-B accesses -- the HBM traffic and the CU occupancy an all-reduce of that range brings to a rank,
// without a peer.  Few registers, no LDS: like a channel, it needs a CU slot of its own only because this build's heavy kernels
// hold a CU's whole register file and LDS.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/comm_standin.hip -o tools/lab_build/libcomm_standin.so
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void standin_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n_vec, int passes) {
    const long stride = (long)gridDim.x * 256;
    for (int p = 0; p < passes; ++p)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += 4 * stride) {
            f32x4 r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) if (i + j * stride < n_vec) r[j] = __builtin_nontemporal_load(src + i + j * stride);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (i + j * stride < n_vec) __builtin_nontemporal_store(r[j], dst + i + j * stride);
        }
}

extern "C" int comm_standin_launch(const void* src, void* dst, long bytes, int workgroups, int passes, void* stream) {
    if (workgroups <= 0 || bytes <= 0) return 0;
    hipLaunchKernelGGL(standin_copy, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, bytes / 16, passes);
    return (int)hipGetLastError();
}
