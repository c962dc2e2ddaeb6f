// Micro-benchmarks of launch / dependency floors (diagnostics only; not part of the talker path).
#include "common.cuh"

__global__ void dbg_empty_kernel() {}
__global__ void dbg_touch_kernel(float* p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    p[i] = p[i] + 1.0f;
}
// `depth` dependent loads per thread (pointer chase through idx[])
__global__ void dbg_chase_kernel(const int* idx, int* out, int depth) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    for (int d = 0; d < depth; ++d) j = idx[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = j;
}

__global__ __launch_bounds__(512) void dbg_lds_kernel(float* p) {
    extern __shared__ float sm[];
    if (p && threadIdx.x == 9999) p[0] = sm[0];
}
// many live VGPRs, no memory traffic
template <int NV>
__global__ __launch_bounds__(512) void dbg_vgpr_kernel(float* p, int n) {
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0001f + v[(i + 1) % NV];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    if (s == 12345.678f) p[0] = s;
}

extern "C" int omni_debug_launch(int mode, int blocks, int threads, void* p0, void* p1, int arg, int reps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    for (int r = 0; r < reps; ++r) {
        if (mode == 0) hipLaunchKernelGGL(dbg_empty_kernel, dim3(blocks), dim3(threads), 0, st);
        else if (mode == 1) hipLaunchKernelGGL(dbg_touch_kernel, dim3(blocks), dim3(threads), 0, st, (float*)p0);
        else if (mode == 3) hipLaunchKernelGGL(dbg_lds_kernel, dim3(blocks), dim3(threads), (size_t)arg, st, (float*)nullptr);
        else if (mode == 4) hipLaunchKernelGGL(dbg_vgpr_kernel<32>, dim3(blocks), dim3(threads), 0, st, (float*)p0, arg);
        else if (mode == 5) hipLaunchKernelGGL(dbg_vgpr_kernel<120>, dim3(blocks), dim3(threads), 0, st, (float*)p0, arg);
        else hipLaunchKernelGGL(dbg_chase_kernel, dim3(blocks), dim3(threads), 0, st, (const int*)p0, (int*)p1, arg);
    }
    OMNI_CHECK_LAUNCH("omni_debug_launch");
    return OMNI_OK;
}
