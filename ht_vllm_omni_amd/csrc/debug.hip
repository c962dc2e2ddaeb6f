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

extern "C" int omni_debug_launch(int mode, int blocks, int threads, void* p0, void* p1, int arg, int reps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    for (int r = 0; r < reps; ++r) {
        if (mode == 0) hipLaunchKernelGGL(dbg_empty_kernel, dim3(blocks), dim3(threads), 0, st);
        else if (mode == 1) hipLaunchKernelGGL(dbg_touch_kernel, dim3(blocks), dim3(threads), 0, st, (float*)p0);
        else hipLaunchKernelGGL(dbg_chase_kernel, dim3(blocks), dim3(threads), 0, st, (const int*)p0, (int*)p1, arg);
    }
    OMNI_CHECK_LAUNCH("omni_debug_launch");
    return OMNI_OK;
}
