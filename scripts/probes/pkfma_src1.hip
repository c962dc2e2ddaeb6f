// Micro-probe behind the round-4 finding (NOTEBOOK "Round 4", pa_body.cuh): does v_pk_fma_f32 with a BROADCAST src1 pair that was just
// written by v_cvt_pk_f32_fp8 differ from the same FMAs with the broadcast in src0 / from scalar FMAs -- alone, and beside other work?
// Every lane runs the three forms on the same data in the same order of operations and counts bitwise mismatches.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pkfma_src1 scripts/probes/pkfma_src1.hip && /tmp/pkfma_src1 [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <chrono>

typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void probe(const uint32_t* __restrict__ words, const float* __restrict__ qv, unsigned long long* bad,
                                             int n_words, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 q0 = {qv[(tid * 4 + 0) & 1023], qv[(tid * 4 + 1) & 1023]};
    f32x2 q1 = {qv[(tid * 4 + 2) & 1023], qv[(tid * 4 + 3) & 1023]};
    unsigned long long mism1 = 0, mism0 = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t w = words[(tid + it * 977) % n_words] & 0x7E7E7E7Eu;       // finite e4m3fn bytes
        f32x2 a_src1 = {0.f, 0.f}, a_src0 = {0.f, 0.f};
        float s0 = 0.f, s1 = 0.f;
        f32x2 k01, k23;
        // the attention loop's pattern: two converts (the second SDWA), then four packed FMAs reading the fresh pairs
        asm volatile(
            "v_cvt_pk_f32_fp8_e32 %[k01], %[w]\n\t"
            "v_cvt_pk_f32_fp8_sdwa %[k23], %[w] src0_sel:WORD_1\n\t"
            "v_pk_fma_f32 %[a1], %[q0], %[k01], %[a1] op_sel_hi:[1,0,1]\n\t"      // k01.lo broadcast in src1
            "v_pk_fma_f32 %[a1], %[q1], %[k01], %[a1] op_sel:[0,1,0]\n\t"         // k01.hi broadcast in src1
            "v_pk_fma_f32 %[a1], %[q0], %[k23], %[a1] op_sel_hi:[1,0,1]\n\t"
            "v_pk_fma_f32 %[a1], %[q1], %[k23], %[a1] op_sel:[0,1,0]\n\t"
            : [k01] "=&v"(k01), [k23] "=&v"(k23), [a1] "+v"(a_src1)
            : [w] "v"(w), [q0] "v"(q0), [q1] "v"(q1));
        asm volatile(
            "v_cvt_pk_f32_fp8_e32 %[k01], %[w]\n\t"
            "v_cvt_pk_f32_fp8_sdwa %[k23], %[w] src0_sel:WORD_1\n\t"
            "v_pk_fma_f32 %[a0], %[k01], %[q0], %[a0] op_sel_hi:[0,1,1]\n\t"      // the same products, broadcast in src0
            "v_pk_fma_f32 %[a0], %[k01], %[q1], %[a0] op_sel:[1,0,0]\n\t"
            "v_pk_fma_f32 %[a0], %[k23], %[q0], %[a0] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[a0], %[k23], %[q1], %[a0] op_sel:[1,0,0]\n\t"
            : [k01] "=&v"(k01), [k23] "=&v"(k23), [a0] "+v"(a_src0)
            : [w] "v"(w), [q0] "v"(q0), [q1] "v"(q1));
        const f32x2 r01 = __builtin_amdgcn_cvt_pk_f32_fp8(w, false), r23 = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
        // the reference chain as SCALAR v_fma_f32 (inline asm: hipcc packs a C++ loop into the very form under test)
#define SFMA(acc, a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
        SFMA(s0, q0[0], r01[0]); SFMA(s1, q0[1], r01[0]);
        SFMA(s0, q1[0], r01[1]); SFMA(s1, q1[1], r01[1]);
        SFMA(s0, q0[0], r23[0]); SFMA(s1, q0[1], r23[0]);
        SFMA(s0, q1[0], r23[1]); SFMA(s1, q1[1], r23[1]);
#undef SFMA
        mism1 += (__float_as_uint(a_src1[0]) != __float_as_uint(s0)) + (__float_as_uint(a_src1[1]) != __float_as_uint(s1));
        mism0 += (__float_as_uint(a_src0[0]) != __float_as_uint(s0)) + (__float_as_uint(a_src0[1]) != __float_as_uint(s1));
    }
    if (mism1) atomicAdd(&bad[0], mism1);
    if (mism0) atomicAdd(&bad[1], mism0);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    const int n_words = 1 << 20;
    uint32_t* hw = (uint32_t*)malloc(n_words * 4);
    float hq[1024];
    srand(7);
    for (int i = 0; i < n_words; ++i) hw[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    for (int i = 0; i < 1024; ++i) hq[i] = (float)(rand() % 2001 - 1000) / 997.0f;
    uint32_t* dw; float* dq; unsigned long long* dbad;
    if (hipMalloc(&dw, n_words * 4) != hipSuccess || hipMalloc(&dq, 4096) != hipSuccess || hipMalloc(&dbad, 16) != hipSuccess) return 2;
    (void)hipMemcpy(dw, hw, n_words * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dq, hq, 4096, hipMemcpyHostToDevice);
    (void)hipMemset(dbad, 0, 16);
    const auto t0 = std::chrono::steady_clock::now();
    long long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, dw, dq, dbad, n_words, 256);
        if (hipDeviceSynchronize() != hipSuccess) return 3;
        ++launches;
    }
    unsigned long long bad[2];
    (void)hipMemcpy(bad, dbad, 16, hipMemcpyDeviceToHost);
    printf("launches %lld  FMA groups %.3g  mismatches vs scalar: src1-broadcast form %llu, src0-broadcast form %llu\n", launches,
           (double)launches * 2048 * 256 * 256, bad[0], bad[1]);
    return 0;
}
