// Micro-probe behind the round-4 finding (NOTEBOOK "Round 4", pa_body.cuh): packed fp32 operations whose operand is BROADCAST by
// op_sel / op_sel_hi, each form against scalar v_fma_f32 / v_mul_f32 on the same data, counted per form.  Run alone and beside another
// process (scripts/probes/run_pkfma_probe.sh: the Code2Wav loop of tests/test_gpu_colocation.py).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pkfma_src1 scripts/probes/pkfma_src1.hip && /tmp/pkfma_src1 [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <chrono>

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define NFORMS 10
static const char* FORM[NFORMS] = {
    "fma  src0 low  broadcast   op_sel_hi:[0,1,1]", "fma  src0 high broadcast   op_sel:[1,0,0]",
    "fma  src1 low  broadcast   op_sel_hi:[1,0,1]", "fma  src1 high broadcast   op_sel:[0,1,0]",
    "fma  src2 low  broadcast   op_sel_hi:[1,1,0]", "fma  src2 high broadcast   op_sel:[0,0,1]",
    "mul  src0 low  broadcast   op_sel_hi:[0,1]  ", "mul  src1 low  broadcast   op_sel_hi:[1,0]  ",
    "mul  src1 high broadcast   op_sel:[0,1]     ", "fma  no selector (control)                  "};

#define SFMA(acc, a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define SMUL(d, a, b) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define CHECK(i, got, e0, e1) bad[i] += (__float_as_uint(got[0]) != __float_as_uint(e0)) + (__float_as_uint(got[1]) != __float_as_uint(e1))

__global__ __launch_bounds__(256) void probe(const uint32_t* __restrict__ words, const float* __restrict__ qv, unsigned long long* out,
                                             int n_words, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const f32x2 q = {qv[(tid * 4 + 0) & 1023], qv[(tid * 4 + 1) & 1023]};
    const f32x2 c = {qv[(tid * 4 + 2) & 1023], qv[(tid * 4 + 3) & 1023]};
    unsigned long long bad[NFORMS];
    for (int i = 0; i < NFORMS; ++i) bad[i] = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t w = words[(tid + it * 977) % n_words] & 0x7E7E7E7Eu;       // finite e4m3fn bytes
        f32x2 k;                                                                   // a pair fresh out of the convert, as in the attention loop
        asm volatile("v_cvt_pk_f32_fp8_e32 %0, %1" : "=v"(k) : "v"(w));
        f32x2 d;
        float e0, e1;
#define PK3(i, MODS, A, B, C, X0, Y0, Z0, X1, Y1, Z1)                                                   \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=&v"(d) : "v"(A), "v"(B), "v"(C));          \
        e0 = Z0; SFMA(e0, X0, Y0); e1 = Z1; SFMA(e1, X1, Y1); CHECK(i, d, e0, e1)
        PK3(0, "op_sel_hi:[0,1,1]", k, q, c, k[0], q[0], c[0], k[0], q[1], c[1]);
        PK3(1, "op_sel:[1,0,0]", k, q, c, k[1], q[0], c[0], k[1], q[1], c[1]);
        PK3(2, "op_sel_hi:[1,0,1]", q, k, c, q[0], k[0], c[0], q[1], k[0], c[1]);
        PK3(3, "op_sel:[0,1,0]", q, k, c, q[0], k[1], c[0], q[1], k[1], c[1]);
        PK3(4, "op_sel_hi:[1,1,0]", q, c, k, q[0], c[0], k[0], q[1], c[1], k[0]);
        PK3(5, "op_sel:[0,0,1]", q, c, k, q[0], c[0], k[1], q[1], c[1], k[1]);
        PK3(9, "", q, k, c, q[0], k[0], c[0], q[1], k[1], c[1]);
#define PK2(i, MODS, A, B, X0, Y0, X1, Y1)                                                              \
        asm volatile("v_pk_mul_f32 %0, %1, %2 " MODS : "=&v"(d) : "v"(A), "v"(B));                      \
        SMUL(e0, X0, Y0); SMUL(e1, X1, Y1); CHECK(i, d, e0, e1)
        PK2(6, "op_sel_hi:[0,1]", k, q, k[0], q[0], k[0], q[1]);
        PK2(7, "op_sel_hi:[1,0]", q, k, q[0], k[0], q[1], k[0]);
        PK2(8, "op_sel:[0,1]", q, k, q[0], k[1], q[1], k[1]);
    }
    for (int i = 0; i < NFORMS; ++i)
        if (bad[i]) atomicAdd(&out[i], bad[i]);
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
    if (hipMalloc(&dw, n_words * 4) != hipSuccess || hipMalloc(&dq, 4096) != hipSuccess || hipMalloc(&dbad, 8 * NFORMS) != hipSuccess) return 2;
    (void)hipMemcpy(dw, hw, n_words * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dq, hq, 4096, hipMemcpyHostToDevice);
    (void)hipMemset(dbad, 0, 8 * NFORMS);
    const auto t0 = std::chrono::steady_clock::now();
    long long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, dw, dq, dbad, n_words, 256);
        if (hipDeviceSynchronize() != hipSuccess) return 3;
        ++launches;
    }
    unsigned long long bad[NFORMS];
    (void)hipMemcpy(bad, dbad, 8 * NFORMS, hipMemcpyDeviceToHost);
    printf("launches %lld, %.3g executions of every form; results differing from the scalar instruction:\n", launches,
           (double)launches * 2048 * 256 * 256);
    for (int i = 0; i < NFORMS; ++i) printf("   %s  %llu\n", FORM[i], bad[i]);
    return 0;
}
