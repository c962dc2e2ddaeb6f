// Probe (round 5): where does a K-tile of the two-group prefill GEMM tile (gemm_prefill.hip, gemm_tile_pp_kernel) spend its cycles?
// Compiles the product source with -DPP_STAMPS: one workgroup keeps the shader clock at every section boundary of K-tiles 8 ... 23 for the
// first wave of each group.  Per phase: [loads + 4 LDS-DMA pieces | barrier + lgkmcnt(0)] -> stamp b -> [MFMAs issued] -> stamp c ->
// [closing barrier] -> stamp d.  d(prev) -> b = own loads, the wait for the fragments and for the OTHER group's MFMA section;
// b -> c = this wave's MFMA issue; c -> d = the wait for the other group's loads.  PP_STAMPS=2 adds stamp a in front of the first barrier.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPP_STAMPS=1 -Iinclude -Iht_vllm_omni_amd/csrc -o /tmp/gemm_pp_stamps scripts/probes/gemm_pp_stamps.hip
//   /tmp/gemm_pp_stamps [tile_hint 5 | 6 | 7 = 256 | 224 | 192 rows] [M] [N] [K]
#include "../../ht_vllm_omni_amd/csrc/gemm_prefill.hip"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

void omni_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_kernel(uint16_t* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((h & 0xFFFF) / 65536.0f - 0.5f) * 0.25f;
        p[i] = (uint16_t)(__float_as_uint(v) >> 16);
    }
}

int main(int argc, char** argv) {
    const int hint = argc > 1 ? atoi(argv[1]) : 5;
    const int M = argc > 2 ? atoi(argv[2]) : 6438, N = argc > 3 ? atoi(argv[3]) : 2048, K = argc > 4 ? atoi(argv[4]) : 2048;
    uint16_t *x, *w, *o;
    CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&w, (size_t)N * K * 2)); CK(hipMalloc(&o, (size_t)M * N * 2));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, x, (size_t)M * K, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, w, (size_t)N * K, 2u);
    omni_tile_gemm g = {};
    g.x = x; g.x_rows = M; g.ldx = K; g.seg_len = K; g.seg_rows = 1; g.w = w; g.out = o; g.ldo = N; g.M = M; g.N = N; g.K = K; g.tile_hint = hint;
    for (int wg : {0, 100}) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pp_stamp_wg), &wg, sizeof(int)));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 20; ++i) if (omni_gemm_tile(&g, nullptr)) return 1;
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) omni_gemm_tile(&g, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        static unsigned long long st[2][1024];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_pp_stamps), sizeof(st)));
#if PP_STAMPS == 3
        printf("tile_hint %d  M %d N %d K %d: %.1f us per launch; workgroup %d\n", hint, M, N, K, ms * 50, wg);
        for (int grp = 0; grp < 2; ++grp) {
            const unsigned long long* t = &st[grp][1];       // (shader clock, 100 MHz clock) x {entry, loop start, loop end, end}
            const int nkt = (K / 32 + 1) / 2, nkt2 = (nkt + 1) & ~1;
            const double ghz = (double)(t[4] - t[2]) / ((double)(t[5] - t[3]) * 10.0);
            printf("  group %d: prologue %llu cycles, K loop %llu = %.0f per K-tile (%d K-tiles run) at %.2f GHz (ideal 2048), epilogue %llu; "
                   "entry -> end %.1f us\n", grp, t[2] - t[0], t[4] - t[2], (double)(t[4] - t[2]) / nkt2, nkt2, ghz, t[6] - t[4], (t[7] - t[1]) / 100.0);
        }
        continue;
#endif
        const int per = PP_STAMPS == 2 ? 4 : 3, phases = 2;
        printf("tile_hint %d  M %d N %d K %d: %.1f us per launch (stamped build); workgroup %d, K-tiles 8-23, shader cycles (median over the 16 K-tiles)\n", hint, M, N, K, ms * 50, wg);
        for (int grp = 0; grp < 2; ++grp) {
            const int n = (int)st[grp][0];
            const unsigned long long* t = &st[grp][1];
            const int ktiles = (n - 1) / (phases * per);
            printf("  group %d (%d stamps, %d K-tiles): total per K-tile %llu\n", grp, n, ktiles, ktiles ? (t[n - 1] - t[0]) / ktiles : 0ull);
            for (int ph = 0; ph < phases; ++ph) {
                std::vector<long long> seg[4];
                for (int kt = 0; kt < ktiles; ++kt) {
                    const int base = (kt * phases + ph) * per;       // t[base] = the previous closing barrier passed
                    for (int s2 = 0; s2 < per; ++s2) seg[s2].push_back((long long)(t[base + s2 + 1] - t[base + s2]));
                }
                printf("    phase %d:", ph + 1);
                const char* names3[] = {"loads+barrier+frags", "mfma issue", "closing barrier"};
                const char* names4[] = {"loads (frags in)", "barrier", "mfma issue", "closing barrier"};
                for (int s2 = 0; s2 < per; ++s2) {
                    std::sort(seg[s2].begin(), seg[s2].end());
                    printf("  %s %lld", per == 3 ? names3[s2] : names4[s2], seg[s2].empty() ? 0ll : seg[s2][seg[s2].size() / 2]);
                }
                printf("\n");
            }
        }
    }
    return 0;
}
