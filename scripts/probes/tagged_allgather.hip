// Probe (round 5): what does ONE all-to-all hand-off of the persistent chains cost with the product's protocol -- sc1 stores, every wave's
// drain, workgroup barrier, a flag word per workgroup, wave 0 polls the row group's 64 flags, barrier, sc1 loads -- against DATA-TAGGED
// granules (MI355X_MICROARCH "handoff-1to1" / "allgather": every 16-byte sc1 store carries three payload dwords and the stage number; a
// consumer loads the granules it needs and re-loads the ones whose tag is not there yet: no drain, no flag, no poll round trip, and no
// barrier between "arrived" and "use" -- a wave goes on as soon as ITS granules are in)?
// Geometry of the code predictor's chain: 256 co-resident workgroups x 8 waves, four independent row groups of 64 workgroups, one stage =
// every workgroup of a group writes its share (43 granules = 129 dwords) of the group's ~32 KB activation block and then reads the WHOLE block.
// A stage's output depends on all of its input (a checksum), so the chain cannot run ahead of its data; both modes compute the same values.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tagged_allgather scripts/probes/tagged_allgather.hip && /tmp/tagged_allgather [stages]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define WGS 256
#define THREADS 512
#ifndef GROUP
#define GROUP 64                                      // workgroups that exchange one block (64: a 16-row group of the predictor; 256: the backbone)
#endif
#ifndef GR_PER_WG
#define GR_PER_WG 43                                  // granules (tagged mode) / 3-dword items (flag mode) a workgroup produces per stage
#endif                                                // (43 x 64 x 12 B = 32 KB of payload; 86: twice the loads, the cost of an overflow granule per fragment)
#define ITEMS (GROUP * GR_PER_WG)                     // 2752 per row group and stage
#define NBUF 4
#define SC1 16

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}

// block-wide sum of a uint32 (wraps): LDS, two barriers
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* lds) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
    for (int w = 0; w < THREADS / 64; ++w) t += lds[w];
    __syncthreads();
    return t;
}

// mode 0: the product's flag protocol.  data: [4 groups][NBUF][ITEMS] granules whose 4th dword is unused; flags: [WGS] stage counts
// mode 1: tagged granules, a workgroup barrier only inside the checksum (the stand-in for the LDS combine a real stage has anyway)
template <int MODE>
__global__ __launch_bounds__(THREADS) void chain_kernel(u32x4* data, uint32_t* flags, uint32_t* out, int stages, int* err) {
    __shared__ uint32_t lds[16];
    const int wg = blockIdx.x, grp = wg / GROUP, wi = wg % GROUP;
    constexpr int NGRP = WGS / GROUP;
    const __amdgpu_buffer_rsrc_t drs = rsrc(data), frs = rsrc(flags);
    uint32_t val = 0x9E3779B9u * (uint32_t)(wg + 1);
    for (int s = 1; s <= stages; ++s) {
        // ---- produce: this workgroup's 43 granules of stage s (values depend on everything read in stage s - 1)
        const uint32_t base = (uint32_t)((grp * NBUF + (s & (NBUF - 1))) * ITEMS);
        for (int i = threadIdx.x; i < GR_PER_WG; i += THREADS) {
            const uint32_t g = base + wi * GR_PER_WG + i;
            const uint32_t d0 = val + 3u * i;
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){d0, d0 + 1u, d0 + 2u, (uint32_t)s}, drs, g * 16u, 0, SC1);
        }
        if (MODE == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // the flag word in 8 copies 8 KB apart, one store instruction; a workgroup polls copy (blockIdx.x & 7) -- the product's form
            // (coherent.cuh chain_flag_publish / chain_flag_copy: the pollers of a domain do not all sit on the same four lines)
            if (threadIdx.x < 8) __builtin_amdgcn_raw_buffer_store_b32((uint32_t)s, frs, (threadIdx.x * 2048u + wg) * 4u, 0, SC1);
            if (threadIdx.x < GROUP / 4) {             // wave 0 polls the group's flags, 4 per lane
                unsigned spins = 0;
                for (;;) {
                    asm volatile("" ::: "memory");
                    const u32x4 f = __builtin_amdgcn_raw_buffer_load_b128(frs, (uint32_t)((wg & 7) * 2048 + grp * GROUP + threadIdx.x * 4) * 4u, 0, SC1);
                    const bool behind = (int)(f[0] - s) < 0 || (int)(f[1] - s) < 0 || (int)(f[2] - s) < 0 || (int)(f[3] - s) < 0;
                    if (__builtin_amdgcn_ballot_w64(behind) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { *err = s; break; }
                }
            }
            __syncthreads();
        }
        // ---- consume: the whole block of the group, 16 bytes per load, ITEMS / THREADS = 5.4 loads per thread
        uint32_t acc = 0;
        constexpr int PER = (ITEMS + THREADS - 1) / THREADS;
        if (MODE == 0) {
            u32x4 v[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const uint32_t i = threadIdx.x + k * THREADS;
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(drs, (base + (i < ITEMS ? i : 0)) * 16u, 0, SC1);
            }
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (threadIdx.x + k * THREADS < ITEMS) acc += v[k][0] + v[k][1] + v[k][2];
        } else {
            u32x4 v[PER];
            static_assert(PER <= 64, "pending mask");
            unsigned long long pending = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (threadIdx.x + k * THREADS < ITEMS) pending |= 1ull << k;
            unsigned spins = 0;
            while (__builtin_amdgcn_ballot_w64(pending != 0) != 0) {        // the WAVE goes on when all of ITS granules carry the tag
#pragma unroll
                for (int k = 0; k < PER; ++k)
                    if (pending & (1ull << k)) v[k] = __builtin_amdgcn_raw_buffer_load_b128(drs, (base + threadIdx.x + k * THREADS) * 16u, 0, SC1);
#pragma unroll
                for (int k = 0; k < PER; ++k)
                    if ((pending & (1ull << k)) && v[k][3] == (uint32_t)s) pending &= ~(1ull << k);
                if (__builtin_amdgcn_ballot_w64(pending != 0) != 0) __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { *err = s; break; }
            }
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (threadIdx.x + k * THREADS < ITEMS) acc += v[k][0] + v[k][1] + v[k][2];
        }
        val = block_sum(acc, lds) * 2654435761u + (uint32_t)wg + (uint32_t)s;
    }
    if (threadIdx.x == 0) out[wg] = val;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 4000;
    u32x4* data; uint32_t *flags, *out; int* err;
    printf("%d workgroups per block, %d granules each: %.1f KB of payload per block and stage, %d 16-byte loads per thread\n", GROUP, GR_PER_WG,
           ITEMS * 12 / 1024.0, (ITEMS + THREADS - 1) / THREADS);
    CK(hipMalloc(&data, (size_t)(WGS / GROUP) * NBUF * ITEMS * 16));
    CK(hipMalloc(&flags, 8 * 2048 * 4));
    CK(hipMalloc(&out, WGS * 4));
    CK(hipMalloc(&err, 4));
    uint32_t h[2][WGS];
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 2; ++mode) {
            CK(hipMemset(data, 0, (size_t)(WGS / GROUP) * NBUF * ITEMS * 16));
            CK(hipMemset(flags, 0, 8 * 2048 * 4));
            CK(hipMemset(err, 0, 4));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(WGS), dim3(THREADS), 0, 0, data, flags, out, stages, err);
            else hipLaunchKernelGGL(chain_kernel<1>, dim3(WGS), dim3(THREADS), 0, 0, data, flags, out, stages, err);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            int herr = 0;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h[mode], out, WGS * 4, hipMemcpyDeviceToHost));
            uint32_t cs = 0;
            for (int i = 0; i < WGS; ++i) cs = cs * 31u + h[mode][i];
            printf("%-40s %d stages: %8.3f ms = %6.3f us per stage   checksum %08x%s\n",
                   mode == 0 ? "flags x 8 copies (drain, poll, barrier)" : "tagged 16-byte granules", stages, ms, ms * 1e3 / stages, cs,
                   herr ? "   TIMED OUT" : "");
        }
    int same = 1;
    for (int i = 0; i < WGS; ++i) same &= h[0][i] == h[1][i];
    printf("both protocols computed %s values\n", same ? "the SAME" : "DIFFERENT");
    return 0;
}
