// One-shot all-reduce of the tensor-parallel decode step over peer-mapped buffers (xGMI), fused with the residual add and the
// sum-of-squares slabs of the norm-free stream.
//
// Why not RCCL here: the message is [B, H] bf16 = 256 KiB at most, 56 times per step, inside a hipGraph.  A ring needs
// 2 (N - 1) serial hops of ~10 us each on point-to-point xGMI; one-shot = every rank reads every peer's partial buffer
// directly (7 links in parallel, 256 KiB each) behind ONE flag exchange.  And because the sum lands in this kernel, the
// residual add and the per-row sum(r^2) slabs ride along: tensor-parallel ranks keep the norm-free residual stream
// (omni_gemm_xnorm consumers) instead of falling back to separate RMSNorm launches.  RCCL stays for prefill-sized messages.
//
// Protocol (per call, epoch e = *epoch + 1, read by every workgroup at its start):
//   producer  the GEMM before this launch wrote this rank's partial [M16, H] bf16, FRAGMENT-major, into its symmetric
//             buffer `data[rank]` (fine-grained device memory, hipIpc-mapped into every peer);
//   arrive    workgroup 0, lane p < world: system-scope release store  flags[p][rank] = e  (peer p's flag array);
//   wait      every workgroup: lane p polls  flags[rank][p] >= e  (system-scope relaxed load + s_sleep, BOUNDED: a peer that
//             never arrives sets *error and the kernel carries on -- a wrong step, never a hung GPU); then a system-scope
//             acquire fence;
//   reduce    sum over ranks 0 .. world-1 IN RANK ORDER in fp32 (every rank computes the same bits), one rounding to bf16,
//             r = bf16(r + sum) on the fragment-major residual stream, slab[col / 16][row] = this group's share of sum(r^2);
//   leave     the last workgroup (ticket counter) bumps *epoch.
// Buffer reuse: attn and mlp all-reduces alternate between two data buffers; a rank can only pass the flag wait of call
// e + 1 after every peer STARTED call e + 1, i.e. finished reading for call e -- so the buffer of call e is free for the
// GEMM of call e + 2.
#include <string.h>

#include "common.cuh"
#include "kernels.h"

#define AR_MAX_WORLD 8
// A peer is waited for by WALL CLOCK (a poll of fine-grained peer memory takes ~1 us, so an iteration count of 2^27 was
// minutes, not seconds): AR_WAIT_TICKS of the constant 100 MHz counter (s_memrealtime), read once per 1024 polls.  5 s covers a
// rank that is late by a graph upload or a GC pause.
#define AR_WAIT_TICKS 500000000ull

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ArArgs {
    int world, rank;
    const uint16_t* data[AR_MAX_WORLD];
    uint32_t* flags[AR_MAX_WORLD];
    uint32_t* epoch;           // [0] epoch, [1] ticket counter
    int32_t* error;
    uint16_t* r_io; int accumulate; float* part; int pstride;
    uint16_t* out_rm;          // optional row-major copy of the SUM (bf16 [M, H]) for consumers outside the fused stream
    int M, H;
};

// grid = H / 32 workgroups of 4 waves; wave w = row tile w (16 rows); lane = (k chunk of 8 columns, row)
__global__ __launch_bounds__(256) void allreduce_resid_kernel(const ArArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t e = __hip_atomic_load(a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (a.world > 1) {
        if (blockIdx.x == 0 && threadIdx.x < a.world)
            __hip_atomic_store(a.flags[threadIdx.x] + a.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // a peer that failed to arrive once (error word set) is not waited for again: a dead rank costs ONE timeout, not one
        // per all-reduce of every remaining step; the caller reads the error word and voids the results
        if (threadIdx.x < a.world && __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            const uint32_t* f = a.flags[a.rank] + threadIdx.x;
            uint32_t spins = 0;
            unsigned long long t0 = 0;
            while ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0) {
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (spins == 1024u) t0 = now;
                    else if (now - t0 > AR_WAIT_TICKS) {
                        // peer `threadIdx.x` did not arrive.  The word goes into EVERY rank's control block (the error word sits at
                        // the same offset from the flags in each of them: tp_comm.CTL layout), so that the ranks that did see
                        // their peers learn of it within the step too and all of them stop
                        const int code = 1 + (int)threadIdx.x + 256 * (a.rank + 1);
                        const ptrdiff_t eoff = a.error - reinterpret_cast<int32_t*>(a.flags[a.rank]);
                        for (int q = 0; q < a.world; ++q)
                            __hip_atomic_store(reinterpret_cast<int32_t*>(a.flags[q]) + eoff, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // system scope: the peers' partials below are fresh
        __syncthreads();
    }
    const int tiles = (a.M + 15) >> 4;
    if (wave < tiles) {
        const int ksteps = a.H >> 5;
        const size_t off = ((size_t)wave * ksteps + blockIdx.x) * 512 + lane * 8;       // frag_off(16 wave + row, 32 bx + 8 kc, H)
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int p = 0; p < a.world; ++p) {                    // rank order: identical bits on every rank
            const u32x4 v = *reinterpret_cast<const u32x4*>(a.data[p] + off);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[2 * j] += bf_lo(v[j]); acc[2 * j + 1] += bf_hi(v[j]); }
        }
        const int row = wave * 16 + (lane & 15), kc = lane >> 4;
        u32x4 old = (u32x4){0, 0, 0, 0};
        if (a.r_io && a.accumulate) old = *reinterpret_cast<const u32x4*>(a.r_io + off);
        u32x4 o, s;
        float rr[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s0 = bfround(acc[2 * j]), s1 = bfround(acc[2 * j + 1]);          // the all-reduced delta, bf16
            s[j] = pack_bf2(s0, s1);
            rr[2 * j] = a.accumulate ? bfround(bf_lo(old[j]) + s0) : s0;
            rr[2 * j + 1] = a.accumulate ? bfround(bf_hi(old[j]) + s1) : s1;
            o[j] = pack_bf2(rr[2 * j], rr[2 * j + 1]);
        }
        // sum(r^2) of the slab's 16 columns in the order of the chains' residual epilogue (chain_gemm.cuh, gemm.hip RESID: four columns left
        // to right, then groups 0 + 1 and 2 + 3, then the halves) -- so that the all-reduce inside a backbone chain stage (round 6) and this
        // launch write the same slab bits
        float ss = sq4_sum(rr[0], rr[1], rr[2], rr[3]) + sq4_sum(rr[4], rr[5], rr[6], rr[7]);
        if (a.r_io) *reinterpret_cast<u32x4*>(a.r_io + off) = o;
        if (a.out_rm && row < a.M) *reinterpret_cast<u32x4*>(a.out_rm + (size_t)row * a.H + blockIdx.x * 32 + kc * 8) = s;
        if (a.part) {
            ss = xor16_sum(ss);                                // k chunks {0,1} -> slab 2 bx, {2,3} -> slab 2 bx + 1
            if ((kc & 1) == 0 && row < a.M) a.part[(size_t)(2 * blockIdx.x + (kc >> 1)) * a.pstride + row] = ss;
        }
    }
    // last workgroup out closes the epoch (every workgroup has read it by now: the ticket comes after its own read)
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(a.epoch + 1, 1u);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(a.epoch + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.epoch, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" int omni_allreduce_resid(const omni_ar_peers* p, void* r_io, int accumulate, float* partials, int pstride, void* out_rowmajor,
                                    int M, int H, void* stream) {
    OMNI_CHECK_ARG(p && p->world >= 1 && p->world <= AR_MAX_WORLD && p->rank >= 0 && p->rank < p->world, "omni_allreduce_resid: bad peer table");
    OMNI_CHECK_ARG(p->epoch && p->error, "omni_allreduce_resid: null epoch / error word");
    OMNI_CHECK_ARG(M >= 1 && M <= 64 && H % 32 == 0, "omni_allreduce_resid: M=%d H=%d (M <= 64, H %% 32 == 0)", M, H);
    OMNI_CHECK_ARG(r_io || out_rowmajor, "omni_allreduce_resid: no output");
    OMNI_CHECK_ARG(!partials || pstride >= M, "omni_allreduce_resid: slab stride %d < M", pstride);
    ArArgs a{};
    a.world = p->world; a.rank = p->rank;
    for (int i = 0; i < p->world; ++i) {
        OMNI_CHECK_ARG(p->data[i] && (p->world == 1 || p->flags[i]), "omni_allreduce_resid: peer %d not mapped", i);
        a.data[i] = (const uint16_t*)p->data[i];
        a.flags[i] = p->flags[i];
    }
    a.epoch = p->epoch; a.error = p->error;
    a.r_io = (uint16_t*)r_io; a.accumulate = accumulate; a.part = partials; a.pstride = pstride; a.out_rm = (uint16_t*)out_rowmajor;
    a.M = M; a.H = H;
    hipLaunchKernelGGL(allreduce_resid_kernel, dim3(H / 32), dim3(256), 0, (hipStream_t)stream, a);
    OMNI_CHECK_LAUNCH("omni_allreduce_resid");
    return OMNI_OK;
}

// ---- symmetric memory: fine-grained device allocations that peers map through hipIpc (one process per GPU)
extern "C" int omni_ar_alloc(int64_t bytes, void** ptr, void* ipc_handle64) {
    OMNI_CHECK_ARG(bytes > 0 && ptr && ipc_handle64, "omni_ar_alloc: bad arguments");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { omni_set_error("omni_ar_alloc: hipExtMallocWithFlags(%lld): %s", (long long)bytes, hipGetErrorString(e)); return OMNI_EHIP; }
    e = hipMemset(p, 0, (size_t)bytes);
    if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(ipc_handle64), p);
    if (e != hipSuccess) { (void)hipFree(p); omni_set_error("omni_ar_alloc: %s", hipGetErrorString(e)); return OMNI_EHIP; }
    *ptr = p;
    return OMNI_OK;
}
extern "C" int omni_ar_open(const void* ipc_handle64, void** ptr) {
    OMNI_CHECK_ARG(ipc_handle64 && ptr, "omni_ar_open: bad arguments");
    hipIpcMemHandle_t h = *reinterpret_cast<const hipIpcMemHandle_t*>(ipc_handle64);
    hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { omni_set_error("omni_ar_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e)); return OMNI_EHIP; }
    return OMNI_OK;
}
extern "C" int omni_ar_close(void* ptr) {
    if (ptr && hipIpcCloseMemHandle(ptr) != hipSuccess) { omni_set_error("omni_ar_close failed"); return OMNI_EHIP; }
    return OMNI_OK;
}
extern "C" int omni_ar_free(void* ptr) {
    if (ptr && hipFree(ptr) != hipSuccess) { omni_set_error("omni_ar_free failed"); return OMNI_EHIP; }
    return OMNI_OK;
}
