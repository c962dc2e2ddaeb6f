// The talker backbone between two attention launches as ONE persistent launch per decoder layer:
//     o_proj(l) -> gate_up(l) -> down_proj(l) -> qkv_proj(l + 1)
// (reference: the vLLM Qwen3 decoder layer reached through qwen3_tts_talker.py:341,414-422; four of the five launches of
// a layer on the launch-per-op path).  Same protocol as the code-predictor chain (cp_chain.hip, chain_gemm.cuh): a stage's
// weight slice -- 64-196 KB per workgroup, 32-96 VGPRs per lane -- goes out before the stage's flags, so the HBM stream keeps
// running through the hand-off instead of stopping for a kernel boundary (1.65 us) plus a cold first fetch; activations
// cross behind the flags with sc1 accesses.  Tiles = the launch path's (pick_tile at 64 rows), so results are bit-identical.
// Released 1.7B shape only (hidden 2048, 16 x 128 attention width, intermediate 6144, qkv 4096) at 49-64 rows; every other
// configuration stays on the launch path.
#include "chain_gemm.cuh"
#include "common.cuh"
#include "kernels.h"

#define BB_LDS_BYTES ((CH_WAVES * 12 * 64 * 16) + CH_WAVES * 64 * 4)      // combine slots of gate_up (NT * MT = 12) + rstd area

struct BbArgs {
    const uint16_t *wo, *ln2, *wgu, *wdown, *ln1_next, *wqkv_next;      // *_next == NULL: last layer, no qkv stage
    const uint16_t* wo_next;            // next layer's o_proj weights (prefetch target only) or NULL
    int prefetch;                       // wave 8 warms this workgroup's XCD L2 with the NEXT stage's weight slice
    const uint16_t* attn;               // fragment-major [64][2048]: the attention launch's output
    uint16_t* resid; float* part;       // fragment-major residual stream [64][2048] + sum(r^2) slabs [128][64]
    uint16_t* act;                      // fragment-major [64][6144]
    uint16_t* qkv;                      // row-major [B][4096]
    int B, nap; float eps;
    uint32_t* flags; int32_t* err;
    unsigned long long* stamps;
};

// ---- wave 8: the prefetcher.  A wave's loads return in order, so a compute wave cannot run a slow HBM stream ahead of the
// loads it is waiting for; this wave does nothing else.  It touches one dword per 64 bytes of the slice its OWN workgroup
// will load at the next stage's entry (same CU, hence same XCD L2: placement is fixed for the lifetime of the launch), so
// that the compute waves' register fills hit L2 (~70 GB/s per CU) instead of HBM (~24 GB/s per CU).  Nothing depends on
// these loads: a line evicted before its use is simply fetched again.  The wave joins every workgroup barrier of the compute
// waves' stage sequence (o_proj 2, gate_up 4, down_proj 3, qkv 4), which is what paces it: one burst per stage.
#define BB_THREADS (CH_THREADS + 64)
// `sink` is the ONE register every touch loads into: the loads are asynchronous and hipcc does not know it, so the register
// must stay reserved (live in / out of every asm statement) until the wave's final vmcnt(0) -- a dead output register would be
// handed to the next address computation and overwritten when the data lands.
__device__ __forceinline__ void bb_touch(const uint16_t* base, uint32_t byte_off, uint32_t bytes, uint32_t& sink) {
    const char* p = reinterpret_cast<const char*>(base) + byte_off + (threadIdx.x & 63) * 64;
    for (uint32_t off = 0; off < bytes; off += 4096) asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p + off) : "memory");
}
__device__ __forceinline__ void bb_prefetcher(const BbArgs& a) {
    const int wg = blockIdx.x, half = wg >> 7, tile = wg & 127;
    const bool on = a.prefetch != 0;
    uint32_t sink = 0;
    // o_proj stage (2 barriers): the first half of gate_up's three 64 KB tiles (all of it would overflow the XCD's 4 MB L2)
    if (on)
        for (int j = 0; j < 3; ++j) bb_touch(a.wgu, (uint32_t)(wg * 3 + j) * 65536u, 32768u, sink);
    __syncthreads(); __syncthreads();
    // gate_up stage (4): behind its gate, down_proj's 192 KB tile -- the two workgroups that share it take one half each
    __syncthreads();
    if (on) bb_touch(a.wdown, (uint32_t)tile * 196608u + (uint32_t)half * 98304u, 98304u, sink);
    __syncthreads(); __syncthreads(); __syncthreads();
    // down_proj stage (3): the next layer's qkv tiles (2 x 64 KB per workgroup pair, one each)
    __syncthreads();
    if (on && a.wqkv_next) bb_touch(a.wqkv_next, (uint32_t)(tile * 2 + half) * 65536u, 65536u, sink);
    __syncthreads(); __syncthreads();
    if (a.wqkv_next) {
        // qkv stage (4): the next layer's o_proj tile (64 KB per pair), for the segment launch behind the attention launch
        __syncthreads();
        if (on && a.wo_next) bb_touch(a.wo_next, (uint32_t)tile * 65536u + (uint32_t)half * 32768u, 32768u, sink);
        __syncthreads(); __syncthreads(); __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
}

__global__ __launch_bounds__(BB_THREADS) void bb_chain_kernel(const BbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (threadIdx.x >= CH_THREADS) {
        bb_prefetcher(a);
        return;
    }
    ChainGate g;
    chain_gate_init(g, a.flags, a.err);
    g.dom = 8;                          // gate_up's 64-row tiles tie every row group together
    g.nap = a.nap;
    const int wg = blockIdx.x;
    constexpr int H = 2048, I = 6144, NQ = 4096;
    // stage codes (error word): 0x1001 .. 0x1004
    chain_gemm<2, 1, 8, 0, OMNI_EPI_RESID, 0>(a.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, lds, g,
                                              false, 0x1001, a.stamps);
    chain_gemm<4, 3, 8, 2, OMNI_EPI_SILU_MUL_GU8, 2, true>(a.wgu, a.ln2, a.resid, a.part, H / 16, a.act, 0, nullptr, a.B, I, a.eps, wg, 0, lds, g,
                                                     true, 0x1002, a.stamps);
    chain_gemm<2, 1, 24, 0, OMNI_EPI_RESID, 4>(a.wdown, nullptr, a.act, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, lds, g,
                                               true, 0x1003, a.stamps);
    if (a.wqkv_next)
        chain_gemm<2, 2, 8, 2, OMNI_EPI_BF16, 4>(a.wqkv_next, a.ln1_next, a.resid, a.part, H / 16, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg & 127,
                                                 wg >> 7, lds, g, true, 0x1004, a.stamps);
}

OMNI_KNOB g_bb_chain = 1, g_bb_nap = 1, g_bb_prefetch = 0;      // prefetch wave: measured +0.10 ms per step, off
#ifdef OMNI_DEBUG_HOOKS
static unsigned long long* g_bb_stamps = nullptr;
extern "C" void omni_debug_bb_chain(int on) { g_bb_chain = on != 0; g_bb_prefetch = on == 2; }      // 2: with the prefetch wave's loads
extern "C" void omni_debug_bb_stamps(void* buf) { g_bb_stamps = (unsigned long long*)buf; }
#endif

bool k_bb_chain_supported(const omni_talker_desc& d, int B, bool has_ar) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 0;
    }
    return g_bb_chain && d.cp_chain && cus >= OMNI_CHAIN_WGS && d.fused_norm && d.frag_layout && !has_ar && d.moe_experts == 0 &&
           d.hidden == 2048 && d.inter == 6144 && d.head_dim == 128 && d.q_heads * 128 == 2048 &&
           (d.q_heads + 2 * d.kv_heads) * 128 == 4096 && B > 48 && B <= 64;
}

int k_bb_chain(const omni_layer_weights& w, const omni_layer_weights* next, const void* attn, void* resid, float* part, void* act, void* qkv,
               int B, float eps, uint32_t* flags, int32_t* err, void* stream) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)bb_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
        attr = true;
    }
    BbArgs a{};
    a.wo = (const uint16_t*)w.wo; a.ln2 = (const uint16_t*)w.ln2; a.wgu = (const uint16_t*)w.wgu; a.wdown = (const uint16_t*)w.wdown;
    a.ln1_next = next ? (const uint16_t*)next->ln1 : nullptr;
    a.wqkv_next = next ? (const uint16_t*)next->wqkv : nullptr;
    a.wo_next = next ? (const uint16_t*)next->wo : nullptr;
    a.prefetch = g_bb_prefetch;
    a.attn = (const uint16_t*)attn; a.resid = (uint16_t*)resid; a.part = part; a.act = (uint16_t*)act; a.qkv = (uint16_t*)qkv;
    a.B = B; a.nap = g_bb_nap; a.eps = eps; a.flags = flags; a.err = err;
#ifdef OMNI_DEBUG_HOOKS
    a.stamps = g_bb_stamps;
#endif
    hipLaunchKernelGGL(bb_chain_kernel, dim3(OMNI_CHAIN_WGS), dim3(BB_THREADS), BB_LDS_BYTES, (hipStream_t)stream, a);
    OMNI_CHECK_LAUNCH("bb_chain");
    return OMNI_OK;
}
