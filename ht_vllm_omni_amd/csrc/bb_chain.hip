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
    const uint16_t* attn;               // fragment-major [64][2048]: the attention launch's output
    uint16_t* resid; float* part;       // fragment-major residual stream [64][2048] + sum(r^2) slabs [128][64]
    uint16_t* act;                      // fragment-major [64][6144]
    uint16_t* qkv;                      // row-major [B][4096]
    int B, nap; float eps;
    uint32_t* flags; int32_t* err;
    unsigned long long* stamps;
};

__global__ __launch_bounds__(CH_THREADS) void bb_chain_kernel(const BbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    ChainGate g;
    chain_gate_init(g, a.flags, a.err);
    g.dom = 8;                          // gate_up's 64-row tiles tie every row group together
    g.nap = a.nap;
    const int wg = blockIdx.x;
    constexpr int H = 2048, I = 6144, NQ = 4096;
    // stage codes (error word): 0x1001 .. 0x1004
    chain_gemm<2, 1, 8, 0, OMNI_EPI_RESID, 0>(a.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, lds, g,
                                              false, 0x1001, a.stamps);
    chain_gemm<4, 3, 8, 2, OMNI_EPI_SILU_MUL_GU8, 2>(a.wgu, a.ln2, a.resid, a.part, H / 16, a.act, 0, nullptr, a.B, I, a.eps, wg, 0, lds, g,
                                                     true, 0x1002, a.stamps);
    chain_gemm<2, 1, 24, 0, OMNI_EPI_RESID, 4>(a.wdown, nullptr, a.act, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, lds, g,
                                               true, 0x1003, a.stamps);
    if (a.wqkv_next)
        chain_gemm<2, 2, 8, 2, OMNI_EPI_BF16, 0>(a.wqkv_next, a.ln1_next, a.resid, a.part, H / 16, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg & 127,
                                                 wg >> 7, lds, g, true, 0x1004, a.stamps);
}

OMNI_KNOB g_bb_chain = 1, g_bb_nap = 1;
#ifdef OMNI_DEBUG_HOOKS
static unsigned long long* g_bb_stamps = nullptr;
extern "C" void omni_debug_bb_chain(int on) { g_bb_chain = on; }
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
    a.attn = (const uint16_t*)attn; a.resid = (uint16_t*)resid; a.part = part; a.act = (uint16_t*)act; a.qkv = (uint16_t*)qkv;
    a.B = B; a.nap = g_bb_nap; a.eps = eps; a.flags = flags; a.err = err;
#ifdef OMNI_DEBUG_HOOKS
    a.stamps = g_bb_stamps;
#endif
    hipLaunchKernelGGL(bb_chain_kernel, dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), BB_LDS_BYTES, (hipStream_t)stream, a);
    OMNI_CHECK_LAUNCH("bb_chain");
    return OMNI_OK;
}
