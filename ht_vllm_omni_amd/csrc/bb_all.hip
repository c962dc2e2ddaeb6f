// The talker backbone of one decode step as ONE persistent launch (reference: the vLLM Qwen3 decoder stack reached through
// qwen3_tts_talker.py:341,414-422, then compute_logits 424-443):
//     qkv(0) -> [ attention(l) -> o_proj(l) -> gate_up(l) -> down_proj(l) -> qkv(l + 1) ] x L
// Round 3 ran it as 1 + 2 L launches (a paged-attention launch and a four-stage persistent launch per layer): 56 kernel
// boundaries, each with its grid ramp, a cold first fetch and a store drain.  Here the paged attention is a STAGE of the same
// grid -- 256 co-resident workgroups x 8 waves, a workgroup works two (row, kv head) pairs with four waves each -- behind the
// same flag hand-off as the GEMM stages (coherent.cuh): the first K / V batch of a pair is in flight before the qkv stage's
// flags are seen (the history is final since an earlier launch), q / k / v of the new token cross with sc1 loads, the
// attention output with sc1 stores.  Arithmetic = the launch path's (pa_body.cuh, chain_gemm.cuh): bit-identical.
// Released 1.7B shape only (hidden 2048, 16 x 128 attention width over 8 kv heads, intermediate 6144), 49-64 rows, single rank.
#include <stddef.h>

#include <vector>

#include "chain_gemm.cuh"
#include "common.cuh"
#include "kernels.h"
#include "pa_body.cuh"

#define BBA_PAIR_FLOATS PA_LDS_FLOATS(2)      // one (row, kv head) pair's attention scratch (G = 2)
#define BBA_LDS_BYTES ((CH_WAVES * 12 * 64 * 16) + CH_WAVES * 64 * 4)           // the GEMM stages' combine slots (gate_up: 12 tiles in ONE pass) + rstd area
static_assert(2 * BBA_PAIR_FLOATS * 4 <= BBA_LDS_BYTES, "bb_all: two attention pairs must fit the chain's LDS");

// one decoder layer's pointers, in DEVICE memory (the table is written once at engine creation; k / v scale may be rewritten by
// the host when the fp8 scales are calibrated -- the captured launch reads them here)
struct BbLayerDev {
    const uint16_t *ln1, *wqkv, *qnorm, *knorm, *wo, *ln2, *wgu, *wdown;
    void *kc, *vc;
    float *ks, *vs;                     // int8 KV: per-(token, head) scales
    float k_scale, v_scale;             // fp8 KV: per-layer scalars
};

struct BbAllArgs {
    const BbLayerDev* layers; int L;
    PAArgs pa;                          // the attention stage's launch-invariant fields (per-layer ones come from the table)
    uint16_t* attn;                     // fragment-major [64][2048]
    uint16_t* resid; float* part;       // fragment-major residual stream [64][2048] + sum(r^2) slabs [128][64]
    uint16_t* act;                      // fragment-major [64][6144]
    uint16_t* qkv;                      // row-major [B][4096]
    int B, nap; float eps;
    uint32_t* flags; int32_t* err;
    unsigned long long* stamps; int stamp_layer;
};

// A layer's table entry as wave-uniform values.  The table lives in global memory and the kernel stores to global memory, so hipcc
// loads the entry with vector loads and -- not knowing that every lane read the same bytes -- wraps each buffer access made through
// one of its pointers in a waterfall loop (hundreds of them in this kernel).  readfirstlane pins every field to SGPRs.
template <typename T>
__device__ __forceinline__ T* bba_uniform(T* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ BbLayerDev bba_layer(const BbLayerDev* table, int l) {
    const BbLayerDev t = table[l];
    BbLayerDev u;
    u.ln1 = bba_uniform(t.ln1); u.wqkv = bba_uniform(t.wqkv); u.qnorm = bba_uniform(t.qnorm); u.knorm = bba_uniform(t.knorm);
    u.wo = bba_uniform(t.wo); u.ln2 = bba_uniform(t.ln2); u.wgu = bba_uniform(t.wgu); u.wdown = bba_uniform(t.wdown);
    u.kc = bba_uniform(t.kc); u.vc = bba_uniform(t.vc); u.ks = bba_uniform(t.ks); u.vs = bba_uniform(t.vs);
    u.k_scale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(t.k_scale)));
    u.v_scale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(t.v_scale)));
    return u;
}

// The attention stage's work split: a workgroup takes two (row, kv head) pairs of the SAME kv head, four waves each.  Which two
// rows: the batch's rows sorted by context length, shortest paired with longest (rank i with rank B - 1 - i) -- every
// workgroup then streams about the same number of K / V bytes, and the stage ends when the average pair does, not when the two
// longest rows that happened to share a workgroup do.  (Which workgroup computes a pair does not change its bits.)
__device__ __forceinline__ void bba_pairs(const int32_t* seq_lens, int B, int i, int* ord /* LDS [64] */, int& rowA, int& rowB) {
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        const int len = l < B ? seq_lens[l] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < 64; ++j) {
            const int lj = __builtin_amdgcn_readlane(len, j);
            rank += (lj < len || (lj == len && j < l)) ? 1 : 0;
        }
        ord[rank] = l;                  // rows >= B rank behind every live row
    }
    __syncthreads();
    const int nhalf = (B + 1) / 2, jb = B - 1 - i;
    rowA = i < nhalf ? ord[i] : -1;
    rowB = (i < nhalf && jb > i) ? ord[jb] : -1;
    __syncthreads();
}

template <int KV, bool BAL, bool PRE>
__global__ __launch_bounds__(CH_THREADS) void bb_all_kernel(const BbAllArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    ChainGate g;
    chain_gate_init(g, a.flags, a.err);
    g.dom = 8;                          // gate_up's 64-row tiles and the attention's (row, head) pairs tie every row group together
    g.nap = a.nap;
    const int wg = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int H = 2048, I = 6144, NQ = 4096;
    // ---- this workgroup's two attention pairs (fixed for the launch) and the block ids of their first K / V batch
    const int half = wave >> 2, w4 = wave & 3;
    const int kvh = wg % a.pa.kv_heads;
    int prow;
    if (BAL) {
        int rowA, rowB;
        bba_pairs(a.pa.seq_lens, a.B, wg / a.pa.kv_heads, reinterpret_cast<int*>(lds), rowA, rowB);
        prow = __builtin_amdgcn_readfirstlane(half ? rowB : rowA);
    } else {
        const int pair = wg + OMNI_CHAIN_WGS * half;       // rows 32 apart
        prow = pair < a.B * a.pa.kv_heads ? pair / a.pa.kv_heads : -1;
    }
    const bool active = prow >= 0;
    int blk[PA_U];
    PaPre<KV> pre;
    PAArgs pl = a.pa;                   // the attention arguments of the layer whose qkv stage runs / ran last
    if (PRE) pa_pre_blocks(a.pa, active ? prow : 0, w4, blk);
    // issued behind a qkv stage's epilogue stores (chain_gemm's prefetch hook): every wave issues exactly PA_PRE_LOADS loads --
    // a wave without a pair re-reads row 0's (the flag's store drain counts them)
    auto pf_ = [&]() { pa_pre_issue<KV>(pl, kvh, w4, blk, pre); };
    ChainPrefetch<PRE ? PA_PRE_LOADS(KV) : 0, decltype(pf_)> pf{pf_};
    auto set_layer = [&](const BbLayerDev& Ly, int l) {
        pl.qnorm_w = Ly.qnorm; pl.knorm_w = Ly.knorm;
        pl.k_cache = Ly.kc; pl.v_cache = Ly.vc; pl.k_scales = Ly.ks; pl.v_scales = Ly.vs;
        pl.k_scale = Ly.k_scale; pl.v_scale = Ly.v_scale;
        pl.slot_out = l == 0 ? a.pa.slot_out : nullptr;
    };
    // error-word stage codes: 0x20000 | layer << 8 | stage (1 qkv, 2 attention, 3 o_proj, 4 gate_up, 5 down_proj)
    {
        const BbLayerDev Ly = bba_layer(a.layers, 0);
        set_layer(Ly, 0);
        if (PRE)
            chain_gemm<2, 2, 8, 3, OMNI_EPI_BF16, 4, 0>(Ly.wqkv, Ly.ln1, a.resid, a.part, 1, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg & 127, wg >> 7, lds, g,
                                                        false, 0x20001, a.stamp_layer == 0 ? a.stamps : nullptr, nullptr, pf);
        else
            chain_gemm<2, 2, 8, 3, OMNI_EPI_BF16, 4>(Ly.wqkv, Ly.ln1, a.resid, a.part, 1, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg & 127, wg >> 7, lds, g,
                                                     false, 0x20001, a.stamp_layer == 0 ? a.stamps : nullptr);
    }
    for (int l = 0; l < a.L; ++l) {
        const BbLayerDev Ly = bba_layer(a.layers, l);
        unsigned long long* st = a.stamp_layer == l ? a.stamps : nullptr;
        const int lc = 0x20000 | (l << 8);
        g.stamps = st;
        pa_decode_body<KV, 2, true, true>(pl, lds + half * BBA_PAIR_FLOATS, kvh, prow, 0, w4, threadIdx.x & 255, active, &g, lc | 2,
                                          PRE ? &pre : nullptr);
        chain_gemm<2, 1, 8, 0, OMNI_EPI_RESID, 0>(Ly.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, lds, g,
                                                  true, lc | 3, st);
        chain_gemm<4, 3, 8, 3, OMNI_EPI_SILU_MUL_GU8, 2, 1, ChainNoPrefetch, true>(Ly.wgu, Ly.ln2, a.resid, a.part, H / 16, a.act, 0, nullptr, a.B, I, a.eps,
                                                                                   wg, 0, lds, g, true, lc | 4, st);
        chain_gemm<2, 1, 24, 0, OMNI_EPI_RESID, 4>(Ly.wdown, nullptr, a.act, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, lds, g,
                                                   true, lc | 5, st);
        if (l + 1 < a.L) {
            const BbLayerDev Ln = bba_layer(a.layers, l + 1);
            set_layer(Ln, l + 1);
            if (PRE)
                chain_gemm<2, 2, 8, 3, OMNI_EPI_BF16, 4, 0>(Ln.wqkv, Ln.ln1, a.resid, a.part, H / 16, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg & 127, wg >> 7,
                                                            lds, g, true, lc | 1, a.stamp_layer == l + 1 ? a.stamps : nullptr, nullptr, pf);
            else
                chain_gemm<2, 2, 8, 3, OMNI_EPI_BF16, 4>(Ln.wqkv, Ln.ln1, a.resid, a.part, H / 16, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg & 127, wg >> 7,
                                                         lds, g, true, lc | 1, a.stamp_layer == l + 1 ? a.stamps : nullptr);
        }
    }
}

OMNI_KNOB g_bb_all = 0, g_bb_all_nap = 1;      // off: measured +0.09 .. +0.12 ms per step against round 3's two launches per layer (profiles/r04_ab_table.txt)
#ifdef OMNI_DEBUG_HOOKS
static int g_bb_all_bal = 1, g_bb_all_pre = 1;
#endif
#ifdef OMNI_DEBUG_HOOKS
static unsigned long long* g_bba_stamps = nullptr;
static int g_bba_stamp_layer = -1;
extern "C" void omni_debug_bb_all(int on) { g_bb_all = on; }
extern "C" void omni_debug_bb_all_mode(int bal, int pre) { g_bb_all_bal = bal; g_bb_all_pre = pre; }      // A/B: balanced pairs, K / V batch 0 ahead of the stage
extern "C" void omni_debug_bb_all_stamps(void* buf, int layer) { g_bba_stamps = (unsigned long long*)buf; g_bba_stamp_layer = layer; }
#endif

size_t k_bb_all_table_bytes(int layers) { return (size_t)layers * sizeof(BbLayerDev); }

// fill the device table (engine creation); scales: per-layer fp8 scalars (NULL: desc.k_scale / v_scale for every layer)
int k_bb_all_table(void* table_dev, const omni_talker_desc& d, const omni_layer_weights* layers, void* const* k_cache, void* const* v_cache,
                   float* const* k_scales, float* const* v_scales) {
    std::vector<BbLayerDev> h(d.layers);
    for (int l = 0; l < d.layers; ++l) {
        const omni_layer_weights& w = layers[l];
        h[l] = BbLayerDev{(const uint16_t*)w.ln1, (const uint16_t*)w.wqkv, (const uint16_t*)w.qnorm, (const uint16_t*)w.knorm, (const uint16_t*)w.wo,
                          (const uint16_t*)w.ln2, (const uint16_t*)w.wgu, (const uint16_t*)w.wdown, k_cache[l], v_cache[l],
                          k_scales ? k_scales[l] : nullptr, v_scales ? v_scales[l] : nullptr, d.k_scale, d.v_scale};
    }
    if (d.layers > 0 && hipMemcpy(table_dev, h.data(), h.size() * sizeof(BbLayerDev), hipMemcpyHostToDevice) != hipSuccess) {
        omni_set_error("bb_all: layer table upload failed");
        return OMNI_EHIP;
    }
    return OMNI_OK;
}

// per-layer fp8 scales into the table (calibration): k[l], v[l] host arrays
int k_bb_all_set_scales(void* table_dev, int layers, const float* k, const float* v, void* stream) {
    for (int l = 0; l < layers; ++l) {
        const float kv[2] = {k[l], v[l]};
        char* dst = reinterpret_cast<char*>(table_dev) + (size_t)l * sizeof(BbLayerDev) + offsetof(BbLayerDev, k_scale);
        if (hipMemcpyAsync(dst, kv, sizeof(kv), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
            omni_set_error("bb_all: scale upload failed");
            return OMNI_EHIP;
        }
    }
    return OMNI_OK;
}

bool k_bb_all_supported(const omni_talker_desc& d, int B, bool has_ar) {
    return g_bb_all && !has_ar && k_bb_chain_supported(d, B) && d.hidden == 2048 && d.kv_heads == 8 && d.layers >= 1 &&
           (d.kv_dtype == OMNI_KV_FP8 || d.kv_dtype == OMNI_KV_BF16 || d.kv_dtype == OMNI_KV_FP16 || d.kv_dtype == OMNI_KV_INT8);
}

int k_bb_all(const omni_talker_desc& d, const void* table_dev, const omni_step_io* io, void* attn, void* resid, float* part, void* act, void* qkv,
             uint32_t* flags, int32_t* err, void* stream) {
    BbAllArgs a{};
    a.layers = reinterpret_cast<const BbLayerDev*>(table_dev); a.L = d.layers;
    PAArgs& p = a.pa;
    p.qkv = (const uint16_t*)qkv; p.positions = io->positions; p.rope_delta = io->rope_delta; p.cos_sin = (const uint16_t*)d.cos_sin;
    p.slot_out = io->slot_mapping; p.eps = d.eps;
    p.block_table = io->block_table; p.bt_stride = d.bt_stride; p.seq_lens = io->seq_lens; p.out = (uint16_t*)attn;
    p.q_heads = d.q_heads; p.kv_heads = d.kv_heads; p.bs = d.block_size; p.bs_shift = __builtin_ctz((unsigned)d.block_size);
    p.k_scale = d.k_scale; p.v_scale = d.v_scale; p.sm_scale = 1.0f / sqrtf((float)d.head_dim);
    p.nsplit = 1; p.out_frag = 1; p.kv_rep = 1; p.dense_pos = -1; p.num_live = io->num_live;
    p.rope_rows = io->rope_delta ? (d.rope_rows > 0 ? d.rope_rows : d.max_model_len) : 0;
    a.attn = (uint16_t*)attn; a.resid = (uint16_t*)resid; a.part = part; a.act = (uint16_t*)act; a.qkv = (uint16_t*)qkv;
    a.B = io->B; a.nap = g_bb_all_nap; a.eps = d.eps; a.flags = flags; a.err = err;
    a.stamp_layer = -1;
#ifdef OMNI_DEBUG_HOOKS
    a.stamps = g_bba_stamps; a.stamp_layer = g_bba_stamp_layer;
#endif
#ifdef OMNI_DEBUG_HOOKS
#define BBA_LAUNCH(KV)                                                     \
    do {                                                                   \
        if (g_bb_all_bal && g_bb_all_pre) BBA_LAUNCH_(KV, true, true);     \
        else if (g_bb_all_bal) BBA_LAUNCH_(KV, true, false);               \
        else if (g_bb_all_pre) BBA_LAUNCH_(KV, false, true);               \
        else BBA_LAUNCH_(KV, false, false);                                \
    } while (0)
#else
#define BBA_LAUNCH(KV) BBA_LAUNCH_(KV, true, true)
#endif
#define BBA_LAUNCH_(KV, BAL, PRE)                                                                                                          \
    do {                                                                                                                                   \
        static bool attr_ = false;                                                                                                         \
        if (!attr_) {                                                                                                                      \
            (void)hipFuncSetAttribute((const void*)bb_all_kernel<KV, BAL, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, BBA_LDS_BYTES); \
            attr_ = true;                                                                                                                  \
        }                                                                                                                                  \
        hipLaunchKernelGGL((bb_all_kernel<KV, BAL, PRE>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), BBA_LDS_BYTES, (hipStream_t)stream, a);   \
    } while (0)
    switch (d.kv_dtype) {
        case OMNI_KV_FP8: BBA_LAUNCH(OMNI_KV_FP8); break;
        case OMNI_KV_BF16: BBA_LAUNCH(OMNI_KV_BF16); break;
        case OMNI_KV_FP16: BBA_LAUNCH(OMNI_KV_FP16); break;
        default: BBA_LAUNCH(OMNI_KV_INT8); break;
    }
#undef BBA_LAUNCH_
#undef BBA_LAUNCH
    OMNI_CHECK_LAUNCH("bb_all");
    return OMNI_OK;
}
