// The talker backbone between two attention launches as ONE persistent launch per decoder layer:
//     o_proj(l) -> gate_up(l) -> down_proj(l) -> qkv_proj(l + 1)
// (reference: the vLLM Qwen3 decoder layer reached through qwen3_tts_talker.py:341,414-422; four of the five launches of
// a layer on the launch-per-op path).  Same protocol as the code-predictor chain (cp_chain.hip, chain_gemm.cuh): a stage's
// weight slice -- 64-196 KB per workgroup, 32-96 VGPRs per lane -- goes out before the stage's flags, so the HBM stream keeps
// running through the hand-off instead of stopping for a kernel boundary (1.65 us) plus a cold first fetch; activations
// cross behind the flags with sc1 accesses.  Tiles = the launch path's (pick_tile at 64 rows), so results are bit-identical.
// Shapes (round 6: parameters, not literals -- the checkpoint decides them, configuration_qwen3_tts.py:192-216): the stage set is a template
// on the k-steps per wave of its three reduction widths -- KO = q_heads * 128 / 256 (o_proj), KH = hidden / 256 (gate_up, qkv), KI =
// intermediate / 256 (down_proj) -- and reads the output widths (hidden, intermediate, qkv) from its arguments; BB_SHAPES lists the
// instantiated triples: the released 1.7B shape (8, 8, 24) and (6, 6, 18) = hidden 1536 / intermediate 4608 / 12 q heads, a shape no released
// checkpoint has (tests/test_gpu_chain.py).  A shape runs here when its triple is listed and every stage's tile grid fits the 256
// workgroups (k_bb_chain_supported); since the rstd moved into the epilogues (PRO 3) and the slab order is canonical (gemm_frag.cuh) the
// bits no longer depend on which row tiles the launch path picks for the same GEMM.  33-64 rows: bb_chain_kernel (33-48 rows run the 64-row
// stage set with their last row tile partly filled), 1-32 rows: bb_chain_b32_kernel, the 0.6B shape (hidden 1024 = the code predictor's
// layer dimensions, 16-row tiles): bb_chain_small_kernel; every other configuration stays on the launch path.
#include "chain_gemm.cuh"
#include "common.cuh"
#include "kernels.h"

#define BB_LDS_BYTES ((CH_WAVES * 12 * 64 * 16) + CH_WAVES * 64 * 4)      // combine slots (gate_up: 12 tiles, ONE pass since round 4) + rstd area

struct BbArgs {
    const uint16_t *wo, *ln2, *wgu, *wdown, *ln1_next, *wqkv_next;      // *_next == NULL: last layer, no qkv stage
    int pf;                             // cross-stage weight prefetch (0: every stage fetches its own slice at its entry)
    const uint16_t* attn;               // fragment-major [64][2048]: the attention launch's output
    uint16_t* resid; float* part;       // fragment-major residual stream [64][2048] + sum(r^2) slabs [128][64]
    uint16_t* act;                      // fragment-major [64][6144]
    uint16_t* qkv;                      // row-major [B][NQ]
    int H, I, NQ;                       // output widths: hidden, intermediate, (q_heads + 2 kv_heads) * 128
    // the LAST layer's launch (no next qkv) may end in the talker's own head instead (round 6): final norm folded into the lm_head GEMM, its
    // normalised rows = h[t + 1] (row-major, rows below *num_live), logits fp32 [B][V] holding bf16-rounded values, the codec mask applied
    const uint16_t *lm_head, *final_norm; float* logits; uint16_t* last_hidden; const uint8_t* mask; const int32_t* num_live; float mask_fill; int V;
    int B, nap; float eps;
    uint32_t* flags; int32_t* err;
    unsigned long long* stamps;
    // tensor-parallel rank (round 6): o_proj and down_proj are row-parallel partials; their all-reduce rides in the two residual epilogues
    // (chain_gemm AR).  ar_epoch = allreduce.hip's epoch / ticket words: this launch makes calls epoch + 1 (o_proj) and epoch + 2 (down_proj)
    ChainAr ar_o, ar_dn;
    uint32_t* ar_epoch;
};

// (Round 3 also tried a ninth wave that warms the XCD's L2 with the next stage's slice -- one dword per 64 bytes, nothing
// depending on it: +0.10 ms per step, removed; and the loader / consumer split of bb_engine.hip.  What is kept is below:
// the NEXT stage's weight loads issued by the compute waves themselves, behind their last activation load.)
// GU_G / DN_G / QK_G: activation-ring depth (k-steps per wave in flight behind the flags) of gate_up (its weights ride in the same
// ring) / down_proj / the next qkv; PF: the cross-stage weight prefetch arm
// GU1P: gate_up's 12 tiles combine in one pass (chain_gemm ONEPASS; round 4: epilogue 2.4 -> 1.0 us); GUW0: gate_up's whole weight
// slice (96 registers) goes out at stage entry, ahead of the flags, instead of riding in the activation ring (A/B arm)
// DEFER: the RMSNorm's rstd applied in the epilogue (chain_gemm PRO 3) -- round 4's A/B arm of VERDICT r3 item 1b; the product form since
// round 6 (gemm_skinny_kernel takes it for the same GEMMs); false = round 5's exact-rstd stages, debug library only
// WNT: bit 0 = non-temporal weight loads in gate_up (every byte read by exactly ONE workgroup), bit 1 = in o_proj / down_proj / qkv (each slice
// read by the two workgroups of a column tile's row halves) -- round 5 A/B arm (MI355X_MICROARCH "nt-weights")
// HEAD: the instantiation the LAST layer launches when the step's head rides along (its own code object entry: the other 27 launches keep the
// leaner kernel -- with the head stage compiled into every launch the segment cost 0.9 us more per layer, register allocation of the shared stages)
// AR: the instantiation of a tensor-parallel rank (the all-reduces inside the o_proj / down_proj stages); HALF: 128 workgroups play the 256 of the
// stage grid, two VIRTUAL workgroups each, stage by stage (a virtual workgroup's stage needs only stages that every physical workgroup has
// already run or will run before it waits again: no cycle) -- two such launches of two processes fit the chip side by side, which is how the
// cross-process exchange is tested on one GPU (tests/test_gpu_tp.py) and how a co-located second engine keeps its backbone chain
template <int KO, int KH, int KI, int GU_G, int DN_G, int QK_G, bool PF, bool GU1P = true, bool GUW0 = false, bool DEFER = true, int WNT = 0, bool HEAD = false,
          bool AR = false, bool HALF = false>
__global__ __launch_bounds__(CH_THREADS) void bb_chain_kernel(const BbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(!PF || (!AR && !HALF), "the cross-stage prefetch arm is a single-rank, full-grid experiment");
    uint32_t e0 = 0;
    if constexpr (AR) e0 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.ar_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    ChainGate g;
    chain_gate_init(g, a.flags, a.err);
    g.dom = 8;                          // gate_up's 64-row tiles tie every row group together
    g.nap = a.nap;
    const int wg = blockIdx.x;
    const int H = a.H, I = a.I, NQ = a.NQ;
    // workgroup -> tile: o_proj / down_proj 16 columns x 32 rows, gate_up 24 activation columns x 64 rows, qkv 32 columns x 32 rows; a
    // workgroup past a stage's grid (column tiles x row tiles < 256 at widths below the 1.7B shape's) publishes the stage and runs ahead
    const int nc_h = H >> 4, nc_gu = I / 24, nc_q = NQ >> 5;
    // stage codes (error word): 0x1001 .. 0x1004.  Weight registers of the stage AFTER the current one are filled by the current
    // stage (chain_gemm WSRC 3): gate_up's 96 registers during o_proj, down_proj's 96 during gate_up (the allocator reuses
    // gate_up's as its k-steps retire), the next qkv's 64 during down_proj.
    if constexpr (PF) {
        static_assert(!PF || (KO == 8 && KH == 8 && KI == 24), "the cross-stage prefetch arm (debug library) is built for the 1.7B shape");
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const uint32_t lane16 = lane * 16;
        u32x4 Wg[8][3], Wd[24][1], Wk[8][2];
        auto pf_g_ = [&]() {
            const coh_rsrc_t rs = coh_rsrc(a.wgu);
#pragma unroll
            for (int d = 0; d < 8; ++d)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    Wg[d][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (uint32_t)(((wg * 3 + j) * 64 + wave + d * CH_WAVES) * 1024), 0);
        };
        auto pf_d_ = [&]() {
            const coh_rsrc_t rs = coh_rsrc(a.wdown);
#pragma unroll
            for (int d = 0; d < 24; ++d) Wd[d][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (uint32_t)(((wg & 127) * 192 + wave + d * CH_WAVES) * 1024), 0);
        };
        auto pf_k_ = [&]() {
            if (a.wqkv_next == nullptr) return;
            const coh_rsrc_t rs = coh_rsrc(a.wqkv_next);
#pragma unroll
            for (int d = 0; d < 8; ++d)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    Wk[d][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (uint32_t)((((wg & 127) * 2 + j) * 64 + wave + d * CH_WAVES) * 1024), 0);
        };
        ChainPrefetch<24, decltype(pf_g_)> pf_g{pf_g_};
        ChainPrefetch<24, decltype(pf_d_)> pf_d{pf_d_};
        ChainPrefetch<16, decltype(pf_k_)> pf_k{pf_k_};
        chain_gemm<2, 1, 8, 0, OMNI_EPI_RESID, 0, 0>(a.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, lds, g,
                                                     false, 0x1001, a.stamps, nullptr, pf_g);
        chain_gemm<4, 3, 8, 3, OMNI_EPI_SILU_MUL_GU8, 2, 3>(a.wgu, a.ln2, a.resid, a.part, H / 16, a.act, 0, nullptr, a.B, I, a.eps, wg, 0, lds, g,
                                                            true, 0x1002, a.stamps, Wg, pf_d);
        chain_gemm<2, 1, 24, 0, OMNI_EPI_RESID, 4, 3>(a.wdown, nullptr, a.act, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, lds,
                                                      g, true, 0x1003, a.stamps, Wd, pf_k);
        if (a.wqkv_next)
            chain_gemm<2, 2, 8, 3, OMNI_EPI_BF16, 4, 3>(a.wqkv_next, a.ln1_next, a.resid, a.part, H / 16, a.qkv, NQ, nullptr, a.B, NQ, a.eps,
                                                        wg & 127, wg >> 7, lds, g, true, 0x1004, a.stamps, Wk);
        return;
    } else {
    constexpr int NT1 = (WNT & 1) ? OMNI_AUX_NT : 0, NT2 = (WNT & 2) ? OMNI_AUX_NT : 0;
    // the four stages of (virtual) workgroup WG_ behind gate G_
#define BB_O(WG_, G_)                                                                                                                                   \
    chain_gemm<2, 1, KO, 0, OMNI_EPI_RESID, 0, 0, ChainNoPrefetch, false, 64, NT2, AR>(a.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps,    \
                                                                                       (WG_) % nc_h, (WG_) / nc_h, lds, G_, false, 0x1001, a.stamps, nullptr,  \
                                                                                       ChainNoPrefetch(), nullptr, nullptr, 0.f, nullptr, &a.ar_o, e0 + 1u)
#define BB_GU(WG_, G_)                                                                                                                                  \
    do {                                                                                                                                                \
        if ((WG_) < nc_gu)                                                                                                                              \
            chain_gemm<4, 3, KH, DEFER ? 3 : 2, OMNI_EPI_SILU_MUL_GU8, GU_G, GUW0 ? 0 : 1, ChainNoPrefetch, GU1P, 64, NT1>(                              \
                a.wgu, a.ln2, a.resid, a.part, H / 16, a.act, 0, nullptr, a.B, I, a.eps, WG_, 0, lds, G_, true, 0x1002, a.stamps);                        \
        else                                                                                                                                            \
            chain_gate_skip(G_);                                                                                                                        \
    } while (0)
#define BB_DN(WG_, G_)                                                                                                                                  \
    chain_gemm<2, 1, KI, 0, OMNI_EPI_RESID, DN_G, 0, ChainNoPrefetch, false, 64, NT2, AR>(a.wdown, nullptr, a.act, nullptr, 0, a.resid, 0, a.part, a.B, H,       \
                                                                                          a.eps, (WG_) % nc_h, (WG_) / nc_h, lds, G_, true, 0x1003, a.stamps,   \
                                                                                          nullptr, ChainNoPrefetch(), nullptr, nullptr, 0.f, nullptr, &a.ar_dn,  \
                                                                                          e0 + 2u)
#define BB_QK(WG_, G_)                                                                                                                                  \
    chain_gemm<2, 2, KH, DEFER ? 3 : 2, OMNI_EPI_BF16, QK_G, 0, ChainNoPrefetch, false, 64, NT2>(a.wqkv_next, a.ln1_next, a.resid, a.part, H / 16, a.qkv, NQ,    \
                                                                                                 nullptr, a.B, NQ, a.eps, (WG_) % nc_q, (WG_) / nc_q, lds, G_,   \
                                                                                                 true, 0x1004, a.stamps)
    if constexpr (HALF) {
        static_assert(!HALF || !HEAD, "the half grid has no head stage (a tensor-parallel rank's lm_head stays its own launch)");
        ChainGate g2;
        chain_gate_init(g2, a.flags, a.err, wg + OMNI_CHAIN_WGS / 2);
        g2.dom = 8;
        g2.nap = a.nap;
        const int wg2 = wg + OMNI_CHAIN_WGS / 2;
        BB_O(wg, g); BB_O(wg2, g2);
        BB_GU(wg, g); BB_GU(wg2, g2);
        BB_DN(wg, g); BB_DN(wg2, g2);
        if (a.wqkv_next) { BB_QK(wg, g); BB_QK(wg2, g2); }
    } else {
    BB_O(wg, g);
    BB_GU(wg, g);
    BB_DN(wg, g);
    if (a.wqkv_next)
        BB_QK(wg, g);
    else if (HEAD && a.lm_head) {
        // the step's head as the last stage of its last backbone launch (one launch and one cold start less: 9.8 us + a boundary -> a stage):
        // gemm_skinny_kernel<2, 2, PRO 2, F32_BF16RND>'s tile and arithmetic (exact norm: its normalised rows are h[t + 1])
        const int nc_v = a.V >> 5;
        chain_gemm<2, 2, KH, 2, OMNI_EPI_F32_BF16RND, QK_G, 0, ChainNoPrefetch, false, 64, NT2>(a.lm_head, a.final_norm, a.resid, a.part, H / 16, a.logits, a.V,
                                                                                                nullptr, a.B, a.V, a.eps, wg % nc_v, wg / nc_v, lds, g, true, 0x1005,
                                                                                                a.stamps, nullptr, ChainNoPrefetch(), a.last_hidden, a.mask,
                                                                                                a.mask_fill, a.num_live);
    }
    }
#undef BB_O
#undef BB_GU
#undef BB_DN
#undef BB_QK
    if constexpr (AR) {
        // the last workgroup out closes the launch's two all-reduce calls (allreduce.hip's epoch word: every workgroup read it at its start)
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t t = atomicAdd(a.ar_epoch + 1, 1u);
            if (t == gridDim.x - 1) {
                __hip_atomic_store(a.ar_epoch + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.ar_epoch, e0 + 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    }
}

// ---- the 0.6B backbone shape (hidden 1024, 16 x 128 attention width over 8 kv heads, intermediate 3072: BASELINE config #2) --
// the dimensions of the code predictor's layers, so the segment runs on the stage set of cp_chain.hip: 16-row tiles, four
// independent row groups of 64 workgroups (gate_up on 32-row tiles above 32 rows, as the launch path picks them: same bits), any
// batch size 1..64; a workgroup without rows in any stage publishes the launch's stage count and leaves.  Launch-per-op this
// layer is five launches of 7-11 us for 25 MB of weights (profiles/r03_other_configs.txt: 56 us per layer).
#define BBS_LDS_FLOATS ((CH_WAVES * 6 * 4 * 64) + CH_WAVES * 64)
template <bool GU_NARROW>
__global__ __launch_bounds__(CH_THREADS) void bb_chain_small_kernel(const BbArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[BBS_LDS_FLOATS];
    ChainGate g;
    chain_gate_init(g, a.flags, a.err);
    g.dom = GU_NARROW ? 7 : 6;          // 32-row gate_up tiles tie two 16-row groups together
    g.nap = a.nap;
    const int wg = blockIdx.x;
    constexpr int H = 1024, I = 3072, NQ = 4096, KO = 2048;
    if (!((wg >> 6) * 16 < a.B || (GU_NARROW && (wg >> 7) * 32 < a.B))) {
        if (threadIdx.x < 64) chain_flag_publish(g.frs, wg, g.epoch + (a.wqkv_next ? 4u : 3u));
        return;
    }
    static_assert(KO == 8 * 256 && H == 4 * 256 && I == 12 * 256, "bb_chain_small: k-steps per wave");
    chain_gemm<1, 1, 8, 0, OMNI_EPI_RESID>(a.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 63, wg >> 6, lds, g, false, 0x1001,
                                           a.stamps);
    if (GU_NARROW)
        chain_gemm<2, 3, 4, 3, OMNI_EPI_SILU_MUL_GU8>(a.wgu, a.ln2, a.resid, a.part, H / 16, a.act, 0, nullptr, a.B, I, a.eps, wg & 127, wg >> 7, lds, g,
                                                      true, 0x1002, a.stamps);
    else
        chain_gemm<1, 6, 4, 3, OMNI_EPI_SILU_MUL_GU8>(a.wgu, a.ln2, a.resid, a.part, H / 16, a.act, 0, nullptr, a.B, I, a.eps, wg & 63, wg >> 6, lds, g,
                                                      true, 0x1002, a.stamps);
    chain_gemm<1, 1, 12, 0, OMNI_EPI_RESID>(a.wdown, nullptr, a.act, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 63, wg >> 6, lds, g, true, 0x1003,
                                            a.stamps);
    if (a.wqkv_next)
        chain_gemm<1, 4, 4, 3, OMNI_EPI_BF16>(a.wqkv_next, a.ln1_next, a.resid, a.part, H / 16, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg & 63, wg >> 6, lds, g,
                                              true, 0x1004, a.stamps);
}

// ---- the 1.7B shape at 1-32 rows (round 4): the launch path's tiles at those batch sizes -- 16-row tiles for qkv / o_proj / down_proj
// (128 column tiles x 2 row tiles), gate_up on 16 rows x 24 columns up to 16 rows and 32 x 24 above (the rstd summation order follows the
// rows per tile: same bits as the launch path).  33-48 rows: bb_chain_kernel (the 64-row stage set).
template <int KO, int KH, int KI, int GU_MT, bool HEAD = false>
__global__ __launch_bounds__(CH_THREADS) void bb_chain_b32_kernel(const BbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    ChainGate g;
    chain_gate_init(g, a.flags, a.err);
    g.dom = 8;
    g.nap = a.nap;
    const int wg = blockIdx.x;
    const int H = a.H, I = a.I, NQ = a.NQ;
    const int nc_h = H >> 4, nc_gu = I / 24, nc_q = NQ >> 5;        // 16-row tiles: two row tiles of each column tile cover the 32 rows
    chain_gemm<1, 1, KO, 0, OMNI_EPI_RESID, 0>(a.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg % nc_h, wg / nc_h, lds, g, false, 0x1001,
                                               a.stamps);
    if (wg < nc_gu)
        chain_gemm<GU_MT, 3, KH, 3, OMNI_EPI_SILU_MUL_GU8, 2, 1>(a.wgu, a.ln2, a.resid, a.part, H / 16, a.act, 0, nullptr, a.B, I, a.eps, wg, 0, lds, g, true, 0x1002,
                                                                 a.stamps);
    else
        chain_gate_skip(g);
    chain_gemm<1, 1, KI, 0, OMNI_EPI_RESID, 4>(a.wdown, nullptr, a.act, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg % nc_h, wg / nc_h, lds, g, true, 0x1003,
                                               a.stamps);
    if (a.wqkv_next)
        chain_gemm<1, 2, KH, 3, OMNI_EPI_BF16, 4>(a.wqkv_next, a.ln1_next, a.resid, a.part, H / 16, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg % nc_q, wg / nc_q, lds, g,
                                                  true, 0x1004, a.stamps);
    else if (HEAD && a.lm_head) {
        const int nc_v = a.V >> 5;
        chain_gemm<1, 2, KH, 2, OMNI_EPI_F32_BF16RND, 4>(a.lm_head, a.final_norm, a.resid, a.part, H / 16, a.logits, a.V, nullptr, a.B, a.V, a.eps, wg % nc_v,
                                                         wg / nc_v, lds, g, true, 0x1005, a.stamps, nullptr, ChainNoPrefetch(), a.last_hidden, a.mask,
                                                         a.mask_fill, a.num_live);
    }
}

OMNI_KNOB g_bb_chain = 1, g_bb_nap = 1, g_bb_prefetch = 0, g_bb_deep = 0, g_bb_min_rows = 33, g_bb_b32 = 1, g_bb_ar = 1;      // cross-stage prefetch: measured +0.4 ms per step, off
#ifdef OMNI_DEBUG_HOOKS
static unsigned long long* g_bb_stamps = nullptr;
extern "C" void omni_debug_bb_chain(int on) { g_bb_chain = on != 0; g_bb_prefetch = on == 2; }      // 2: with the cross-stage weight prefetch
extern "C" void omni_debug_bb_stamps(void* buf) { g_bb_stamps = (unsigned long long*)buf; }
extern "C" void omni_debug_bb_deep(int mode) { g_bb_deep = mode; }
extern "C" void omni_debug_bb_ar(int on) { g_bb_ar = on; }                  // 0: tensor-parallel ranks keep the launch-per-op backbone (round 5)
extern "C" void omni_debug_bb_min_rows(int rows) { g_bb_min_rows = rows; g_bb_b32 = rows <= 49; }                          // smallest batch the backbone chain takes                                  // deeper activation / weight rings
#endif

// the instantiated shape triples (KO, KH, KI) = (q_heads * 128, hidden, intermediate) / 256: the released 1.7B shape and one no released
// checkpoint has (hidden 1536, intermediate 4608, 12 q heads: tests/test_gpu_chain.py, tests/test_gpu_engine.py); a new width = one more line
#define BB_SHAPES(X) X(8, 8, 24) X(6, 6, 18)
// ... and the triples of a tensor-parallel RANK (the all-reduce instantiations, full and half grid): the 1.7B shape whole (a one-rank group:
// bench.py --tp-force), over 2 ranks (8 q / 4 kv heads, intermediate 3072), over 4 and over the 8 GPUs of a node (2 q / 1 kv head, 768)
#define BB_AR_SHAPES(X) X(8, 8, 24) X(4, 8, 12) X(2, 8, 6) X(1, 8, 3)

// ar: the rank's peer table (NULL = not tensor parallel); half: the 128-workgroup grid (omni_talker_set_chains(t, 2))
bool k_bb_chain_supported(const omni_talker_desc& d, int B, const omni_ar_peers* ar, bool half) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 0;
    }
    const bool has_ar = ar != nullptr;
    // a tensor-parallel rank: every peer's tile flags mapped (omni_ar_peers::tile_flags; a one-rank group exchanges nothing)
    bool ar_ok = has_ar && g_bb_ar;
    if (has_ar)
        for (int p = 0; p < ar->world; ++p) ar_ok = ar_ok && (ar->world == 1 || ar->tile_flags[p] != nullptr);
    const bool common = g_bb_chain && d.cp_chain && cus >= OMNI_CHAIN_WGS && d.fused_norm && d.frag_layout && (!has_ar || ar_ok) && d.moe_experts == 0 &&
                        d.head_dim == 128 && B <= 64;
    if (common && !has_ar && !half && k_bb_chain_small(d)) return B >= 1;      // 0.6B shape: 16-row tiles, every batch size
    const int H = d.hidden, I = d.inter, KOW = d.q_heads * 128, NQ = (d.q_heads + 2 * d.kv_heads) * 128;
    bool listed = false;
#define X(KO_, KH_, KI_) listed = listed || (KOW == KO_ * 256 && H == KH_ * 256 && I == KI_ * 256);
    if (has_ar) { BB_AR_SHAPES(X) }
    else if (half) { X(8, 8, 24) }
    else { BB_SHAPES(X) }
#undef X
    if (has_ar || half) return common && listed && I % 24 == 0 && NQ % 32 == 0 && B >= 33;      // the 64-row stage set only
    // every stage's tile grid inside the 256 workgroups: o_proj / down_proj 16 columns x 2 row tiles, gate_up 24 activation columns, qkv 32 x 2
    const bool grids = I % 24 == 0 && NQ % 32 == 0 && (H / 16) * 2 <= OMNI_CHAIN_WGS && I / 24 <= OMNI_CHAIN_WGS && (NQ / 32) * 2 <= OMNI_CHAIN_WGS;
    return common && listed && grids && (B >= g_bb_min_rows || (g_bb_b32 && B <= 32));
}
bool k_bb_chain_small(const omni_talker_desc& d) {
    return d.hidden == 1024 && d.inter == 3072 && d.q_heads * 128 == 2048 && (d.q_heads + 2 * d.kv_heads) * 128 == 4096;
}

// one shape's launches: the 64-row stage set (33-64 rows) or the 16-row one (1-32 rows); the debug library's A/B arms exist for the 1.7B triple
template <int KO, int KH, int KI>
static int bb_launch_shape(const BbArgs& a, int B, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)bb_chain_b32_kernel<KO, KH, KI, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)bb_chain_b32_kernel<KO, KH, KI, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)bb_chain_b32_kernel<KO, KH, KI, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)bb_chain_b32_kernel<KO, KH, KI, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)bb_chain_kernel<KO, KH, KI, 2, 4, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)bb_chain_kernel<KO, KH, KI, 2, 4, 4, false, true, false, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
        attr = true;
    }
    const bool head = a.lm_head != nullptr;
    if (B <= 32) {
        if (B <= 16) {
            if (head) hipLaunchKernelGGL((bb_chain_b32_kernel<KO, KH, KI, 1, true>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), BB_LDS_BYTES, stream, a);
            else hipLaunchKernelGGL((bb_chain_b32_kernel<KO, KH, KI, 1>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), BB_LDS_BYTES, stream, a);
        } else {
            if (head) hipLaunchKernelGGL((bb_chain_b32_kernel<KO, KH, KI, 2, true>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), BB_LDS_BYTES, stream, a);
            else hipLaunchKernelGGL((bb_chain_b32_kernel<KO, KH, KI, 2>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), BB_LDS_BYTES, stream, a);
        }
        OMNI_CHECK_LAUNCH("bb_chain_b32");
        return OMNI_OK;
    }
    if (head) hipLaunchKernelGGL((bb_chain_kernel<KO, KH, KI, 2, 4, 4, false, true, false, true, 0, true>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), BB_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((bb_chain_kernel<KO, KH, KI, 2, 4, 4, false>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), BB_LDS_BYTES, stream, a);
    OMNI_CHECK_LAUNCH("bb_chain");
    return OMNI_OK;
}

// the all-reduce / half-grid instantiations of one triple (64-row stage set)
template <int KO, int KH, int KI, bool AR>
static int bb_launch_ar(const BbArgs& a, bool half, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)bb_chain_kernel<KO, KH, KI, 2, 4, 4, false, true, false, true, 0, false, AR, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)bb_chain_kernel<KO, KH, KI, 2, 4, 4, false, true, false, true, 0, false, AR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
        if (AR) (void)hipFuncSetAttribute((const void*)bb_chain_kernel<KO, KH, KI, 2, 4, 4, false, true, false, true, 0, true, AR, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
        attr = true;
    }
    if (AR && !half && a.lm_head != nullptr) {      // a tensor-parallel rank's last layer: the (replicated) lm_head as the launch's last stage
        hipLaunchKernelGGL((bb_chain_kernel<KO, KH, KI, 2, 4, 4, false, true, false, true, 0, true, AR, false>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), BB_LDS_BYTES, stream, a);
        OMNI_CHECK_LAUNCH("bb_chain(all-reduce, head)");
        return OMNI_OK;
    }
    if (half) hipLaunchKernelGGL((bb_chain_kernel<KO, KH, KI, 2, 4, 4, false, true, false, true, 0, false, AR, true>), dim3(OMNI_CHAIN_WGS / 2), dim3(CH_THREADS), BB_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((bb_chain_kernel<KO, KH, KI, 2, 4, 4, false, true, false, true, 0, false, AR, false>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), BB_LDS_BYTES, stream, a);
    OMNI_CHECK_LAUNCH("bb_chain(all-reduce / half grid)");
    return OMNI_OK;
}

static void bb_fill_ar(ChainAr& c, const omni_ar_peers& p) {
    c.world = p.world; c.rank = p.rank;
    const ptrdiff_t eoff = p.error - reinterpret_cast<int32_t*>(p.flags[p.rank]);      // the error word sits at the same offset in every rank's control block
    for (int r = 0; r < CH_AR_MAX_WORLD; ++r) {
        c.data[r] = r < p.world ? (const uint16_t*)p.data[r] : nullptr;
        c.tflags[r] = r < p.world ? p.tile_flags[r] : nullptr;
        c.error[r] = r < p.world ? (p.flags[r] ? reinterpret_cast<int32_t*>(p.flags[r]) + eoff : p.error) : nullptr;
    }
}

bool k_bb_chain_head_supported(const omni_talker_desc& d) {      // the lm_head stage's tile grid: 32 columns x 2 row tiles inside 256 workgroups
    return !k_bb_chain_small(d) && d.vocab % 32 == 0 && (d.vocab / 32) * 2 <= OMNI_CHAIN_WGS && d.lm_head != nullptr &&
           !g_bb_prefetch && !g_bb_deep;      // (the debug library's A/B arms keep the head as its own launch)
}

int k_bb_chain(const omni_talker_desc& d, const omni_layer_weights& w, const omni_layer_weights* next, const void* attn, void* resid, float* part,
               void* act, void* qkv, int B, float eps, uint32_t* flags, int32_t* err, void* stream, bool small, const omni_bb_head* head,
               const omni_bb_ar* ar, bool half) {
    BbArgs a{};
    a.wo = (const uint16_t*)w.wo; a.ln2 = (const uint16_t*)w.ln2; a.wgu = (const uint16_t*)w.wgu; a.wdown = (const uint16_t*)w.wdown;
    a.ln1_next = next ? (const uint16_t*)next->ln1 : nullptr;
    a.wqkv_next = next ? (const uint16_t*)next->wqkv : nullptr;
    a.attn = (const uint16_t*)attn; a.resid = (uint16_t*)resid; a.part = part; a.act = (uint16_t*)act; a.qkv = (uint16_t*)qkv;
    a.H = d.hidden; a.I = d.inter; a.NQ = (d.q_heads + 2 * d.kv_heads) * 128;
    a.B = B; a.nap = g_bb_nap; a.eps = eps; a.flags = flags; a.err = err;
    if (head != nullptr && next == nullptr && !small) {
        a.lm_head = (const uint16_t*)d.lm_head; a.final_norm = (const uint16_t*)d.final_norm; a.logits = head->logits; a.V = d.vocab;
        a.last_hidden = (uint16_t*)head->last_hidden; a.mask = (const uint8_t*)d.allowed_mask; a.mask_fill = head->mask_fill; a.num_live = head->num_live;
    }
#ifdef OMNI_DEBUG_HOOKS
    a.stamps = g_bb_stamps;
#endif
    if (small) {
        // (gate_up on 32-row tiles above 48 rows, on 16-row tiles below: the row group without rows then leaves at launch start -- round 5;
        //  the bits do not depend on the choice since round 6)
        if (B > 48) hipLaunchKernelGGL(bb_chain_small_kernel<true>, dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(bb_chain_small_kernel<false>, dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), 0, (hipStream_t)stream, a);
        OMNI_CHECK_LAUNCH("bb_chain_small");
        return OMNI_OK;
    }
    const int ko = d.q_heads * 128 / 256, kh = d.hidden / 256, ki = d.inter / 256;
    if (ar != nullptr || half) {
        OMNI_CHECK_ARG(!small && B > 32 && (a.lm_head == nullptr || (ar != nullptr && !half)), "bb_chain: the all-reduce / half-grid launches are the 64-row stage set; the half grid has no head stage");
        if (ar != nullptr) {
            OMNI_CHECK_ARG(ar->attn && ar->mlp && ar->attn->epoch == ar->mlp->epoch, "bb_chain: the two all-reduces share one epoch word");
            bb_fill_ar(a.ar_o, *ar->attn);
            bb_fill_ar(a.ar_dn, *ar->mlp);
            a.ar_epoch = ar->attn->epoch;
#define X(KO_, KH_, KI_) if (ko == KO_ && kh == KH_ && ki == KI_) return bb_launch_ar<KO_, KH_, KI_, true>(a, half, (hipStream_t)stream);
            BB_AR_SHAPES(X)
#undef X
        } else if (ko == 8 && kh == 8 && ki == 24) {
            return bb_launch_ar<8, 8, 24, false>(a, true, (hipStream_t)stream);
        }
        omni_set_error("bb_chain: no all-reduce / half-grid stage set for (q width, hidden, intermediate) = (%d, %d, %d)", d.q_heads * 128, d.hidden, d.inter);
        return OMNI_EINVAL;
    }
#ifdef OMNI_DEBUG_HOOKS      // the A/B arms of rounds 3-5 (1.7B triple only)
    a.pf = g_bb_prefetch;
    if (ko == 8 && kh == 8 && ki == 24 && B > 32 && (a.pf || g_bb_deep) && a.lm_head == nullptr) {
        static bool attr = false;
#define BB_ARM(...) bb_chain_kernel<8, 8, 24, __VA_ARGS__>
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)BB_ARM(2, 4, 4, true), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)BB_ARM(4, 8, 8, false), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)BB_ARM(3, 6, 8, false), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)BB_ARM(2, 8, 8, false), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)BB_ARM(2, 4, 4, false, false), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)BB_ARM(2, 4, 4, false, true, true), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)BB_ARM(2, 4, 4, false, true, false, false), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)BB_ARM(2, 4, 4, false, true, false, true, 1), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)BB_ARM(2, 4, 4, false, true, false, true, 2), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)BB_ARM(2, 4, 4, false, true, false, true, 3), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS_BYTES);
            attr = true;
        }
#define BB_LAUNCH(...) hipLaunchKernelGGL((BB_ARM(__VA_ARGS__)), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), BB_LDS_BYTES, (hipStream_t)stream, a)
        if (a.pf) BB_LAUNCH(2, 4, 4, true);
        else if (g_bb_deep == 1) BB_LAUNCH(4, 8, 8, false);
        else if (g_bb_deep == 2) BB_LAUNCH(3, 6, 8, false);
        else if (g_bb_deep == 3) BB_LAUNCH(2, 8, 8, false);
        else if (g_bb_deep == 4) BB_LAUNCH(2, 4, 4, false, false);             // round 3's two-pass gate_up combine
        else if (g_bb_deep == 5) BB_LAUNCH(2, 4, 4, false, true, true);        // gate_up weights ahead of the flags
        else if (g_bb_deep == 6) BB_LAUNCH(2, 4, 4, false, true, false, false); // round 5's exact rstd (timing arm: no longer the launch path's bits)
        else if (g_bb_deep == 7) BB_LAUNCH(2, 4, 4, false, true, false, true, 1);   // nt weight loads: gate_up only
        else if (g_bb_deep == 8) BB_LAUNCH(2, 4, 4, false, true, false, true, 2);   // nt: o_proj / down_proj / qkv
        else if (g_bb_deep == 9) BB_LAUNCH(2, 4, 4, false, true, false, true, 3);   // nt: every weight load of the launch
        else BB_LAUNCH(2, 4, 4, false);
#undef BB_LAUNCH
#undef BB_ARM
        OMNI_CHECK_LAUNCH("bb_chain(arm)");
        return OMNI_OK;
    }
#endif
#define X(KO_, KH_, KI_) if (ko == KO_ && kh == KH_ && ki == KI_) return bb_launch_shape<KO_, KH_, KI_>(a, B, (hipStream_t)stream);
    BB_SHAPES(X)
#undef X
    omni_set_error("bb_chain: no stage set for (q width, hidden, intermediate) = (%d, %d, %d)", d.q_heads * 128, d.hidden, d.inter);
    return OMNI_EINVAL;
}
