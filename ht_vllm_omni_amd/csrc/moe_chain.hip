// The sparse-MoE backbone layer of the Qwen3-Omni talker between its attention launch and its expert GEMMs as ONE persistent launch
// (round 5; reference: the talker's decoder layer, qwen3_omni.py:586-649 / HF Qwen3OmniMoeTalkerTextSparseMoeBlock; BASELINE configs #4, #5):
//     o_proj  ->  { router GEMM + the normalised rows | shared expert gate_up }  ->  { shared expert down_proj, then the top-k routing }
// -- five of the layer's ten launches on the launch-per-op path (o_proj 5.1 us, router 5.1, routing 6.7, shared gate_up 5.4, shared down 4.8
// by rocprof, each behind a 1.7 us boundary and a cold first fetch) -- as three stages of 256 co-resident workgroups with the hand-off
// protocol of cp_chain.hip / chain_gemm.cuh (weights ahead of the flags, sc1 write-through activations, one flag word per workgroup).  The
// expert GEMMs keep their own launches: they stream ~300 MB per layer at the chip's rate and gain nothing from a stage.
// Tiles = the launch path's (pick_tile at these shapes: o_proj 16 x 16 over K = 2048; router 16 x 16, its normalised rows spread over the 8
// column groups; shared gate_up 16 rows x (16 gate + 16 up) columns; shared down 16 x 16 over K = 768), the routing is moe_route_kernel's
// wave-per-token arithmetic: logits, top-k ids / weights, the shared expert's output, the residual stream and its slabs are BIT-IDENTICAL
// to the launch path (tests/test_gpu_chain.py).  Released Omni talker shape only: hidden 1024, 16 x 128 attention width, 128 experts,
// shared expert 768 wide, <= 64 rows, single rank.
#include "chain_gemm.cuh"
#include "common.cuh"
#include "kernels.h"

#define MC_LDS_FLOATS ((CH_WAVES * 6 * 4 * 64) + CH_WAVES * 64)

struct MoeChainArgs {
    const uint16_t *wo, *ln2, *router, *sgu, *sdown;
    const uint16_t* attn;              // fragment-major [64][2048]: the attention launch's output
    uint16_t* resid; float* part;      // fragment-major residual stream [64][1024] + sum(r^2) slabs [64][64]
    uint16_t* normed_rm;               // row-major [B][1024]: the normalised rows (the expert kernels' x)
    uint16_t* logits;                  // row-major [B][128] bf16 router logits
    uint16_t* act;                     // fragment-major [64][768]
    uint16_t* shared;                  // row-major [B][1024]: the shared expert's output (combined by the expert launch)
    int32_t* topk_idx; uint16_t* topk_w;
    int B, top_k, norm_topk, nap; float eps;
    uint32_t* flags; int32_t* err;
};

// top-k routing of row t by one wave: moe_route_kernel's arithmetic (moe.hip) on logits read behind the router stage's flags
__device__ __forceinline__ void mc_route_row(const MoeChainArgs& a, int t) {
    constexpr int E = 128, PER = 2;
    const int lane = threadIdx.x & 63;
    const coh_rsrc_t lrs = coh_rsrc(a.logits);
    float p[PER];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int e = lane + 64 * j;
        // two bf16 logits share a dword: lane reads its own half (sc1: the router workgroups' write-through stores)
        const uint32_t w = coh_ld4(lrs, (uint32_t)(((size_t)t * E + (e & ~1)) * 2));
        p[j] = (e & 1) ? bf_hi(w) : bf_lo(w);
        mx = fmaxf(mx, p[j]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        p[j] = expf(p[j] - mx);
        sum += p[j];
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int j = 0; j < PER; ++j) p[j] = p[j] / sum;
    float vals[8];
    int sel[8];
    float vsum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        vals[k] = 0.f;
        sel[k] = 0;
        if (k < a.top_k) {
            float bv = -2.f;
            int bi = 0x7FFFFFFF;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int e = lane + 64 * j;
                if (p[j] > bv || (p[j] == bv && e < bi)) { bv = p[j]; bi = e; }
            }
            wave_argmax(bv, bi);
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (lane + 64 * j == bi) p[j] = -1.f;
            vals[k] = bv;
            sel[k] = bi;
            vsum += bv;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (lane == k && k < a.top_k) {
            a.topk_idx[(size_t)t * a.top_k + k] = sel[k];
            a.topk_w[(size_t)t * a.top_k + k] = f2bf(a.norm_topk ? vals[k] / vsum : vals[k]);
        }
}

__global__ __launch_bounds__(CH_THREADS) void moe_chain_kernel(const MoeChainArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[MC_LDS_FLOATS];
    ChainGate g;
    chain_gate_init(g, a.flags, a.err);
    g.dom = 8;                          // the stages deal rows to workgroups differently (64 / 48 / 8 column groups): one domain
    g.nap = a.nap;
    const int wg = blockIdx.x;
    constexpr int H = 1024, KO = 2048, E = 128, IS = 768;
    static_assert(KO == 8 * 256 && H == 4 * 256 && IS == 3 * 256, "moe_chain: k-steps per wave");
    // stage codes (error word): 0x2001 .. 0x2003
    chain_gemm<1, 1, 8, 0, OMNI_EPI_RESID>(a.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 63, wg >> 6, lds, g, false, 0x2001,
                                           nullptr);
    if (wg < 192)            // shared expert gate_up: 48 column groups of 16 activation columns x 4 row groups
        chain_gemm<1, 2, 4, 3, OMNI_EPI_SILU_MUL>(a.sgu, a.ln2, a.resid, a.part, H / 16, a.act, 0, nullptr, a.B, IS, a.eps, wg % 48, wg / 48, lds, g, true,
                                                  0x2002, nullptr);
    else if (wg < 224)       // router: 8 column groups x 4 row groups; column group x also writes k-steps x, x + 8, ... of the normalised rows
        chain_gemm<1, 1, 4, 2, OMNI_EPI_BF16>(a.router, a.ln2, a.resid, a.part, H / 16, a.logits, E, nullptr, a.B, E, a.eps, (wg - 192) & 7, (wg - 192) >> 3,
                                              lds, g, true, 0x2002, nullptr, nullptr, ChainNoPrefetch(), a.normed_rm);
    else
        chain_gate_skip(g);
    chain_gemm<1, 1, 3, 0, OMNI_EPI_BF16>(a.sdown, nullptr, a.act, nullptr, 0, a.shared, H, nullptr, a.B, H, a.eps, wg & 63, wg >> 6, lds, g, true, 0x2003,
                                          nullptr);
    // the routing: one wave per row, on the first ceil(B / 8) workgroups, behind their down_proj tile (its wait covered the router stage;
    // the ids / weights are for the NEXT launch: plain stores)
    const int t = wg * CH_WAVES + (int)(threadIdx.x >> 6);
    if (t < a.B && !g.dead) mc_route_row(a, t);
}

// ---- the layer's tail: { combine -> next layer's qkv } behind the two expert GEMM launches.  The combine is moe_combine_kernel<2>'s
// arithmetic (moe.hip: per token the experts' weighted outputs in ascending expert index, bf16 accumulation, + the sigmoid-gated shared
// expert, r = bf16(r + delta) on the fragment-major stream, the slabs) on the first 256 threads of workgroup t; the qkv GEMM reads r and
// the slabs behind the combine's flags (16 rows x 64 columns per workgroup: the launch path's 16 x 32 tile differs in shape only -- a
// column's dot product, its split over the 8 waves and the combine order do not depend on how many columns a workgroup owns)
struct MoeTailArgs {
    const uint16_t *y, *x, *w_sg, *shared; const int32_t* topk_idx;
    uint16_t* resid; float* part;
    const uint16_t *ln1_next, *wqkv_next; uint16_t* qkv;
    int B, top_k, e_lo, e_hi, nap; float eps;
    uint32_t* flags; int32_t* err;
};

__device__ __forceinline__ void mc_combine_row(const MoeTailArgs& a, int t, float* lds) {
    constexpr int H = 1024, MAXK = 8;
    int* order = reinterpret_cast<int*>(lds);
    float* red = lds + 8;
    float* s_gate = lds + 12;
    const int top_k = a.top_k;
    const bool worker = threadIdx.x < 256;
    if (threadIdx.x < (unsigned)top_k) {                  // thread k: rank of expert k among the token's experts
        const int mine = a.topk_idx[(size_t)t * top_k + threadIdx.x];
        int rank = 0;
        for (int j = 0; j < top_k; ++j) rank += a.topk_idx[(size_t)t * top_k + j] < mine ? 1 : 0;
        order[rank] = threadIdx.x;
    }
    float gate = 0.f;
    if (a.shared && worker) {
        float dsum = 0.f;
        for (int h = threadIdx.x; h < H; h += 256) dsum = fmaf(bf2f(a.x[(size_t)t * H + h]), bf2f(a.w_sg[h]), dsum);
        dsum = wave_sum(dsum);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dsum;
    }
    __syncthreads();
    if (a.shared) {
        if (threadIdx.x == 0) {
            const float lg = bfround((red[0] + red[1]) + (red[2] + red[3]));
            *s_gate = bfround(1.0f / (1.0f + expf(-lg)));
        }
        __syncthreads();
        gate = *s_gate;
    }
    if (!worker) return;
    int ord[MAXK];
    bool loc[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        ord[k] = k < top_k ? order[k] : 0;
        const int ek = k < top_k ? a.topk_idx[(size_t)t * top_k + ord[k]] : -1;
        loc[k] = ek >= a.e_lo && ek < a.e_hi;
    }
    const coh_rsrc_t rrs = coh_rsrc(a.resid), prs = coh_rsrc(a.part);
    for (int h = threadIdx.x * 2; h < H; h += 512) {
        uint32_t yv[MAXK];
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            yv[k] = (k < top_k && loc[k]) ? *reinterpret_cast<const uint32_t*>(a.y + ((size_t)t * top_k + ord[k]) * H + h) : 0u;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < top_k && loc[k]) { a0 = bfround(a0 + bf_lo(yv[k])); a1 = bfround(a1 + bf_hi(yv[k])); }
        if (a.shared) {
            const uint32_t sv = *reinterpret_cast<const uint32_t*>(a.shared + (size_t)t * H + h);
            a0 = bfround(a0 + bfround(gate * bf_lo(sv)));
            a1 = bfround(a1 + bfround(gate * bf_hi(sv)));
        }
        const uint32_t off = (uint32_t)frag_off(t, h, H) * 2;
        const uint32_t ro = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(a.resid) + off);      // written by an earlier launch
        const float r0 = bfround(bf_lo(ro) + a0), r1 = bfround(bf_hi(ro) + a1);
        coh_st4(rrs, off, pack_bf2(r0, r1));
        float ss = r0 * r0 + r1 * r1;                      // 8 adjacent threads hold the 16 columns of one slab
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        ss += __shfl_xor(ss, 4, 64);
        if ((threadIdx.x & 7) == 0) coh_st4(prs, (uint32_t)((h >> 4) * 64 + t) * 4, __float_as_uint(ss));
    }
}

__global__ __launch_bounds__(CH_THREADS) void moe_tail_kernel(const MoeTailArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[MC_LDS_FLOATS];
    ChainGate g;
    chain_gate_init(g, a.flags, a.err);
    g.dom = 8;
    g.nap = a.nap;
    const int wg = blockIdx.x;
    constexpr int H = 1024, NQ = 2560;
    if (wg < a.B) {                      // stage 0x2101: the combine of row wg
        mc_combine_row(a, wg, lds);
        chain_gate_arrive(g);
    } else {
        chain_gate_skip(g);
    }
    if (wg < 160)                        // stage 0x2102: the next layer's qkv, 40 column groups of 64 x 4 row groups
        chain_gemm<1, 4, 4, 3, OMNI_EPI_BF16>(a.wqkv_next, a.ln1_next, a.resid, a.part, H / 16, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg % 40, wg / 40, lds, g,
                                              true, 0x2102, nullptr);
    else
        chain_gate_skip(g);
}

OMNI_KNOB g_moe_chain = 1, g_moe_chain_nap = 1, g_moe_tail = 1;
#ifdef OMNI_DEBUG_HOOKS
extern "C" void omni_debug_moe_chain(int on) { g_moe_chain = on != 0; g_moe_tail = on != 2; }      // 2: the first chain only (the tail as launches)
#endif
bool k_moe_tail_enabled() { return g_moe_tail != 0; }

int k_moe_tail(const omni_talker_desc& d, const omni_layer_weights& w, const omni_layer_weights& next, const void* y_ws, const void* normed_rm,
               const void* shared, const int32_t* topk_idx, void* resid, float* part, void* qkv, int B, uint32_t* flags, int32_t* err, void* stream) {
    MoeTailArgs a{};
    a.y = (const uint16_t*)y_ws; a.x = (const uint16_t*)normed_rm; a.w_sg = (const uint16_t*)w.moe_shared_gate; a.shared = (const uint16_t*)shared;
    a.topk_idx = topk_idx; a.resid = (uint16_t*)resid; a.part = part;
    a.ln1_next = (const uint16_t*)next.ln1; a.wqkv_next = (const uint16_t*)next.wqkv; a.qkv = (uint16_t*)qkv;
    a.B = B; a.top_k = d.moe_top_k; a.e_lo = d.moe_e0; a.e_hi = d.moe_e0 + (d.moe_experts_local > 0 ? d.moe_experts_local : d.moe_experts);
    a.nap = g_moe_chain_nap; a.eps = d.eps; a.flags = flags; a.err = err;
    hipLaunchKernelGGL(moe_tail_kernel, dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), 0, (hipStream_t)stream, a);
    OMNI_CHECK_LAUNCH("moe_tail");
    return OMNI_OK;
}

bool k_moe_chain_supported(const omni_talker_desc& d, int B, bool has_ar) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 0;
    }
    return g_moe_chain && d.cp_chain && cus >= OMNI_CHAIN_WGS && d.fused_norm && d.frag_layout && !has_ar && d.moe_experts == 128 &&
           (d.moe_experts_local == 0 || d.moe_experts_local == 128) && d.moe_top_k >= 1 && d.moe_top_k <= 8 && d.moe_shared_inter == 768 &&
           d.hidden == 1024 && d.head_dim == 128 && d.q_heads * 128 == 2048 && d.kv_heads == 2 && B >= 1 && B <= 64;
}

int k_moe_chain(const omni_talker_desc& d, const omni_layer_weights& w, const void* attn, void* resid, float* part, void* normed_rm, void* logits,
                void* act, void* shared, int32_t* topk_idx, void* topk_w, int B, uint32_t* flags, int32_t* err, void* stream) {
    MoeChainArgs a{};
    a.wo = (const uint16_t*)w.wo; a.ln2 = (const uint16_t*)w.ln2; a.router = (const uint16_t*)w.moe_router;
    a.sgu = (const uint16_t*)w.moe_shared_gate_up; a.sdown = (const uint16_t*)w.moe_shared_down;
    a.attn = (const uint16_t*)attn; a.resid = (uint16_t*)resid; a.part = part; a.normed_rm = (uint16_t*)normed_rm; a.logits = (uint16_t*)logits;
    a.act = (uint16_t*)act; a.shared = (uint16_t*)shared; a.topk_idx = topk_idx; a.topk_w = (uint16_t*)topk_w;
    a.B = B; a.top_k = d.moe_top_k; a.norm_topk = d.moe_norm_topk; a.nap = g_moe_chain_nap; a.eps = d.eps;
    a.flags = flags; a.err = err;
    hipLaunchKernelGGL(moe_chain_kernel, dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), 0, (hipStream_t)stream, a);
    OMNI_CHECK_LAUNCH("moe_chain");
    return OMNI_OK;
}
