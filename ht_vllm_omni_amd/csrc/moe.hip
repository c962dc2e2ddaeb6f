// Sparse-MoE MLP of the Qwen3-Omni talker backbone at decode batch sizes (T <= 64 tokens).
// Arithmetic = oracle/talker_oracle.py moe_route / moe_block (HF Qwen3OmniMoeTalkerTextSparseMoeBlock, bf16 rounding
// points): bf16 router logits -> fp32 softmax -> top-k (weights to bf16) -> per expert gate_up, SiLU * up, down,
// * weight (each rounded to bf16) -> per token bf16 accumulation IN EXPERT-INDEX ORDER -> + sigmoid-gated shared expert.
//
// HBM-bound like the dense MLP, but the bytes are the experts that were hit: with T * k = 512 assignments over 128
// experts nearly every expert streams its 2.4 MB once per layer.  One workgroup = (expert, n group); it finds the
// expert's tokens itself (T * k <= 512 routing entries, one per thread, ballot compaction in token order), exits when
// the expert was not hit, and otherwise runs the skinny-GEMM scheme of gemm.hip: expert weights fragment-major straight
// into MFMA A operands, the (gathered) activation rows as B operand, 8 waves splitting K, LDS combine.
#include "common.cuh"
#include "kernels.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define MOE_WAVES 8
#define MOE_THREADS (MOE_WAVES * 64)
#define MOE_MAXT 64            // tokens per call (decode batch)
#define MOE_MAXK 8
#define MOE_MAXE 256

__device__ __forceinline__ f32x4 moe_mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------- routing: one wave per token
__global__ __launch_bounds__(256) void moe_route_kernel(const uint16_t* __restrict__ logits, int T, int E, int top_k, int norm,
                                                        int32_t* __restrict__ topk_idx, uint16_t* __restrict__ topk_w) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    constexpr int PER = MOE_MAXE / 64;
    float p[PER];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int e = lane + 64 * j;
        p[j] = e < E ? bf2f(logits[(size_t)t * E + e]) : -INFINITY;
        mx = fmaxf(mx, p[j]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        p[j] = (lane + 64 * j < E) ? expf(p[j] - mx) : 0.f;
        sum += p[j];
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int j = 0; j < PER; ++j) p[j] = (lane + 64 * j < E) ? p[j] / sum : -1.f;      // -1: never selected
    // top_k rounds of a wave-wide argmax (value desc, index asc); fully unrolled so vals[] stays in registers
    float vals[MOE_MAXK];
    int sel[MOE_MAXK];
    float vsum = 0.f;
#pragma unroll
    for (int k = 0; k < MOE_MAXK; ++k) {
        vals[k] = 0.f;
        sel[k] = 0;
        if (k < top_k) {
            float bv = -2.f;
            int bi = 0x7FFFFFFF;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int e = lane + 64 * j;
                if (p[j] > bv || (p[j] == bv && e < bi)) { bv = p[j]; bi = e; }
            }
            wave_argmax(bv, bi);      // DPP / permlane exchanges: 8 rounds of ds_bpermute butterflies were most of this kernel
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (lane + 64 * j == bi) p[j] = -1.f;
            vals[k] = bv;
            sel[k] = bi;
            vsum += bv;
        }
    }
    // lane k writes entry k
#pragma unroll
    for (int k = 0; k < MOE_MAXK; ++k)
        if (lane == k && k < top_k) {
            topk_idx[(size_t)t * top_k + k] = sel[k];
            topk_w[(size_t)t * top_k + k] = f2bf(norm ? vals[k] / vsum : vals[k]);
        }
}

// ---------------------------------------------------------------- expert GEMMs
struct MoeArgs {
    const uint16_t* x;          // phase 1: bf16 [T, K] tokens; phase 2: bf16 [T * k, K] activation slots
    const void* W;              // [E_local, N_rows, K] fragment-major per expert (phase 1: N_rows = 2 * I gate | up; phase 2: H);
                                // bf16, or (W8) fp8 e4m3fn bytes in the same element order + w_scale
    const float* w_scale;       // W8: fp32 [E_local, N_rows] per-output-row dequantisation scales
    const int32_t* topk_idx; const uint16_t* topk_w;
    uint16_t* out;              // phase 1: act [T * k, I]; phase 2: y [T * k, H] (already * routing weight)
    int T, top_k, K, N;         // N = output columns (I or H)
    int e0;                     // expert parallel: this rank holds experts [e0, e0 + gridDim.y) of the router's numbering
};

// fp8 e4m3fn weights, dequantised in registers to the bf16 values bf16(fp8 * row scale) -- exactly the matrix a
// weight-only-dequantised bf16 model holds (oracle: HF block on those weights) -- then the same bf16 MFMA
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 deq_fp8x8(u32x2 q, float s) {
    float f[8];
    unpack_fp8x4(q[0], f);
    unpack_fp8x4(q[1], f + 4);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf2(f[2 * j] * s, f[2 * j + 1] * s);
    return o;
}

// PHASE 1: NT = 2 tiles (16 gate + 16 up columns) -> SiLU * up; PHASE 2: NT tiles of 16 output columns, * weight
template <int PHASE, int NT, bool W8>
__global__ __launch_bounds__(MOE_THREADS) void moe_expert_kernel(const MoeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [WAVES][NT*4][64] partial sums
    __shared__ int s_tok[MOE_MAXT], s_slot[MOE_MAXT];
    __shared__ int s_cnt[MOE_WAVES + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int e = a.e0 + blockIdx.y;                     // the router's expert id; blockIdx.y indexes this rank's weights
    const int K = a.K, nsteps = K >> 5;

    // ---- this expert's (token, slot) list, in token order: entry i = token i / k, position i % k
    const int entries = a.T * a.top_k;
    const bool mine = (int)threadIdx.x < entries && a.topk_idx[threadIdx.x] == e;
    const uint64_t bal = __ballot(mine);
    if (lane == 0) s_cnt[wave + 1] = __builtin_popcountll(bal);
    if (threadIdx.x == 0) s_cnt[0] = 0;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < MOE_WAVES; ++w) {
        const int c = s_cnt[w + 1];
        if (w < wave) base += c;
        total += c;
    }
    if (total == 0) return;                               // expert not hit: nothing to stream
    if (mine) {
        const int o = base + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        if (o < MOE_MAXT) { s_tok[o] = threadIdx.x / a.top_k; s_slot[o] = threadIdx.x; }
    }
    __syncthreads();
    const int n_e = min(total, MOE_MAXT);                 // a token selects an expert at most once: total <= T <= 64

    // ---- W rows of this workgroup (fragment-major per expert: tile (row tile, k step) = 512 contiguous elements)
    const size_t rows_e = PHASE == 1 ? 2 * (size_t)a.N : (size_t)a.N;
    constexpr int WB = W8 ? 1 : 2;                        // bytes per weight element
    const char* We = reinterpret_cast<const char*>(a.W) + (size_t)blockIdx.y * rows_e * K * WB;
    const char* wrow[NT];
    int tile_of[NT];
    if (PHASE == 1) {
        tile_of[0] = blockIdx.x;                          // gate tile
        tile_of[NT - 1] = a.N / 16 + blockIdx.x;          // matching up tile
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) tile_of[j] = blockIdx.x * NT + j;
    }
    float wsc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        wrow[j] = We + (((size_t)tile_of[j] * nsteps) * 512 + lane * 8) * WB;
        wsc[j] = W8 ? a.w_scale[(size_t)blockIdx.y * rows_e + tile_of[j] * 16 + r] : 1.0f;      // lane (r, q) holds W row r of the tile
    }

    for (int m0 = 0; m0 < n_e; m0 += 16) {
        const int mrow = min(m0 + r, n_e - 1);            // rows past the end: valid address, result discarded
        const uint16_t* xrow = a.x + (size_t)(PHASE == 1 ? s_tok[mrow] : s_slot[mrow]) * K + 8 * q;
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = wave; ks < nsteps; ks += MOE_WAVES) {
            const u32x4 X = *reinterpret_cast<const u32x4*>(xrow + ks * 32);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                u32x4 Wf;
                if (W8) Wf = deq_fp8x8(__builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wrow[j] + (size_t)ks * 512)), wsc[j]);
                else Wf = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[j] + (size_t)ks * 1024));
                acc[j] = moe_mfma(Wf, X, acc[j]);
            }
        }
        __syncthreads();                                   // previous m tile's LDS reads are done
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) lds[(wave * NT * 4 + j * 4 + g) * 64 + lane] = acc[j][g];
        __syncthreads();
        // item = lane l of output tile j: row m = l & 15, 4 consecutive columns n = 4 * (l >> 4) + g
        constexpr int NTO = PHASE == 1 ? 1 : NT;
        for (int it = threadIdx.x; it < NTO * 64; it += MOE_THREADS) {
            const int l = it & 63, j = it >> 6;
            const int m = m0 + (l & 15);
            if (m >= n_e) continue;
            float v[4], v2[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < MOE_WAVES; ++w) {
                    s1 += lds[(w * NT * 4 + j * 4 + g) * 64 + l];
                    if (PHASE == 1) s2 += lds[(w * NT * 4 + (NT - 1) * 4 + g) * 64 + l];
                }
                v[g] = s1;
                v2[g] = s2;
            }
            const int slot = s_slot[m];
            float o[4];
            if (PHASE == 1) {
                const int n = blockIdx.x * 16 + 4 * (l >> 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float gt = bfround(v[g]), up = bfround(v2[g]);
                    o[g] = bfround(gt / (1.0f + expf(-gt))) * up;
                }
                *reinterpret_cast<uint2*>(a.out + (size_t)slot * a.N + n) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            } else {
                const int n = (blockIdx.x * NT + j) * 16 + 4 * (l >> 4);
                const float wgt = bf2f(a.topk_w[slot]);
#pragma unroll
                for (int g = 0; g < 4; ++g) o[g] = bfround(v[g]) * wgt;
                *reinterpret_cast<uint2*>(a.out + (size_t)slot * a.N + n) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        }
    }
}

// ---------------------------------------------------------------- combine: one workgroup per token
// out = bf16( sum over the token's experts in ascending expert index of y[slot] (bf16 accumulate)
//             + bf16(sigmoid(bf16(x . w_shared_gate)) * shared) )
// OUT 0: out bf16 [T, H] row-major (the separate-norm path's mlp_out).  OUT 1: the same values fragment-major (a tensor- /
// expert-parallel rank's partial, handed to omni_allreduce_resid).  OUT 2: the norm-free stream -- out IS the fragment-major
// residual r: r = bf16(r + delta) in place, and part[c][t] = sum of r^2 over the 16 columns of slab c (what gemm.hip's
// EPI_RESID leaves: the next RMSNorm is folded into the GEMMs that read r).
template <int OUT>
__global__ __launch_bounds__(256) void moe_combine_kernel(const uint16_t* __restrict__ y, const int32_t* __restrict__ topk_idx,
                                                          const uint16_t* __restrict__ x, const uint16_t* __restrict__ w_sg,
                                                          const uint16_t* __restrict__ shared, uint16_t* __restrict__ out,
                                                          float* __restrict__ part, int top_k, int H, int e_lo, int e_hi) {
    const int t = blockIdx.x;
    __shared__ int order[MOE_MAXK];
    __shared__ float red[4];
    __shared__ float s_gate;
    if (threadIdx.x < (unsigned)top_k) {                  // thread k: rank of expert k among the token's experts
        const int mine = topk_idx[(size_t)t * top_k + threadIdx.x];
        int rank = 0;
        for (int j = 0; j < top_k; ++j) rank += topk_idx[(size_t)t * top_k + j] < mine ? 1 : 0;
        order[rank] = threadIdx.x;                        // experts of a token are distinct: ranks are a permutation
    }
    float gate = 0.f;
    if (shared) {
        float d = 0.f;
        for (int h = threadIdx.x; h < H; h += 256) d = fmaf(bf2f(x[(size_t)t * H + h]), bf2f(w_sg[h]), d);
        d = wave_sum(d);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    }
    __syncthreads();
    if (shared) {
        if (threadIdx.x == 0) {
            const float lg = bfround((red[0] + red[1]) + (red[2] + red[3]));
            s_gate = bfround(1.0f / (1.0f + expf(-lg)));
        }
        __syncthreads();
        gate = s_gate;
    }
    // expert parallel: slots of experts another rank holds were never written here -- they count as zero (the ranks'
    // partial outputs are summed by the all-reduce that follows)
    int ord[MOE_MAXK];
    bool loc[MOE_MAXK];
#pragma unroll
    for (int k = 0; k < MOE_MAXK; ++k) {
        ord[k] = k < top_k ? order[k] : 0;
        const int ek = k < top_k ? topk_idx[(size_t)t * top_k + ord[k]] : -1;
        loc[k] = ek >= e_lo && ek < e_hi;
    }
    for (int h = threadIdx.x * 2; h < H; h += 512) {       // two columns per thread: 4-B accesses
        uint32_t yv[MOE_MAXK];
#pragma unroll
        for (int k = 0; k < MOE_MAXK; ++k)
            yv[k] = (k < top_k && loc[k]) ? *reinterpret_cast<const uint32_t*>(y + ((size_t)t * top_k + ord[k]) * H + h) : 0u;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < MOE_MAXK; ++k)
            if (k < top_k && loc[k]) { a0 = bfround(a0 + bf_lo(yv[k])); a1 = bfround(a1 + bf_hi(yv[k])); }
        if (shared) {
            const uint32_t sv = *reinterpret_cast<const uint32_t*>(shared + (size_t)t * H + h);
            a0 = bfround(a0 + bfround(gate * bf_lo(sv)));
            a1 = bfround(a1 + bfround(gate * bf_hi(sv)));
        }
        if (OUT == 0) {
            *reinterpret_cast<uint32_t*>(out + (size_t)t * H + h) = pack_bf2(a0, a1);
        } else if (OUT == 1) {
            *reinterpret_cast<uint32_t*>(out + frag_off(t, h, H)) = pack_bf2(a0, a1);
        } else {
            uint32_t* rp = reinterpret_cast<uint32_t*>(out + frag_off(t, h, H));
            const uint32_t ro = *rp;
            const float r0 = bfround(bf_lo(ro) + a0), r1 = bfround(bf_hi(ro) + a1);
            *rp = pack_bf2(r0, r1);
            float ss = r0 * r0 + r1 * r1;                  // 8 adjacent threads hold the 16 columns of one slab
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            ss += __shfl_xor(ss, 4, 64);
            if ((threadIdx.x & 7) == 0) part[(size_t)(h >> 4) * 64 + t] = ss;
        }
    }
}

// ---------------------------------------------------------------- C entry points
extern "C" int omni_moe_route(const void* logits, int T, int E, int top_k, int norm_topk_prob, int32_t* topk_idx, void* topk_w,
                              void* stream) {
    OMNI_CHECK_ARG(logits && topk_idx && topk_w, "omni_moe_route: null pointer");
    OMNI_CHECK_ARG(T >= 0 && E >= 1 && E <= MOE_MAXE && top_k >= 1 && top_k <= MOE_MAXK && top_k <= E,
                   "omni_moe_route: T=%d E=%d top_k=%d (E <= %d, top_k <= %d)", T, E, top_k, MOE_MAXE, MOE_MAXK);
    if (T == 0) return OMNI_OK;
    hipLaunchKernelGGL(moe_route_kernel, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)logits, T, E,
                       top_k, norm_topk_prob, topk_idx, (uint16_t*)topk_w);
    OMNI_CHECK_LAUNCH("omni_moe_route");
    return OMNI_OK;
}

// E = experts held here (the router's ids [e0, e0 + E)); scales != NULL: fp8 e4m3fn weights + per-row fp32 scales
static int moe_experts_impl(const void* x, const int32_t* topk_idx, const void* topk_w, const void* w_gate_up, const float* s_gate_up,
                            const void* w_down, const float* s_down, const void* shared, const void* w_shared_gate, void* act_ws,
                            void* y_ws, void* out, int T, int H, int I, int E, int e0, int top_k, void* stream, int out_mode = 0,
                            float* part = nullptr) {
    OMNI_CHECK_ARG(x && topk_idx && topk_w && w_gate_up && w_down && act_ws && y_ws && (out || out_mode == 3), "omni_moe_experts: null pointer");
    OMNI_CHECK_ARG(T >= 1 && T <= MOE_MAXT && top_k >= 1 && top_k <= MOE_MAXK && T * top_k <= MOE_THREADS,
                   "omni_moe_experts: T=%d top_k=%d (T <= %d, T * top_k <= %d)", T, top_k, MOE_MAXT, MOE_THREADS);
    OMNI_CHECK_ARG(H % 64 == 0 && I % 32 == 0 && E >= 1 && E <= 65535 && e0 >= 0, "omni_moe_experts: H=%d I=%d E=%d e0=%d (H %% 64, I %% 32)", H, I, E, e0);
    OMNI_CHECK_ARG(!shared || w_shared_gate, "omni_moe_experts: shared expert output without its gate weight");
    OMNI_CHECK_ARG((s_gate_up == nullptr) == (s_down == nullptr), "omni_moe_experts: fp8 weights need both scale arrays");
    hipStream_t st = (hipStream_t)stream;
    const bool w8 = s_gate_up != nullptr;
    MoeArgs a{};
    a.topk_idx = topk_idx; a.topk_w = (const uint16_t*)topk_w; a.T = T; a.top_k = top_k; a.e0 = e0;
    // phase 1: act[slot] = silu(x . Wg^T) * (x . Wu^T)
    a.x = (const uint16_t*)x; a.W = w_gate_up; a.w_scale = s_gate_up; a.out = (uint16_t*)act_ws; a.K = H; a.N = I;
    if (w8) hipLaunchKernelGGL((moe_expert_kernel<1, 2, true>), dim3(I / 16, E), dim3(MOE_THREADS), MOE_WAVES * 2 * 4 * 64 * sizeof(float), st, a);
    else hipLaunchKernelGGL((moe_expert_kernel<1, 2, false>), dim3(I / 16, E), dim3(MOE_THREADS), MOE_WAVES * 2 * 4 * 64 * sizeof(float), st, a);
    OMNI_CHECK_LAUNCH("omni_moe_experts(gate_up)");
    // phase 2: y[slot] = bf16(act . Wd^T) * weight
    a.x = (const uint16_t*)act_ws; a.W = w_down; a.w_scale = s_down; a.out = (uint16_t*)y_ws; a.K = I; a.N = H;
    if (w8) hipLaunchKernelGGL((moe_expert_kernel<2, 4, true>), dim3(H / 64, E), dim3(MOE_THREADS), MOE_WAVES * 4 * 4 * 64 * sizeof(float), st, a);
    else hipLaunchKernelGGL((moe_expert_kernel<2, 4, false>), dim3(H / 64, E), dim3(MOE_THREADS), MOE_WAVES * 4 * 4 * 64 * sizeof(float), st, a);
    OMNI_CHECK_LAUNCH("omni_moe_experts(down)");
    if (out_mode == 3) return OMNI_OK;      // the combine is a stage of the caller's persistent launch (moe_chain.hip moe_tail_kernel)
#define COMBINE(M_)                                                                                                        \
    hipLaunchKernelGGL(moe_combine_kernel<M_>, dim3(T), dim3(256), 0, st, (const uint16_t*)y_ws, topk_idx, (const uint16_t*)x, \
                       (const uint16_t*)w_shared_gate, (const uint16_t*)shared, (uint16_t*)out, part, top_k, H, e0, e0 + E)
    if (out_mode == 0) COMBINE(0);
    else if (out_mode == 1) COMBINE(1);
    else COMBINE(2);
#undef COMBINE
    OMNI_CHECK_LAUNCH("omni_moe_experts(combine)");
    return OMNI_OK;
}

extern "C" int omni_moe_experts(const void* x, const int32_t* topk_idx, const void* topk_w, const void* w_gate_up,
                                const void* w_down, const void* shared, const void* w_shared_gate, void* act_ws, void* y_ws,
                                void* out, int T, int H, int I, int E, int top_k, void* stream) {
    return moe_experts_impl(x, topk_idx, topk_w, w_gate_up, nullptr, w_down, nullptr, shared, w_shared_gate, act_ws, y_ws, out, T, H, I,
                            E, 0, top_k, stream);
}

extern "C" int omni_moe_experts_ex(const void* x, const int32_t* topk_idx, const void* topk_w, const void* w_gate_up,
                                   const float* s_gate_up, const void* w_down, const float* s_down, const void* shared,
                                   const void* w_shared_gate, void* act_ws, void* y_ws, void* out, int T, int H, int I,
                                   int E_local, int e0, int top_k, void* stream) {
    return moe_experts_impl(x, topk_idx, topk_w, w_gate_up, s_gate_up, w_down, s_down, shared, w_shared_gate, act_ws, y_ws, out, T, H, I,
                            E_local, e0, top_k, stream);
}

// the two expert GEMM launches alone: y_ws [T * k, H] holds the weighted expert outputs, the combine is the caller's (moe_chain.hip)
int k_moe_experts_phases(const void* x, const int32_t* topk_idx, const void* topk_w, const void* w_gate_up, const float* s_gate_up, const void* w_down,
                         const float* s_down, void* act_ws, void* y_ws, int T, int H, int I, int E_local, int e0, int top_k, void* stream) {
    return moe_experts_impl(x, topk_idx, topk_w, w_gate_up, s_gate_up, w_down, s_down, nullptr, nullptr, act_ws, y_ws, nullptr, T, H, I, E_local, e0,
                            top_k, stream, 3);
}

extern "C" int omni_moe_experts_resid(const void* x, const int32_t* topk_idx, const void* topk_w, const void* w_gate_up,
                                      const float* s_gate_up, const void* w_down, const float* s_down, const void* shared,
                                      const void* w_shared_gate, void* act_ws, void* y_ws, void* resid_frag, float* part,
                                      void* partial_frag, int T, int H, int I, int E_local, int e0, int top_k, void* stream) {
    OMNI_CHECK_ARG(partial_frag || (resid_frag && part), "omni_moe_experts_resid: a partial buffer, or the residual stream and its slabs");
    OMNI_CHECK_ARG(H % 32 == 0 && T <= 64, "omni_moe_experts_resid: fragment-major output needs H %% 32 == 0 and T <= 64 (slab stride)");
    return moe_experts_impl(x, topk_idx, topk_w, w_gate_up, s_gate_up, w_down, s_down, shared, w_shared_gate, act_ws, y_ws,
                            partial_frag ? partial_frag : resid_frag, T, H, I, E_local, e0, top_k, stream, partial_frag ? 1 : 2, part);
}
