// Causal prefill attention over the paged KV cache on the matrix cores (the one place of the path where the
// contraction is dense: SURVEY 8d "MFMA utilisation is reported only for prefill QK^T / PV").
//
// Semantics = oracle/talker_oracle.py attention_rows: token t of request r at position p attends to cache
// positions 0..p of r (the chunk's own K/V were written to the cache by qknorm_rope_kvwrite just before, so
// quantised caches are read back dequantised exactly as the decode kernel does); fp32 scores and softmax
// statistics; P enters the PV product as three bf16 terms that sum to the fp32 value (the oracle keeps P in fp32).
//
// Workgroup = one 16-token tile of the flattened token array x one kv head; wave g = q head kvh*G + g.
// Both products are computed TRANSPOSED so that a lane owns one query row from the first MFMA to the store:
//   S^T[key][q]  = K . Q^T      A = K tile  (lane (key r, chunk qk): 8 dims, ds_read_b128 from LDS)
//                               B = Q^T     (lane (q row c, chunk qk): 8 dims, held in registers all kernel)
//                               D: lane (c = q row, qk) holds keys 4 qk + reg of each 16-key tile
//   O^T[d][q]    = V^T . P^T    B = P^T     (lane (c, qk): 8 keys) -- MFMA's contraction order is free as long
//                               as A and B agree, so contraction slot (qk, j) is DEFINED as key
//                               (j < 4 ? 4 qk + j : 16 + 4 qk + j - 4): exactly the 8 scores the lane already holds
//                               after the two S^T tiles -> P never moves between lanes.
//                               A = V^T     (lane (d r, qk): the same 8 keys at one d: 8 ds_read_u16 of the
//                               row-major V tile, row pitch 264 B => the 4 qk groups hit disjoint banks)
//   D: lane (c = q row, qk) holds d = 16 dt + 4 qk + reg: the per-row rescale (online softmax) is per lane.
// Tiles may straddle requests (ragged prompts): the tile is walked one request segment at a time, rows outside
// the segment masked; a 16-row tile of 32..160-token prompts straddles at most one boundary.
#include "common.cuh"
#include "kernels.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define PF_BN 32          // keys per LDS tile (two 16-key MFMA tiles)
#define PF_ROWB 264       // LDS row pitch in bytes: 128 bf16 + 8 B (4 qk groups x 4 keys apart -> banks 0-7, 8-15, ...)
#define PF_LOG2E 1.4426950408889634f
// waves per SIMD asked of the register allocator (left alone it takes 177 registers = 2 waves: the kernel is latency-bound, two barriers per
// 32-key tile): 4 (<= 128 registers) for the fp8 cache, 3 for the formats whose staging registers would spill at 4
#define PF_WAVES_PER_EU(KV, G) ((G) < 2 ? 1 : ((KV) == OMNI_KV_FP8 ? 4 : 3))

struct PFArgs {
    const uint16_t* q;             // bf16 [T, Hq, 128] (normed + roped)
    const void* k_cache; const void* v_cache; const float* k_scales; const float* v_scales;
    const int32_t* block_table; int bt_stride; const int32_t* req_of_tok; const int32_t* positions;
    uint16_t* out; int T, q_heads, kv_heads, bs; float k_scale, v_scale, sm_scale; int out_frag;
    int bs_shift;                  // log2(bs): blocks are powers of two (token -> block / offset by shift and mask)
};

__device__ __forceinline__ f32x4 pf_mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// 16 cache bytes: loaded raw (pf_load), then written to LDS as bf16 values (pf_store; fp8 e4m3fn and int8 are exact in bf16; scales are
// applied to scores / P).  Two steps so that the NEXT tile's bytes travel while the current tile is computed on.
template <int KV>
__device__ __forceinline__ u32x4 pf_load(const void* cache, size_t row, int ch) {
    if (OMNI_KV_IS16(KV)) return *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(cache) + row * 128 + ch * 8);
    return *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(cache) + row * 128 + ch * 16);
}
template <int KV>
__device__ __forceinline__ void pf_store(unsigned char* dst_row, const u32x4 v, int ch) {
    if (KV == OMNI_KV_BF16) {
        *reinterpret_cast<u32x4*>(dst_row + ch * 16) = v;
    } else if (KV == OMNI_KV_FP16) {
        // half -> bf16: exact for the values a bf16 model writes (a bf16 number inside the half range keeps its 8 significant bits)
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; ++w) o[w] = pack_bf2(h_lo(v[w]), h_hi(v[w]));
        *reinterpret_cast<u32x4*>(dst_row + ch * 16) = o;
    } else {
        uint32_t o[8];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float f[4];
            if (KV == OMNI_KV_FP8) {
                unpack_fp8x4(v[w], f);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = (float)(int8_t)((v[w] >> (8 * e)) & 0xFF);
            }
            o[2 * w] = pack_bf2(f[0], f[1]);
            o[2 * w + 1] = pack_bf2(f[2], f[3]);
        }
        *reinterpret_cast<u32x4*>(dst_row + ch * 32) = (u32x4){o[0], o[1], o[2], o[3]};
        *reinterpret_cast<u32x4*>(dst_row + ch * 32 + 16) = (u32x4){o[4], o[5], o[6], o[7]};
    }
}

template <int KV, int G>
__global__ __launch_bounds__(64 * G, PF_WAVES_PER_EU(KV, G)) void paged_attn_prefill_mfma_kernel(const PFArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[PF_BN * PF_ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char sV[PF_BN * PF_ROWB];
    __shared__ float sKs[PF_BN], sVs[PF_BN];
    constexpr int NT = 64 * G;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = lane & 15, qk = lane >> 4;
    const int t0 = blockIdx.x * 16, kvh = blockIdx.y;
    const int nrows = min(16, a.T - t0);
    const int tok = t0 + min(c, nrows - 1);
    const int my_req = a.req_of_tok[tok], my_pos = a.positions[tok];
    const bool row_ok = c < nrows;
    const int head = kvh * G + g;

    u32x4 Q[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        Q[s] = *reinterpret_cast<const u32x4*>(a.q + ((size_t)tok * a.q_heads + head) * 128 + 32 * s + 8 * qk);
    f32x4 O[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) O[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -1e30f, l = 0.f;       // finite sentinel: masked scores never enter an exp
    const float qs = a.sm_scale * PF_LOG2E * (KV == OMNI_KV_FP8 ? a.k_scale : 1.0f);

    // request segments of the tile: bit r set <=> row r starts a new request (wave-uniform)
    const int prev_req = __shfl_up(my_req, 1, 64);
    const uint32_t starts = (uint32_t)(__ballot(row_ok && qk == 0 && (c == 0 || my_req != prev_req)) & 0xFFFFull);

    int seg = 0;
    while (seg < nrows) {
        const uint32_t rest = starts >> (seg + 1);
        const int seg_end = rest ? seg + 1 + __builtin_ctz(rest) : nrows;
        const int rq = __shfl(my_req, seg, 64);
        const bool in_seg = row_ok && c >= seg && c < seg_end;
        int nkeys = in_seg ? my_pos + 1 : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) nkeys = max(nkeys, __shfl_xor(nkeys, o, 64));
        nkeys = __shfl(nkeys, 0, 64);                    // rows replicate over the 4 qk groups: lane 0 has the max
        const int32_t* bt = a.block_table + (size_t)rq * a.bt_stride;

        constexpr int CH = OMNI_KV_IS16(KV) ? 16 : 8;              // 16-B chunks per cache row
        constexpr int ITERS = (PF_BN * CH + NT - 1) / NT;          // (key, chunk) items per thread and tile
        u32x4 rk[ITERS], rv[ITERS];
        float rks[ITERS], rvs[ITERS];
        auto fetch = [&](int k0) {                                 // tile k0's cache bytes -> registers
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const int it = threadIdx.x + i * NT;
                if (PF_BN * CH % NT != 0 && it >= PF_BN * CH) continue;
                const int kk = it / CH, ch = it - kk * CH;
                const int key = min(k0 + kk, nkeys - 1);           // tail keys: valid address, masked below
                const size_t row = (((size_t)bt[key >> a.bs_shift] << a.bs_shift) + (key & (a.bs - 1))) * a.kv_heads + kvh;
                rk[i] = pf_load<KV>(a.k_cache, row, ch);
                rv[i] = pf_load<KV>(a.v_cache, row, ch);
                if (KV == OMNI_KV_INT8 && ch == 0) { rks[i] = a.k_scales[row]; rvs[i] = a.v_scales[row]; }
            }
        };
        if (nkeys > 0) fetch(0);
        for (int k0 = 0; k0 < nkeys; k0 += PF_BN) {
            __syncthreads();                             // the previous tile has been consumed by every wave
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const int it = threadIdx.x + i * NT;
                if (PF_BN * CH % NT != 0 && it >= PF_BN * CH) continue;
                const int kk = it / CH, ch = it - kk * CH;
                pf_store<KV>(sK + kk * PF_ROWB, rk[i], ch);
                pf_store<KV>(sV + kk * PF_ROWB, rv[i], ch);
                if (KV == OMNI_KV_INT8 && ch == 0) { sKs[kk] = rks[i]; sVs[kk] = rvs[i]; }
            }
            if (k0 + PF_BN < nkeys) fetch(k0 + PF_BN);   // the next tile travels while this one is computed on
            __syncthreads();

            // ---- S^T = K . Q^T for two 16-key tiles
            f32x4 S[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                S[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const u32x4 A = *reinterpret_cast<const u32x4*>(sK + (16 * kt + c) * PF_ROWB + (32 * s + 8 * qk) * 2);
                    S[kt] = pf_mfma(A, Q[s], S[kt]);
                }
            }
            // ---- online softmax of query row c over this lane's 8 keys (+ the 3 other qk groups)
            float sv[8];
            bool ok[8];
            float mx = -1e30f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kidx = (j < 4) ? 4 * qk + j : 16 + 4 * qk + (j - 4);
                float s = S[j >> 2][j & 3] * qs;
                if (KV == OMNI_KV_INT8) s *= sKs[kidx];
                ok[j] = in_seg && (k0 + kidx <= my_pos);
                sv[j] = s;
                if (ok[j]) mx = fmaxf(mx, s);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx);
            const float alpha = exp2f(m - mn);           // m == mn == sentinel -> 1, harmless (l = 0, O = 0)
            float p[8], rs = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                p[j] = ok[j] ? exp2f(sv[j] - mn) : 0.f;
                rs += p[j];
            }
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            l = l * alpha + rs;
            m = mn;
            if (KV == OMNI_KV_INT8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) p[j] *= sVs[(j < 4) ? 4 * qk + j : 16 + 4 * qk + (j - 4)];
            }
            // P as three bf16 terms hi + mid + lo = the fp32 value exactly (three MFMAs per d tile).  The oracle and the
            // decode kernel keep P in fp32; with a single bf16 rounding of P 2-ulp output flips appear on short
            // contexts, with two terms (2^-17) still 0.2 % of the outputs differ from the oracle by one ulp -- enough
            // to decorrelate every token's downstream roundings (scripts/diag_prefill_rows.py, diag_prefill_e2e.py).
            float pm[8], plo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float r1 = p[j] - bfround(p[j]);
                pm[j] = bfround(r1);
                plo[j] = r1 - pm[j];
            }
            const u32x4 P = (u32x4){pack_bf2(p[0], p[1]), pack_bf2(p[2], p[3]), pack_bf2(p[4], p[5]), pack_bf2(p[6], p[7])};
            const u32x4 PM = (u32x4){pack_bf2(pm[0], pm[1]), pack_bf2(pm[2], pm[3]), pack_bf2(pm[4], pm[5]), pack_bf2(pm[6], pm[7])};
            const u32x4 PL = (u32x4){pack_bf2(plo[0], plo[1]), pack_bf2(plo[2], plo[3]), pack_bf2(plo[4], plo[5]), pack_bf2(plo[6], plo[7])};
            // ---- O^T += V^T . P^T
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                uint32_t w[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j0 = 2 * jj, j1 = 2 * jj + 1;
                    const int r0 = (j0 < 4) ? 4 * qk + j0 : 16 + 4 * qk + (j0 - 4);
                    const int r1 = (j1 < 4) ? 4 * qk + j1 : 16 + 4 * qk + (j1 - 4);
                    const uint32_t lo = *reinterpret_cast<const uint16_t*>(sV + r0 * PF_ROWB + (16 * dt + c) * 2);
                    const uint32_t hi = *reinterpret_cast<const uint16_t*>(sV + r1 * PF_ROWB + (16 * dt + c) * 2);
                    w[jj] = lo | (hi << 16);
                }
                f32x4 acc = O[dt];
                acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
                const u32x4 VT = (u32x4){w[0], w[1], w[2], w[3]};
                O[dt] = pf_mfma(VT, P, pf_mfma(VT, PM, pf_mfma(VT, PL, acc)));
            }
        }
        seg = seg_end;
    }

    if (row_ok) {
        const float inv = (l > 0.f ? 1.0f / l : 0.f) * (KV == OMNI_KV_FP8 ? a.v_scale : 1.0f);
        const int K = a.q_heads * 128;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            const int col = head * 128 + 16 * dt + 4 * qk;
            uint16_t* dst = a.out_frag ? a.out + frag_off(tok, col, K) : a.out + (size_t)tok * K + col;
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf2(O[dt][0] * inv, O[dt][1] * inv), pack_bf2(O[dt][2] * inv, O[dt][3] * inv));
        }
    }
}

template <int KV>
static int pf_launch_g(const PFArgs& a, int G, hipStream_t st) {
    dim3 grid((a.T + 15) / 16, a.kv_heads);
    if (G == 1) hipLaunchKernelGGL((paged_attn_prefill_mfma_kernel<KV, 1>), grid, dim3(64), 0, st, a);
    else if (G == 2) hipLaunchKernelGGL((paged_attn_prefill_mfma_kernel<KV, 2>), grid, dim3(128), 0, st, a);
    else if (G == 4) hipLaunchKernelGGL((paged_attn_prefill_mfma_kernel<KV, 4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((paged_attn_prefill_mfma_kernel<KV, 8>), grid, dim3(512), 0, st, a);
    OMNI_CHECK_LAUNCH("omni_paged_attn_prefill(mfma)");
    return OMNI_OK;
}

OMNI_KNOB g_prefill_mfma = 1;
#ifdef OMNI_DEBUG_HOOKS
extern "C" void omni_debug_prefill_mfma(int on) { g_prefill_mfma = on; }   // A/B against the per-token VALU path
#endif

bool k_prefill_mfma_supported(int q_heads, int kv_heads, int head_dim) {
    if (!g_prefill_mfma || head_dim != 128 || kv_heads <= 0 || q_heads % kv_heads) return false;
    const int G = q_heads / kv_heads;
    return G == 1 || G == 2 || G == 4 || G == 8;
}

int k_prefill_mfma(const void* q, const void* k_cache, const void* v_cache, const float* k_scales, const float* v_scales,
                   const int32_t* block_table, int bt_stride, const int32_t* req_of_tok, const int32_t* positions, void* out,
                   int T, int q_heads, int kv_heads, int block_size, int kv_dtype, float k_scale, float v_scale, float sm_scale,
                   int out_frag, void* stream) {
    OMNI_CHECK_ARG(q && k_cache && v_cache && block_table && req_of_tok && positions && out, "omni_paged_attn_prefill: null pointer");
    OMNI_CHECK_ARG(block_size > 0 && (block_size & (block_size - 1)) == 0 && bt_stride > 0, "omni_paged_attn_prefill: block_size=%d (a power of two) / bt_stride", block_size);
    OMNI_CHECK_ARG(kv_dtype != OMNI_KV_INT8 || (k_scales && v_scales), "omni_paged_attn_prefill: int8 KV needs scales");
    if (T <= 0) return OMNI_OK;
    PFArgs a{};
    a.q = (const uint16_t*)q; a.k_cache = k_cache; a.v_cache = v_cache; a.k_scales = k_scales; a.v_scales = v_scales;
    a.block_table = block_table; a.bt_stride = bt_stride; a.req_of_tok = req_of_tok; a.positions = positions;
    a.out = (uint16_t*)out; a.T = T; a.q_heads = q_heads; a.kv_heads = kv_heads; a.bs = block_size; a.bs_shift = __builtin_ctz((unsigned)block_size);
    a.k_scale = k_scale; a.v_scale = v_scale; a.sm_scale = sm_scale; a.out_frag = out_frag;
    const int G = q_heads / kv_heads;
    hipStream_t st = (hipStream_t)stream;
    switch (kv_dtype) {
        case OMNI_KV_BF16: return pf_launch_g<OMNI_KV_BF16>(a, G, st);
        case OMNI_KV_FP8: return pf_launch_g<OMNI_KV_FP8>(a, G, st);
        case OMNI_KV_INT8: return pf_launch_g<OMNI_KV_INT8>(a, G, st);
        case OMNI_KV_FP16: return pf_launch_g<OMNI_KV_FP16>(a, G, st);
        default: omni_set_error("omni_paged_attn_prefill: kv_dtype %d", kv_dtype); return OMNI_EINVAL;
    }
}
