// Paged-attention decode body (query_len = 1; see paged_attn.hip for the design notes).
#pragma once
#include "attn_common.cuh"
#include "coherent.cuh"
#include "common.cuh"
#include "gemm_frag.cuh"
#include "kernels.h"

#define PA_THREADS 256
#define PA_WAVES 4
#ifndef PA_U
#define PA_U 4                 // 8-token groups per load batch per wave
#endif
#define PA_REC 130             // partial record: [0]=m (log2 domain) [1]=l [2..129]=acc (unnormalised)
// the pair's LDS scratch, in floats: [PA_WAVES][G][PA_REC] records | q scratch [PA_WAVES][G][128] | new K, V [2][128] | merge images
// [PA_WAVES][8 token groups][PA_MG_PITCH]
#define PA_MG_PITCH 132
#define PA_LDS_MERGE_OFF(G) (PA_WAVES * (G) * PA_REC + PA_WAVES * (G) * 128 + 256)
#define PA_LDS_FLOATS(G) (PA_LDS_MERGE_OFF(G) + PA_WAVES * 8 * PA_MG_PITCH)

struct PAArgs {
    const uint16_t* q;             // bf16 [rows, Hq*128]           (unfused)
    const uint16_t* qkv;           // bf16 [rows, (Hq+2Hkv)*128]    (fused)
    const uint16_t* qnorm_w; const uint16_t* knorm_w; const int32_t* positions; const int32_t* rope_delta; const uint16_t* cos_sin;
    int64_t* slot_out; float eps;
    void* k_cache; void* v_cache; float* k_scales; float* v_scales;
    const int32_t* block_table; int bt_stride; const int32_t* seq_lens; const int32_t* req_of_row; int seq_from_pos;
    uint16_t* out; float* partial;
    unsigned* merge_cnt;           // KV splits merged by the LAST ARRIVER of a (row, virtual kv head) (round 6): its arrival counters, or NULL = paged_attn_merge_kernel
    int q_heads, kv_heads, bs; float k_scale, v_scale, sm_scale; int nsplit;
    int bs_shift;                  // log2(bs): the paged kernels take power-of-two blocks (vLLM's 8 ... 256) -- token -> (block, offset)
                                   // by shift and mask; a runtime division cost every (token group, batch) ~20 quarter-rate instructions
    const float* scale_dev;        // fp8 KV: device {k_scale, v_scale} of this layer, read at run time instead of the two launch arguments
                                   // (calibrated scales reach captured graphs this way: omni_talker_set_kv_scales); NULL: the arguments
    int out_frag;                  // output in fragment-major layout (x operand of the o_proj GEMM, K = Hq*128)
    int kv_rep;                    // decode kernel: grid.x = kv_heads * kv_rep "virtual" kv heads of q_heads / (kv_heads * kv_rep)
                                   // q heads each (16 q / 2 kv heads run as 4 x 4); the K / V rows are those of vh / kv_rep
    int dense_pos;                 // attn_small DENSE: every row sits at this position of its own block (block = row)
    const int32_t* num_live;       // fused decode: rows >= *num_live (padding of a graph bucket) write no KV / slot (NULL: all live)
    int tail_chunks;               // 1: the LAST, partial 128-token round of a pair's history is dealt out as contiguous 32-token chunks, to waves
                                   // 3, 2, 1, 0 in that order, instead of interleaved 8-token groups over all four: wave 0 -- which also norms,
                                   // quantises, stores and folds the NEW token -- gets a batch only when the round is nearly full, and no wave
                                   // runs a batch that is mostly mask (round 5; another fold order of the online softmax, same arithmetic)
    int rope_rows;                 // rows of the cos / sin table (0: unknown): positions[b] + rope_delta[b] is clamped into it -- the host
                                   // refuses requests whose M-RoPE ids could leave the table (runner._update_states); this is the backstop
};

// row of the cos / sin table for batch row `row` at cache position `pos`
__device__ __forceinline__ int pa_rope_row(const PAArgs& a, int pos, int row) {
    int p = pos + (a.rope_delta ? a.rope_delta[row] : 0);
    if (a.rope_rows > 0) p = min(max(p, 0), a.rope_rows - 1);
    return p;
}

typedef float f32x2v __attribute__((ext_vector_type(2)));
// 2^x for x <= 0 (softmax weights and rescale factors): the bare v_exp_f32.  exp2f() wraps it in a range test and a rescale so that
// results below 2^-126 come out as denormals instead of 0 -- four more instructions per call for weights of 1e-38.
__device__ __forceinline__ float pa_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
template <int KV>
struct KVRaw { u32x4 a, b; };   // b: bf16 only (second 8 elements)

// cache row (= [block][offset][kv head]) of token t of a sequence whose block id is blk: 32-bit arithmetic -- 2^32 rows of >= 128 bytes
// are more than the 288 GB of HBM hold
__device__ __forceinline__ uint32_t pa_cache_row(const PAArgs& a, int blk, int t, int kvh) {
    return (((uint32_t)blk << a.bs_shift) | ((uint32_t)t & (uint32_t)(a.bs - 1))) * (uint32_t)a.kv_heads + (uint32_t)kvh;
}

// element index (0..127) of the e-th value (0..15) held by a lane with sub = lane & 7
template <int KV>
__device__ __forceinline__ int elem_of(int sub, int e) {
    if (OMNI_KV_IS16(KV)) return (e < 8) ? (sub * 8 + e) : (64 + sub * 8 + (e - 8));
    return sub * 16 + e;
}

// The cache pointers are GLOBAL-memory pointers.  hipcc infers that for pointers that come straight from the kernel arguments; the
// persistent backbone launch reads them from a device table (a flat pointer to the compiler -> flat_load, which counts in
// lgkmcnt too and returns out of order): say it explicitly.
#define PA_GLOBAL __attribute__((address_space(1)))
template <int KV>
__device__ __forceinline__ KVRaw<KV> load_row(const void* base, size_t row, int sub) {
    KVRaw<KV> r;
    if (OMNI_KV_IS16(KV)) {
        const PA_GLOBAL uint16_t* p = (const PA_GLOBAL uint16_t*)base + row * 128;
        r.a = __builtin_nontemporal_load((const PA_GLOBAL u32x4*)(p + sub * 8));
        r.b = __builtin_nontemporal_load((const PA_GLOBAL u32x4*)(p + 64 + sub * 8));
    } else {
        const PA_GLOBAL uint8_t* p = (const PA_GLOBAL uint8_t*)base + row * 128;
        r.a = __builtin_nontemporal_load((const PA_GLOBAL u32x4*)(p + sub * 16));
        r.b = (u32x4){0, 0, 0, 0};
    }
    return r;
}

template <int KV>
__device__ __forceinline__ void to_f32(const KVRaw<KV>& r, float* f) {
    if (KV == OMNI_KV_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j] = bf_lo(r.a[j]);
            f[2 * j + 1] = bf_hi(r.a[j]);
            f[8 + 2 * j] = bf_lo(r.b[j]);
            f[8 + 2 * j + 1] = bf_hi(r.b[j]);
        }
    } else if (KV == OMNI_KV_FP16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j] = h_lo(r.a[j]);
            f[2 * j + 1] = h_hi(r.a[j]);
            f[8 + 2 * j] = h_lo(r.b[j]);
            f[8 + 2 * j + 1] = h_hi(r.b[j]);
        }
    } else if (KV == OMNI_KV_FP8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) unpack_fp8x4(r.a[j], f + 4 * j);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = r.a[j] ^ 0x80808080u;       // signed byte + 128 as an unsigned byte: v_cvt_f32_ubyteN, then - 128
            f[4 * j + 0] = (float)(w & 0xFF) - 128.0f;
            f[4 * j + 1] = (float)((w >> 8) & 0xFF) - 128.0f;
            f[4 * j + 2] = (float)((w >> 16) & 0xFF) - 128.0f;
            f[4 * j + 3] = (float)(w >> 24) - 128.0f;
        }
    }
}

// The decode body, shared by the stand-alone launch (paged_attn.hip) and the attention stage of the persistent backbone launch
// (bb_all.hip, CHAIN): one arithmetic, the same bits in both.  A "pair" = (row, virtual kv head[, KV split]) is worked by 4 waves
// (`wave` 0..3, `tid` 0..255 inside the pair); `lds` = the pair's own scratch ([PA_WAVES][G][PA_REC] | q scratch | kv scratch).
// CHAIN: `active` = false for a pair past the batch (the waves only keep the workgroup's barriers); batch 0 of the K / V history
// goes out BEFORE the stage's flags (history is final since an earlier launch), the qkv rows -- written by another workgroup of
// this launch -- are read behind them with sc1 loads, the attention output leaves with sc1 stores (coherent.cuh).
// ---- batch 0 of a pair's K / V history issued AHEAD of the attention stage (bb_all.hip: behind the epilogue stores of the qkv
// stage, so the rows are in flight through that stage's flag, the hand-off and this stage's poll).  The block ids of batch 0
// depend on the block table only: loaded once per launch.  Same addresses as PA_LOAD at T0 = wave * 8 without a token limit.
template <int KV>
struct PaPre {
    KVRaw<KV> k[PA_U], v[PA_U];
    float ks[PA_U], vs[PA_U];
};
#define PA_PRE_LOADS(KV) (OMNI_KV_IS16(KV) ? 4 * PA_U : ((KV) == OMNI_KV_INT8 ? 4 * PA_U : 2 * PA_U))      // vector-memory loads per wave
__device__ __forceinline__ void pa_pre_blocks(const PAArgs& a, int row, int wave, int (&blk)[PA_U]) {
    const int tg = (threadIdx.x & 63) >> 3;
    const int32_t* bt = a.block_table + (size_t)row * a.bt_stride;
#pragma unroll
    for (int u = 0; u < PA_U; ++u) blk[u] = bt[min((wave * 8 + u * PA_WAVES * 8 + tg) >> a.bs_shift, a.bt_stride - 1)];
}
template <int KV>
__device__ __forceinline__ void pa_pre_issue(const PAArgs& a, int kvh, int wave, const int (&blk)[PA_U], PaPre<KV>& pre) {
    const int lane = threadIdx.x & 63, sub = lane & 7, tg = lane >> 3;
#pragma unroll
    for (int u = 0; u < PA_U; ++u) {
        const int t_ = wave * 8 + u * PA_WAVES * 8 + tg;
        const size_t r_ = pa_cache_row(a, blk[u], t_, kvh);
        pre.k[u] = load_row<KV>(a.k_cache, r_, sub);
        pre.v[u] = load_row<KV>(a.v_cache, r_, sub);
        if (KV == OMNI_KV_INT8) { pre.ks[u] = a.k_scales[r_]; pre.vs[u] = a.v_scales[r_]; }
    }
}

#ifdef OMNI_DEBUG_HOOKS      // timeline stamps of the attention stage (slot 1 of the per-layer stamp block of bb_all.hip; 100 MHz counter)
#define PA_STAMP(k)                                                                                                  \
    do {                                                                                                             \
        if (CHAIN && gate->stamps != nullptr && threadIdx.x == 0)                                                    \
            gate->stamps[((size_t)1 * 8 + (k)) * OMNI_CHAIN_WGS + blockIdx.x] = __builtin_amdgcn_s_memrealtime();    \
    } while (0)
#else
#define PA_STAMP(k) do { } while (0)
#endif

template <int KV, int G, bool FUSED, bool CHAIN>
__device__ __forceinline__ void pa_decode_body(const PAArgs& a, float* lds, const int vh, const int row_in, const int sp, const int wave,
                                               const int tid, const bool active, ChainGate* gate, const int code,
                                               const PaPre<KV>* pre = nullptr /* CHAIN: batch 0 already issued (pa_pre_issue) */) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & 7, tg = lane >> 3;
    const int row = active ? row_in : 0;
    const float k_scale = a.scale_dev ? a.scale_dev[0] : a.k_scale, v_scale = a.scale_dev ? a.scale_dev[1] : a.v_scale;
    const int kvh = vh / a.kv_rep;
    const int req = a.req_of_row ? a.req_of_row[row] : row;
    const int kv_heads = a.kv_heads, bs = a.bs;
    const int32_t* bt = a.block_table + (size_t)req * a.bt_stride;
    const int max_blk = a.bt_stride - 1;

    // ---- batch 0 of K/V loads goes out before anything else is known (speculative: tokens past the
    // sequence end resolve to whatever block the table holds there -- block 0, the null block)
    // token groups per load batch: 4 (PA_U), or 2 where four would push the instantiation past 256 registers and down to ONE wave per
    // SIMD -- the 16-bit caches (a row is two 16-byte loads per lane) with two or more heads, and every four-head instantiation
    constexpr int U = (!CHAIN && ((OMNI_KV_IS16(KV) && G >= 2) || G >= 4)) ? 2 : PA_U;
    KVRaw<KV> k0[U], v0[U], k1[U], v1[U];
    float ks0[U], vs0[U], ks1[U], vs1[U];
    // token groups past the end of the history re-read its LAST row (one hot line) instead of whatever block the table
    // holds further on: with blocks allocated ahead of the sequence those were real, cold HBM rows -- up to 127 tokens of
    // wasted traffic per (row, head).  The speculative batch 0 goes out before the length is known and stays unclamped.
    int t_lim = 0x7FFFFFFF;
#define PA_LOAD(KR, VR, KS, VS, T0, STR)                                                     \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                       \
        const int t_ = min((T0) + u * (STR) + tg, t_lim);                                    \
        const int bi_ = min(t_ >> a.bs_shift, max_blk);                                      \
        const size_t r_ = pa_cache_row(a, bt[bi_], t_, kvh);                                 \
        KR[u] = load_row<KV>(a.k_cache, r_, sub);                                            \
        VR[u] = load_row<KV>(a.v_cache, r_, sub);                                            \
        if (KV == OMNI_KV_INT8) { KS[u] = a.k_scales[r_]; VS[u] = a.v_scales[r_]; }          \
    }
    const bool spec = (a.nsplit == 1);
    PA_STAMP(0);
    if (pre != nullptr) {
#pragma unroll
        for (int u = 0; u < U; ++u) { k0[u] = pre->k[u]; v0[u] = pre->v[u]; ks0[u] = pre->ks[u]; vs0[u] = pre->vs[u]; }
    } else if (spec && active) { PA_LOAD(k0, v0, ks0, vs0, wave * 8, PA_WAVES * 8) }
    PA_STAMP(1);
    if (CHAIN) chain_gate_wait(*gate, code);                  // the qkv stage's flags; workgroup barrier inside
    PA_STAMP(2);
    if (active) {

    const int seq_len = a.seq_lens[row] + (a.seq_from_pos ? 1 : 0);
    int per = (seq_len + a.nsplit - 1) / a.nsplit;
    per = (per + 31) & ~31;                       // 8-token groups never straddle splits
    const int t_begin = sp * per;
    const int cur = seq_len - 1;                  // the token computed this step
    const int t_end = min(FUSED ? cur : seq_len, t_begin + per);
    t_lim = max(t_end - 1, 0);
    // batches of this wave: batch b of a full round = four 8-token groups 32 apart (T0 = round start + 8 wave, group stride 32); the
    // tail round (PAArgs::tail_chunks; never round 0, whose loads went out before the length was known) = one contiguous 32-token
    // chunk per wave, waves 3, 2, 1, 0 in that order (T0 = round start + 32 (3 - wave), group stride 8)
    constexpr int ROUND = PA_WAVES * 8 * U;
    const int n_hist = max(t_end - t_begin, 0);
    const int full = n_hist / ROUND, rem = n_hist - full * ROUND;
    const bool tail = !CHAIN && a.tail_chunks && full >= 1 && rem > 0;
    const int nb = tail ? full + ((PA_WAVES - 1 - wave) * (8 * U) < rem ? 1 : 0)
                        : (n_hist - wave * 8 + ROUND - 1 > 0 ? (max(n_hist - wave * 8, 0) + ROUND - 1) / ROUND : 0);
    auto b_t0 = [&](int b) { return (tail && b == full) ? t_begin + full * ROUND + (PA_WAVES - 1 - wave) * (8 * U) : t_begin + wave * 8 + b * ROUND; };
    auto b_str = [&](int b) { return (tail && b == full) ? 8 : PA_WAVES * 8; };
    if (!spec) { PA_LOAD(k0, v0, ks0, vs0, t_begin + wave * 8, PA_WAVES * 8) }

    // ---- q for the G heads of this kv head, pre-scaled into the log2 domain, as PAIRS OF ADJACENT DIMENSIONS: a head's QK product is 8
    // v_pk_fma_f32 on {k[2i], k[2i+1]} x {q[2i], q[2i+1]} (even dims in the low half, odd in the high half, one add at the end) -- both
    // operands are natural register pairs (K straight out of v_cvt_pk_f32_fp8, q out of a 16-byte LDS read): NO op_sel.
    // Why not two heads per FMA with the K element broadcast by op_sel (tried first: the scalar loop's bits, same count): hipcc read
    // the odd K elements as "high dword of src1 into the low lane" (op_sel:[0,1,0]).  On this GPU that selector -- and only that one, for
    // v_pk_fma_f32 and v_pk_mul_f32 alike -- returns wrong results WHILE ANOTHER WAVE ON THE SIMD ISSUES MFMAs (5 % of the executions
    // beside a pure MFMA loop; ~4 per million beside the vocoder's GEMMs; never beside loads, LDS traffic, LDS-DMA or VALU-only
    // kernels: scripts/probes/pkfma_src1.hip + aggressor.hip, profiles/r04_pkfma_probe.txt).  This kernel has no MFMA and a stream runs
    // one kernel at a time, so the attention was parity-green and bit-stable alone -- and gave garbage in one head of a pair in ~10 %
    // of its launches beside the Code2Wav process (tests/test_gpu_colocation.py caught it; NOTEBOOK "Round 4").
    // tests/test_build_rules.py keeps op_sel[1] = 1 out of every kernel.
    f32x2v qf[G][8];
    const float qs = a.sm_scale * LOG2E * (KV == OMNI_KV_FP8 ? k_scale : 1.0f);
    const int nslots = a.q_heads + 2 * kv_heads;
    float* wq = lds + PA_WAVES * G * PA_REC + wave * (G * 128);
    // CHAIN: this row's q heads (and, in the wave that folds the new token, its k / v head) as one dword per lane each, all in one
    // round trip behind the flags; the (l, l + 64) pairing of the norm + RoPE is restored through LDS
    uint32_t kw_new = 0u, vw_new = 0u;
    if (FUSED) {
        const int pos = a.positions[row];
        const uint16_t* cs = a.cos_sin + (size_t)pa_rope_row(a, pos, row) * 128;
        if (CHAIN) {
            const coh_rsrc_t qrs = coh_rsrc(a.qkv);
            uint32_t qw[G];
#pragma unroll
            for (int g = 0; g < G; ++g) qw[g] = coh_ld4(qrs, (uint32_t)(((size_t)row * nslots + vh * G + g) * 128 + 2 * lane) * 2);
            if (wave == 0) {
                kw_new = coh_ld4(qrs, (uint32_t)(((size_t)row * nslots + a.q_heads + kvh) * 128 + 2 * lane) * 2);
                vw_new = coh_ld4(qrs, (uint32_t)(((size_t)row * nslots + a.q_heads + kv_heads + kvh) * 128 + 2 * lane) * 2);
            }
            uint32_t* raw = reinterpret_cast<uint32_t*>(wq);
#pragma unroll
            for (int g = 0; g < G; ++g) raw[g * 64 + lane] = qw[g];
            __builtin_amdgcn_wave_barrier();
            const uint16_t* rq = reinterpret_cast<const uint16_t*>(raw);
            float x0[G], x1[G];
#pragma unroll
            for (int g = 0; g < G; ++g) { x0[g] = bf2f(rq[g * 128 + lane]); x1[g] = bf2f(rq[g * 128 + 64 + lane]); }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float y0, y1;
                head_norm_rope_vals(x0[g], x1[g], a.qnorm_w, cs, a.eps, lane, y0, y1);
                wq[g * 128 + lane] = y0 * qs;          // (scaled here, one scalar multiply each: the read side stays free of packed
                wq[g * 128 + 64 + lane] = y1 * qs;     //  multiplies whose destination would overlap their broadcast operand)
            }
        } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float y0, y1;
            head_norm_rope(a.qkv + ((size_t)row * nslots + vh * G + g) * 128, a.qnorm_w, cs, a.eps, lane, y0, y1);
            wq[g * 128 + lane] = y0 * qs;
            wq[g * 128 + 64 + lane] = y1 * qs;
        }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int e = 0; e < 16; ++e) qf[g][e >> 1][e & 1] = wq[g * 128 + elem_of<KV>(sub, e)];
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint16_t* qp = a.q + ((size_t)row * a.q_heads + vh * G + g) * 128;
#pragma unroll
            for (int e = 0; e < 16; ++e) qf[g][e >> 1][e & 1] = bf2f(qp[elem_of<KV>(sub, e)]) * qs;
        }
    }
    float m[G], l[G], acc[G][16];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
    }

    // ---- fused: the new token's K/V (wave 0 of the split that owns position `cur`) -- BEFORE the history loop: its loads,
    // norm, quantisation and stores overlap the K/V batches already in flight instead of forming a tail after the loop; the
    // running softmax state of token-group 0 simply starts from this token
    if (FUSED) {
        const int sp_cur = cur / per;
        if (wave == 0 && sp == sp_cur) {
            const int pos = a.positions[row];
            const uint16_t* cs = a.cos_sin + (size_t)pa_rope_row(a, pos, row) * 128;
            const int64_t slot = ((int64_t)bt[min(pos >> a.bs_shift, max_blk)] << a.bs_shift) + (pos & (bs - 1));
            // rows of a padded graph bucket past the live count may be live PREFILL rows of the persistent batch: they
            // compute (results discarded) but leave the cache and the slot record alone
            const bool live = !a.num_live || row < *a.num_live;
            if (vh == 0 && lane == 0 && a.slot_out && live) a.slot_out[row] = slot;
            const size_t crow = (size_t)slot * kv_heads + kvh;
            const bool kv_writer = live && vh % a.kv_rep == 0;      // the other groups of this kv head fold the same values
            float* kvs = lds + PA_WAVES * G * PA_REC + PA_WAVES * (G * 128);   // [2][128] dequantised new K, V
            float kx0, kx1, vx0, vx1;
            if (CHAIN) {
                uint32_t* raw = reinterpret_cast<uint32_t*>(kvs);
                raw[lane] = kw_new;
                raw[64 + lane] = vw_new;
                __builtin_amdgcn_wave_barrier();
                const uint16_t* rk = reinterpret_cast<const uint16_t*>(raw);
                const float rk0 = bf2f(rk[lane]), rk1 = bf2f(rk[lane + 64]);
                vx0 = bf2f(rk[128 + lane]); vx1 = bf2f(rk[128 + lane + 64]);
                __builtin_amdgcn_wave_barrier();
                head_norm_rope_vals(rk0, rk1, a.knorm_w, cs, a.eps, lane, kx0, kx1);
            } else {
                head_norm_rope(a.qkv + ((size_t)row * nslots + a.q_heads + kvh) * 128, a.knorm_w, cs, a.eps, lane, kx0, kx1);
                const uint16_t* vsrc = a.qkv + ((size_t)row * nslots + a.q_heads + kv_heads + kvh) * 128;
                vx0 = bf2f(vsrc[lane]); vx1 = bf2f(vsrc[lane + 64]);
            }
            float ksc_new = 1.f, vsc_new = 1.f;
            if (KV == OMNI_KV_BF16) {
                uint16_t* kd = reinterpret_cast<uint16_t*>(a.k_cache) + crow * 128;
                uint16_t* vd = reinterpret_cast<uint16_t*>(a.v_cache) + crow * 128;
                if (kv_writer) {
                    kd[lane] = f2bf(kx0); kd[lane + 64] = f2bf(kx1);
                    vd[lane] = f2bf(vx0); vd[lane + 64] = f2bf(vx1);
                }
            } else if (KV == OMNI_KV_FP16) {
                uint16_t* kd = reinterpret_cast<uint16_t*>(a.k_cache) + crow * 128;
                uint16_t* vd = reinterpret_cast<uint16_t*>(a.v_cache) + crow * 128;
                const uint16_t hk0 = f2h(kx0), hk1 = f2h(kx1), hv0 = f2h(vx0), hv1 = f2h(vx1);
                if (kv_writer) {
                    kd[lane] = hk0; kd[lane + 64] = hk1;
                    vd[lane] = hv0; vd[lane + 64] = hv1;
                }
                kx0 = h2f(hk0); kx1 = h2f(hk1); vx0 = h2f(hv0); vx1 = h2f(hv1);     // what the cache now holds
            } else if (KV == OMNI_KV_FP8) {
                const float ik = k_scale, iv = v_scale;
                const uint32_t pk = pack_fp8x4(ik == 1.f ? kx0 : kx0 / ik, ik == 1.f ? kx1 : kx1 / ik, 0.f, 0.f);
                const uint32_t pv = pack_fp8x4(iv == 1.f ? vx0 : vx0 / iv, iv == 1.f ? vx1 : vx1 / iv, 0.f, 0.f);
                uint8_t* kd = reinterpret_cast<uint8_t*>(a.k_cache) + crow * 128;
                uint8_t* vd = reinterpret_cast<uint8_t*>(a.v_cache) + crow * 128;
                if (kv_writer) {
                    kd[lane] = (uint8_t)(pk & 0xFF); kd[lane + 64] = (uint8_t)((pk >> 8) & 0xFF);
                    vd[lane] = (uint8_t)(pv & 0xFF); vd[lane + 64] = (uint8_t)((pv >> 8) & 0xFF);
                }
                float t4[4];
                unpack_fp8x4(pk, t4); kx0 = t4[0]; kx1 = t4[1];       // what the cache now holds (unscaled)
                unpack_fp8x4(pv, t4); vx0 = t4[0]; vx1 = t4[1];
            } else {
                const float ka = fmaxf(wave_max(fmaxf(fabsf(kx0), fabsf(kx1))), 1e-8f);
                const float va = fmaxf(wave_max(fmaxf(fabsf(vx0), fabsf(vx1))), 1e-8f);
                ksc_new = ka / 127.0f; vsc_new = va / 127.0f;
                kx0 = fminf(fmaxf(rintf(kx0 / ksc_new), -127.f), 127.f); kx1 = fminf(fmaxf(rintf(kx1 / ksc_new), -127.f), 127.f);
                vx0 = fminf(fmaxf(rintf(vx0 / vsc_new), -127.f), 127.f); vx1 = fminf(fmaxf(rintf(vx1 / vsc_new), -127.f), 127.f);
                int8_t* kd = reinterpret_cast<int8_t*>(a.k_cache) + crow * 128;
                int8_t* vd = reinterpret_cast<int8_t*>(a.v_cache) + crow * 128;
                if (kv_writer) {
                    kd[lane] = (int8_t)kx0; kd[lane + 64] = (int8_t)kx1;
                    vd[lane] = (int8_t)vx0; vd[lane + 64] = (int8_t)vx1;
                    if (lane == 0) { a.k_scales[crow] = ksc_new; a.v_scales[crow] = vsc_new; }
                }
            }
            kvs[lane] = kx0; kvs[64 + lane] = kx1;
            kvs[128 + lane] = vx0; kvs[192 + lane] = vx1;
            // token-group 0 of this wave folds the new token into its running softmax state
            float kf[16], vf[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                kf[e] = kvs[elem_of<KV>(sub, e)];
                vf[e] = kvs[128 + elem_of<KV>(sub, e)];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) d = fmaf(qf[g][e >> 1][e & 1], kf[e], d);
                d = group8_sum(d);
                if (KV == OMNI_KV_INT8) d *= ksc_new;
                if (tg == 0) {
                    const float mn = fmaxf(m[g], d);
                    const float corr = (m[g] == -INFINITY) ? 0.f : pa_exp2(m[g] - mn);
                    float p = pa_exp2(d - mn);
                    m[g] = mn;
                    l[g] = l[g] * corr + p;
                    if (KV == OMNI_KV_INT8) p *= vsc_new;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[g][e] = fmaf(p, vf[e], acc[g][e] * corr);
                }
            }
        }
    }

#define PA_COMPUTE(KR, VR, KS, VS, T0, STR)                                                          \
    {                                                                                                \
        float s_[U][G], mx_[G];                                                                   \
        bool ok_[U];                                                                              \
        _Pragma("unroll") for (int g = 0; g < G; ++g) mx_[g] = m[g];                                 \
        float r_[U][G];                                                                           \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                           \
            ok_[u] = (T0) + u * (STR) + tg < t_end;                                                  \
            float kf_[16];                                                                           \
            to_f32<KV>(KR[u], kf_);                                                                  \
            _Pragma("unroll") for (int g = 0; g < G; ++g) {                                          \
                f32x2v dd_ = (f32x2v){0.f, 0.f};                                                     \
                _Pragma("unroll") for (int i = 0; i < 8; ++i)                                        \
                    dd_ = __builtin_elementwise_fma(qf[g][i], (f32x2v){kf_[2 * i], kf_[2 * i + 1]}, dd_); \
                r_[u][g] = dd_[0] + dd_[1];                                                          \
            }                                                                                        \
        }                                                                                            \
        /* the token's 8 lanes: 3 DPP steps, each over all U * G sums (a DPP read 2 wait states behind its producer: */ \
        /* step by step the other sums fill them) -- group8_sum's order of additions */              \
        _Pragma("unroll") for (int u = 0; u < U; ++u)                                             \
            _Pragma("unroll") for (int g = 0; g < G; ++g) r_[u][g] += dpp_f<OMNI_DPP_XOR1>(r_[u][g]); \
        _Pragma("unroll") for (int u = 0; u < U; ++u)                                             \
            _Pragma("unroll") for (int g = 0; g < G; ++g) r_[u][g] += dpp_f<OMNI_DPP_XOR2>(r_[u][g]); \
        _Pragma("unroll") for (int u = 0; u < U; ++u)                                             \
            _Pragma("unroll") for (int g = 0; g < G; ++g) r_[u][g] += dpp_f<OMNI_DPP_HALF_MIRROR>(r_[u][g]); \
        _Pragma("unroll") for (int u = 0; u < U; ++u)                                             \
            _Pragma("unroll") for (int g = 0; g < G; ++g) {                                          \
                float d_ = r_[u][g];                                                                 \
                if (KV == OMNI_KV_INT8) d_ *= KS[u];                                                 \
                d_ = ok_[u] ? d_ : -INFINITY;                                                        \
                s_[u][g] = d_;                                                                       \
                mx_[g] = fmaxf(mx_[g], d_);                                                          \
            }                                                                                        \
        _Pragma("unroll") for (int g = 0; g < G; ++g) {                                              \
            const float corr_ = (mx_[g] == -INFINITY) ? 1.0f : pa_exp2(m[g] - mx_[g]);                 \
            m[g] = mx_[g];                                                                           \
            l[g] *= corr_;                                                                           \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[g][e] *= corr_;                       \
        }                                                                                            \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                           \
            float vf_[16];                                                                           \
            to_f32<KV>(VR[u], vf_);                                                                  \
            _Pragma("unroll") for (int g = 0; g < G; ++g) {                                          \
                float p_ = ok_[u] ? pa_exp2(s_[u][g] - m[g]) : 0.f;                                    \
                l[g] += p_;                                                                          \
                if (KV == OMNI_KV_INT8) p_ *= VS[u];                                                 \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[g][e] = fmaf(p_, vf_[e], acc[g][e]); \
            }                                                                                        \
        }                                                                                            \
    }

    // ---- main loop, loads one batch ahead
    for (int b = 0; b < nb;) {
        if (b + 1 < nb) { PA_LOAD(k1, v1, ks1, vs1, b_t0(b + 1), b_str(b + 1)) }
        PA_COMPUTE(k0, v0, ks0, vs0, b_t0(b), b_str(b))
        if (++b >= nb) break;
        if (b + 1 < nb) { PA_LOAD(k0, v0, ks0, vs0, b_t0(b + 1), b_str(b + 1)) }
        PA_COMPUTE(k1, v1, ks1, vs1, b_t0(b), b_str(b))
        ++b;
    }
#undef PA_LOAD
#undef PA_COMPUTE
    PA_STAMP(3);

    // ---- combine: the 8 token-groups of a wave merge in registers (xor-shuffles over lane bits 3..5),
    // then the 4 wave partials go through LDS
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float mw = m[g];
        mw = fmaxf(mw, dpp_f<OMNI_DPP_ROR8>(mw));
        mw = xor32_max(xor16_max(mw));
        const float sc = (m[g] == -INFINITY) ? 0.f : pa_exp2(m[g] - mw);
        float lw = l[g] * sc;
        lw += dpp_f<OMNI_DPP_ROR8>(lw);
        lw = xor32_sum(xor16_sum(lw));
        // the 16 partial outputs of the 8 token groups meet in this wave's LDS image [token group][128 + 4 (pad: the groups' 16-byte
        // stores on different banks)]: four 16-byte stores per lane, then lane L sums elements 2 L, 2 L + 1 over the groups as the
        // balanced tree ((0 + 1) + (2 + 3)) + ((4 + 5) + (6 + 7)) -- the order of the lane ^ 8 / ^ 16 / ^ 32 butterfly this replaces
        // (~200 DPP / permlane / select instructions per head), so the bits are unchanged.  A wave's LDS operations execute in order.
        float* mg = lds + PA_LDS_MERGE_OFF(G) + wave * (8 * PA_MG_PITCH);
        __builtin_amdgcn_wave_barrier();                                     // the previous head's reads are issued
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 v4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = acc[g][4 * c + e] * sc;
            *reinterpret_cast<f32x4*>(mg + tg * PA_MG_PITCH + elem_of<KV>(sub, 4 * c)) = v4;
        }
        __builtin_amdgcn_wave_barrier();
        f32x2v t8[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) t8[t] = *reinterpret_cast<const f32x2v*>(mg + t * PA_MG_PITCH + 2 * lane);
        const f32x2v o2 = ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
        float* rec = lds + ((size_t)wave * G + g) * PA_REC;
        *reinterpret_cast<f32x2v*>(rec + 2 + 2 * lane) = o2;
        if (lane == 0) {
            rec[0] = mw;
            rec[1] = lw;
        }
    }
    }   // active
    PA_STAMP(4);
    __syncthreads();
    PA_STAMP(5);
    if (active) {
        // one output element: merge of the 4 wave partials (fixed order)
        auto merged = [&](int g, int d, float& M, float& L) -> float {
            M = -INFINITY;
#pragma unroll
            for (int p = 0; p < PA_WAVES; ++p) M = fmaxf(M, lds[((size_t)p * G + g) * PA_REC]);
            L = 0.f;
            float A = 0.f;
#pragma unroll
            for (int p = 0; p < PA_WAVES; ++p) {
                const float* rec = lds + ((size_t)p * G + g) * PA_REC;
                const float w = (rec[0] == -INFINITY) ? 0.f : pa_exp2(rec[0] - M);
                L = fmaf(rec[1], w, L);
                A = fmaf(rec[2 + d], w, A);
            }
            return A;
        };
        if (CHAIN) {
            // write-through dword stores (two adjacent elements per thread): the o_proj stage reads them behind this stage's flag
            const coh_rsrc_t ors = coh_rsrc(a.out);
            const float vs = (KV == OMNI_KV_FP8) ? v_scale : 1.0f;
            for (int it = tid; it < G * 64; it += PA_THREADS) {
                const int g = it >> 6, d = 2 * (it & 63);
                float M, L, M1, L1;
                const float A0 = merged(g, d, M, L), A1 = merged(g, d + 1, M1, L1);
                const int qh = vh * G + g;
                const size_t oo = a.out_frag ? frag_off(row, qh * 128 + d, a.q_heads * 128) : ((size_t)row * a.q_heads + qh) * 128 + d;
                coh_st4(ors, (uint32_t)oo * 2, pack_bf2(L > 0.f ? (A0 / L) * vs : 0.f, L1 > 0.f ? (A1 / L1) * vs : 0.f));
            }
        } else
        for (int it = tid; it < G * 128; it += PA_THREADS) {
            const int g = it >> 7, d = it & 127;
            float M, L;
            const float A = merged(g, d, M, L);
            const int qh = vh * G + g;
            if (a.nsplit == 1) {
                const float vs = (KV == OMNI_KV_FP8) ? v_scale : 1.0f;
                const size_t oo = a.out_frag ? frag_off(row, qh * 128 + d, a.q_heads * 128) : ((size_t)row * a.q_heads + qh) * 128 + d;
                a.out[oo] = f2bf(L > 0.f ? (A / L) * vs : 0.f);
            } else if (a.merge_cnt != nullptr) {
                // write-through stores ("coherent by access", coherent.cuh): the merging workgroup may sit on another XCD
                const coh_rsrc_t prs = coh_rsrc(a.partial);
                const uint32_t ro = (uint32_t)((((size_t)row * a.q_heads + qh) * a.nsplit + sp) * PA_REC) * 4;
                if (d == 0) {
                    coh_st4(prs, ro, __float_as_uint(M));
                    coh_st4(prs, ro + 4, __float_as_uint(L));
                }
                coh_st4(prs, ro + (2 + d) * 4, __float_as_uint(A));
            } else {
                float* rec = a.partial + (((size_t)row * a.q_heads + qh) * a.nsplit + sp) * PA_REC;
                if (d == 0) {
                    rec[0] = M;
                    rec[1] = L;
                }
                rec[2 + d] = A;
            }
        }
    }
    if (!CHAIN && a.nsplit > 1 && a.merge_cnt != nullptr && active) {          // (workgroup-uniform)
        // ---- round 6: the splits of this (row, virtual kv head) are merged by whichever of their workgroups arrives LAST -- no merge launch
        // (5 us + a boundary per layer at 1-32 rows).  Records out (every wave drains its own stores), barrier, one device-scope ticket; the
        // last arriver resets the counter for the next launch and merges the records in SPLIT order (paged_attn_merge_kernel's arithmetic:
        // the result does not depend on who arrives last), reading them with sc1 loads.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                      // (also: every thread is done with the wave partials in LDS)
        int* s_last = reinterpret_cast<int*>(lds);
        if (tid == 0) {
            unsigned* cnt = a.merge_cnt + (size_t)row * (a.kv_heads * a.kv_rep) + vh;
            const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = t == (unsigned)a.nsplit - 1u;
            if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_last = last;
        }
        __syncthreads();
        if (*s_last) {
            const coh_rsrc_t prs = coh_rsrc(a.partial);
            const float vs = (KV == OMNI_KV_FP8) ? v_scale : 1.0f;
            for (int it = tid; it < G * 128; it += PA_THREADS) {
                const int g = it >> 7, d = it & 127;
                const int qh = vh * G + g;
                const uint32_t base = (uint32_t)((((size_t)row * a.q_heads + qh) * a.nsplit) * PA_REC) * 4;
                float M = -INFINITY;
                for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, coh_ldf(prs, base + (uint32_t)s * (PA_REC * 4)));
                float L = 0.f, A = 0.f;
                for (int s = 0; s < a.nsplit; ++s) {
                    const uint32_t ro = base + (uint32_t)s * (PA_REC * 4);
                    const float m = coh_ldf(prs, ro);
                    const float w = (m == -INFINITY) ? 0.f : exp2f(m - M);
                    L = fmaf(w, coh_ldf(prs, ro + 4), L);           // (the shared factor FIRST: see the note on packed FMAs above)
                    A = fmaf(w, coh_ldf(prs, ro + (2 + d) * 4), A);
                }
                const size_t oo = a.out_frag ? frag_off(row, qh * 128 + d, a.q_heads * 128) : ((size_t)row * a.q_heads + qh) * 128 + d;
                a.out[oo] = f2bf(L > 0.f ? (A / L) * vs : 0.f);
            }
        }
    }
    PA_STAMP(6);
    if (CHAIN) chain_gate_arrive(*gate);
    PA_STAMP(7);
}
