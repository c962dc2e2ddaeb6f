// Paged-attention decode (query_len = 1) over bf16 / fp8-e4m3fn / int8 KV blocks, head_dim 128,
// optionally FUSED with the per-head q/k RMSNorm + neox RoPE + KV-cache write of the new token
// (one launch instead of two per layer; the workgroup that owns a (row, kv-head) computes that
// head's new K/V itself, so no cross-workgroup dependency is created).
//
// HBM-bound: every K and V byte of the context is read exactly once with non-temporal 16-B loads,
// each 128-B (fp8/int8) or 256-B (bf16) token row fully consumed.  Workgroup = 4 waves = one
// (row, kv-head[, KV split]); a wave walks 8-token groups (8 lanes per token, 16 elements per
// lane), loads software-pipelined one batch of PA_U groups ahead (the first batch is issued before
// the sequence length is known: unallocated block-table entries point at the null block).  The
// G = Hq/Hkv query heads share every K/V load.  Dequant in registers, fp32 scores, online softmax
// per 8-lane group (no cross-wave traffic in the loop), one LDS combine at the end; optional KV
// splits merged by a second tiny kernel.  Output = oracle attention_rows (fp32 softmax and PV, one
// rounding to bf16); the fused prologue = oracle rms_norm / apply_rope / fp8_quant / int8_quant.
#include "common.cuh"
#include "kernels.h"
#include "attn_common.cuh"
#include "gemm_frag.cuh"

#include "pa_body.cuh"

template <int KV, int G, bool FUSED>
__global__ __launch_bounds__(PA_THREADS) void paged_attn_decode_kernel(const PAArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [PA_WAVES][G][PA_REC] | q scratch | kv scratch
    // workgroup = one (virtual kv head, row, KV split); the body is shared with the persistent backbone launch (pa_body.cuh)
    pa_decode_body<KV, G, FUSED, false>(a, lds, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x >> 6, threadIdx.x, true, nullptr, 0);
}

// ---- short-context variant (code predictor: <= 17 keys): ONE wave per (row, kv-head), bf16 KV, fused
// norm + RoPE + KV write.  The general kernel spends ~12 us of instructions (4 waves x q prologue, batch
// loop, 32-way combine) on a context that fits one 8-token pass or two; this one is ~4x lighter.
// DENSE (the code predictor's private cache): position = a.dense_pos for every row and row b owns block b, so no index
// is loaded at all and every load of the kernel -- both 8-token history groups included -- goes out in one round trip.
// SPLITQ: one wave per (row, Q head) instead of per (row, kv head) -- the wave is bound by its own ~1.5 k-instruction
// dependent chain, not by memory, so the q heads of a group run as separate waves (G = 1; the group's k-norm is
// recomputed by each, its K / V are stored by the first)
template <int G, bool DENSE, bool SPLITQ>
__global__ __launch_bounds__(256) void attn_small_fused_kernel(const PAArgs a, int npairs) {
    constexpr int KV = OMNI_KV_BF16;
    __shared__ float sm[4][G * 128 + 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= npairs) return;                       // no workgroup barrier below
    const int kv_heads = a.kv_heads, bs = a.bs;
    const int ratio = SPLITQ ? a.q_heads / kv_heads : 1;
    const int per_row = SPLITQ ? a.q_heads : kv_heads;
    const int row = pair / per_row;
    const int hsel = pair - row * per_row;             // SPLITQ: the q head; else the kv head
    const int kvh = SPLITQ ? hsel / ratio : hsel;
    const int qh0 = SPLITQ ? hsel : kvh * G;           // first (only) q head of this wave
    const bool kv_writer = !SPLITQ || hsel % ratio == 0;
    const int sub = lane & 7, tg = lane >> 3;
    const int32_t* bt = DENSE ? nullptr : a.block_table + (size_t)row * a.bt_stride;
    const int max_blk = a.bt_stride - 1;
    const int nslots = a.q_heads + 2 * kv_heads;
    // first 8-token group: issued before the sequence length is known
    const size_t r0 = DENSE ? ((size_t)row * bs + tg) * kv_heads + kvh
                            : ((size_t)bt[min(tg / bs, max_blk)] * bs + tg % bs) * kv_heads + kvh;
    KVRaw<KV> kr = load_row<KV>(a.k_cache, r0, sub), vr = load_row<KV>(a.v_cache, r0, sub);
    KVRaw<KV> kr1 = kr, vr1 = vr;
    if (DENSE) {                                      // tokens 8..15 (block_size >= 16: always a valid address)
        const size_t r1 = ((size_t)row * bs + 8 + tg) * kv_heads + kvh;
        kr1 = load_row<KV>(a.k_cache, r1, sub);
        vr1 = load_row<KV>(a.v_cache, r1, sub);
    }
    const int cur = DENSE ? a.dense_pos : a.seq_lens[row] - 1;
    const int pos = DENSE ? a.dense_pos : a.positions[row];
    const uint16_t* cs = a.cos_sin + (size_t)pa_rope_row(a, pos, row) * 128;
    float* wq = sm[wave];
    float* kvs = sm[wave] + G * 128;
    const float qs = a.sm_scale * LOG2E;
    float qf[G][16];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float y0, y1;
        head_norm_rope(a.qkv + ((size_t)row * nslots + qh0 + g) * 128, a.qnorm_w, cs, a.eps, lane, y0, y1);
        wq[g * 128 + lane] = y0;
        wq[g * 128 + 64 + lane] = y1;
    }
    {   // new token: K (norm + rope) and V -> cache and LDS
        const int64_t slot = DENSE ? (int64_t)row * bs + pos : (int64_t)bt[min(pos / bs, max_blk)] * bs + pos % bs;
        const bool live = DENSE || !a.num_live || row < *a.num_live;      // DENSE = the code predictor's private cache
        if (hsel == 0 && lane == 0 && a.slot_out && live) a.slot_out[row] = slot;
        const size_t crow = (size_t)slot * kv_heads + kvh;
        float kx0, kx1;
        head_norm_rope(a.qkv + ((size_t)row * nslots + a.q_heads + kvh) * 128, a.knorm_w, cs, a.eps, lane, kx0, kx1);
        const uint16_t* vsrc = a.qkv + ((size_t)row * nslots + a.q_heads + kv_heads + kvh) * 128;
        const uint16_t v0 = vsrc[lane], v1 = vsrc[lane + 64];
        uint16_t* kd = reinterpret_cast<uint16_t*>(a.k_cache) + crow * 128;
        uint16_t* vd = reinterpret_cast<uint16_t*>(a.v_cache) + crow * 128;
        if (kv_writer && live) {
            kd[lane] = f2bf(kx0); kd[lane + 64] = f2bf(kx1);
            vd[lane] = v0; vd[lane + 64] = v1;
        }
        kvs[lane] = kx0; kvs[64 + lane] = kx1;
        kvs[128 + lane] = bf2f(v0); kvs[192 + lane] = bf2f(v1);
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) qf[g][e] = wq[g * 128 + elem_of<KV>(sub, e)] * qs;
    float m[G], l[G], acc[G][16];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
    }
    for (int t0 = 0; t0 < cur; t0 += 8) {
        if (t0 > 0) {
            if (DENSE) {
                kr = kr1; vr = vr1;
            } else {
                const int t = t0 + tg;
                const size_t r_ = ((size_t)bt[min(t / bs, max_blk)] * bs + t % bs) * kv_heads + kvh;
                kr = load_row<KV>(a.k_cache, r_, sub);
                vr = load_row<KV>(a.v_cache, r_, sub);
            }
        }
        const bool ok = t0 + tg < cur;
        float kf[16], vf[16];
        to_f32<KV>(kr, kf);
        to_f32<KV>(vr, vf);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) d = fmaf(qf[g][e], kf[e], d);
            d = group8_sum(d);
            d = ok ? d : -INFINITY;
            const float mn = fmaxf(m[g], d);
            const float corr = (mn == -INFINITY) ? 1.0f : exp2f(m[g] - mn);
            const float p = ok ? exp2f(d - mn) : 0.f;
            m[g] = mn;
            l[g] = l[g] * corr + p;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[g][e] = fmaf(p, vf[e], acc[g][e] * corr);
        }
    }
    {   // fold the new token into token-group 0
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
            for (int e = 0; e < 16; ++e) d = fmaf(qf[g][e], kf[e], d);
            d = group8_sum(d);
            if (tg == 0) {
                const float mn = fmaxf(m[g], d);
                const float corr = (m[g] == -INFINITY) ? 0.f : exp2f(m[g] - mn);
                const float p = exp2f(d - mn);
                m[g] = mn;
                l[g] = l[g] * corr + p;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[g][e] = fmaf(p, vf[e], acc[g][e] * corr);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float mw = m[g];
        mw = fmaxf(mw, dpp_f<OMNI_DPP_ROR8>(mw));       // the 8 token groups = lane bits 3..5: one in-row step, two cross-row
        mw = xor32_max(xor16_max(mw));
        const float sc = (m[g] == -INFINITY) ? 0.f : exp2f(m[g] - mw);
        float lw = l[g] * sc;
        lw += dpp_f<OMNI_DPP_ROR8>(lw);
        lw = xor32_sum(xor16_sum(lw));
        const float inv = 1.0f / lw;                  // >= the new token's weight, never 0
        // 16 partial outputs per lane, to be summed over the 8 token groups: lane ^ 8 as a DPP step on all 16, then two
        // HALVING exchanges (each lane passes on the half its partner keeps): 8 + 4 ds_bpermute instead of 48, and every
        // lane ends with 4 finished, adjacent outputs
        float a16[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = acc[g][e] * sc;
            a16[e] = v + dpp_f<OMNI_DPP_ROR8>(v);
        }
        const bool b1 = (lane & 16) != 0, b2 = (lane & 32) != 0;
        float a8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float keep = b1 ? a16[8 + e] : a16[e];
            const float send = b1 ? a16[e] : a16[8 + e];
            a8[e] = keep + xchg16(send, b1);
        }
        float a4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float keep = b2 ? a8[4 + e] : a8[e];
            const float send = b2 ? a8[e] : a8[4 + e];
            a4[e] = keep + xchg32(send, b2);
        }
        // the lane holds elements e0 .. e0 + 3 of its sub slice, e0 = 8 b1 + 4 b2 (elem_of<bf16>: e >= 8 -> dims 64 + ...)
        const int kq = (qh0 + g) * 128 + (b1 ? 64 : 0) + sub * 8 + (b2 ? 4 : 0);
        if ((lane & 8) == 0) {          // lanes ^ 8 hold the same four values
            uint16_t* op = a.out_frag ? a.out + frag_off(row, kq, a.q_heads * 128) : a.out + (size_t)row * a.q_heads * 128 + kq;
            *reinterpret_cast<uint2*>(op) = make_uint2(pack_bf2(a4[0] * inv, a4[1] * inv), pack_bf2(a4[2] * inv, a4[3] * inv));
        }
    }
}

// ---- code predictor, position p <= 15 of every row (dense private cache: row b owns block b): ONE wave per (row, q head)
// with the work laid out so that almost nothing is serial -- the general small kernel above spends ~1.5 k dependent
// instructions per wave (16 dims per lane, 8 tokens per pass, 28-step output reduction), 5.2 us for 17 keys, 70 launches a
// step.  Here:
//   scores   lane = (token t = lane / 4, quarter = lane % 4): 32 dims of K_t per lane (4 x 16-B loads), q transposed through
//            LDS, 32 FMAs, 2 DPP steps over the quarter -> all <= 16 history scores at once;
//   softmax  max / sum over the token lanes: 4 exchange steps each (the new token's score is wave-uniform);
//   PV       lane = output dims (2 l, 2 l + 1): p_t arrives by v_readlane (wave-uniform operand), V_t as ONE 4-byte load per
//            token -- no cross-lane reduction of the output at all.
// q / k-norm + RoPE keep the (l, l + 64) pairing of head_norm_rope.  Every load of the kernel is issued before the first use.
__global__ __launch_bounds__(256) void attn_tiny_dense_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ qnorm_w,
                                                              const uint16_t* __restrict__ knorm_w, const uint16_t* __restrict__ cos_sin,
                                                              float eps, uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                                                              uint16_t* __restrict__ out, int npairs, int q_heads, int kv_heads, int bs,
                                                              int pos, float sm_scale, int out_frag) {
    __shared__ float sq[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= npairs) return;                        // whole waves leave: no workgroup barrier below
    const int row = pair / q_heads, h = pair - row * q_heads;
    const int ratio = q_heads / kv_heads, kvh = h / ratio;
    const int nslots = q_heads + 2 * kv_heads;
    const int t = lane >> 2, qd = lane & 3;            // score phase: token, quarter of the head dimension
    // ---- all loads up front
    const size_t hrow = ((size_t)row * bs + t) * kv_heads + kvh;             // history row t (t < 16 <= bs: valid address)
    u32x4 kq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kq[j] = *reinterpret_cast<const u32x4*>(k_cache + hrow * 128 + qd * 32 + j * 8);
    uint32_t vq[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
        vq[u] = *reinterpret_cast<const uint32_t*>(v_cache + (((size_t)row * bs + u) * kv_heads + kvh) * 128 + 2 * lane);
    const uint16_t* qsrc = qkv + ((size_t)row * nslots + h) * 128;
    const uint16_t* ksrc = qkv + ((size_t)row * nslots + q_heads + kvh) * 128;
    const uint16_t* vsrc = qkv + ((size_t)row * nslots + q_heads + kv_heads + kvh) * 128;
    const uint32_t vnew = *reinterpret_cast<const uint32_t*>(vsrc + 2 * lane);
    const uint16_t* cs = cos_sin + (size_t)pos * 128;
    float q0, q1, k0, k1;
    head_norm_rope(qsrc, qnorm_w, cs, eps, lane, q0, q1);
    head_norm_rope(ksrc, knorm_w, cs, eps, lane, k0, k1);
    // ---- the new token's K / V into the private cache (one writer per kv head)
    if (h % ratio == 0) {
        const size_t crow = (((size_t)row * bs + pos) * kv_heads + kvh) * 128;
        k_cache[crow + lane] = f2bf(k0);
        k_cache[crow + 64 + lane] = f2bf(k1);
        *reinterpret_cast<uint32_t*>(v_cache + crow + 2 * lane) = vnew;
    }
    const float qs = sm_scale * LOG2E;
    const float s_new = wave_sum(fmaf(q0 * qs, k0, (q1 * qs) * k1));        // wave-uniform
    // ---- history scores: q through LDS into the (token, quarter) layout
    sq[wave][lane] = q0 * qs;
    sq[wave][lane + 64] = q1 * qs;
    __builtin_amdgcn_wave_barrier();                   // other lanes' writes are read below (same wave: LDS executes in order)
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 qa = *reinterpret_cast<const f32x4*>(&sq[wave][qd * 32 + j * 8]);
        const f32x4 qb = *reinterpret_cast<const f32x4*>(&sq[wave][qd * 32 + j * 8 + 4]);
        d = fmaf(qa[0], bf_lo(kq[j][0]), d); d = fmaf(qa[1], bf_hi(kq[j][0]), d);
        d = fmaf(qa[2], bf_lo(kq[j][1]), d); d = fmaf(qa[3], bf_hi(kq[j][1]), d);
        d = fmaf(qb[0], bf_lo(kq[j][2]), d); d = fmaf(qb[1], bf_hi(kq[j][2]), d);
        d = fmaf(qb[2], bf_lo(kq[j][3]), d); d = fmaf(qb[3], bf_hi(kq[j][3]), d);
    }
    d += dpp_f<OMNI_DPP_XOR1>(d);
    d += dpp_f<OMNI_DPP_XOR2>(d);                      // the 4 quarter lanes of token t now hold s_t
    const float s = t < pos ? d : -INFINITY;
    // ---- softmax over the history tokens (lane bits 2..5) and the new token
    float m = fmaxf(s, dpp_f<OMNI_DPP_HALF_MIRROR>(s));    // lanes i <-> 7 - i: the other token of the 8-lane group
    m = fmaxf(m, dpp_f<OMNI_DPP_MIRROR>(m));               // the other half of the row: 4 tokens
    m = xor32_max(xor16_max(m));
    m = fmaxf(m, s_new);
    const float p = exp2f(s - m);                          // 0 for masked tokens
    const float p_new = exp2f(s_new - m);
    float l = p + dpp_f<OMNI_DPP_HALF_MIRROR>(p);
    l += dpp_f<OMNI_DPP_MIRROR>(l);
    l = xor32_sum(xor16_sum(l));                           // each of the 16 tokens exactly once (the mirrors pair DIFFERENT tokens)
    const float inv = 1.0f / (l + p_new);
    // ---- PV in the (2 l, 2 l + 1) layout: p_t is wave-uniform after a readlane
    float o0 = p_new * bf_lo(vnew), o1 = p_new * bf_hi(vnew);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const float pu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 4 * u));
        o0 = fmaf(pu, bf_lo(vq[u]), o0);
        o1 = fmaf(pu, bf_hi(vq[u]), o1);
    }
    const int col = h * 128 + 2 * lane;
    uint16_t* op = out_frag ? out + frag_off(row, col, q_heads * 128) : out + (size_t)row * q_heads * 128 + col;
    *reinterpret_cast<uint32_t*>(op) = pack_bf2(o0 * inv, o1 * inv);
}

// ---- code predictor, positions 0 and 1 of every row in ONE launch (both inputs are known when the pass starts: the
// talker's last hidden state and the layer-0 code embedding).  One wave per (row, q head): position 0 attends to itself
// only (output = its V row, exactly), position 1 to both; the position-0 K is rebuilt from the qkv rows by every wave
// that needs it instead of being read back from the cache.  Dense private cache: row b owns block b.
__global__ __launch_bounds__(256) void attn_pair01_kernel(const uint16_t* __restrict__ qkv, int row1_off,
                                                          const uint16_t* __restrict__ qnorm_w, const uint16_t* __restrict__ knorm_w,
                                                          const uint16_t* __restrict__ cos_sin, float eps,
                                                          uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                                                          uint16_t* __restrict__ out, int npairs, int q_heads, int kv_heads, int bs,
                                                          float sm_scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= npairs) return;                       // whole waves leave: the reductions below see full waves
    const int row = pair / q_heads, h = pair - row * q_heads;
    const int ratio = q_heads / kv_heads, kvh = h / ratio;
    const int nslots = q_heads + 2 * kv_heads;
    const uint16_t* r0 = qkv + (size_t)row * nslots * 128;
    const uint16_t* r1 = qkv + (size_t)(row1_off + row) * nslots * 128;
    const uint16_t* v0p = r0 + (size_t)(q_heads + kv_heads + kvh) * 128;
    const uint16_t* v1p = r1 + (size_t)(q_heads + kv_heads + kvh) * 128;
    const uint16_t v0a = v0p[lane], v0b = v0p[lane + 64], v1a = v1p[lane], v1b = v1p[lane + 64];
    float k00, k01, k10, k11, q0, q1;
    head_norm_rope(r0 + (size_t)(q_heads + kvh) * 128, knorm_w, cos_sin, eps, lane, k00, k01);
    head_norm_rope(r1 + (size_t)(q_heads + kvh) * 128, knorm_w, cos_sin + 128, eps, lane, k10, k11);
    head_norm_rope(r1 + (size_t)h * 128, qnorm_w, cos_sin + 128, eps, lane, q0, q1);
    if (h % ratio == 0) {                             // one writer per kv head
        const size_t c0 = ((size_t)row * bs * kv_heads + kvh) * 128, c1 = (((size_t)row * bs + 1) * kv_heads + kvh) * 128;
        k_cache[c0 + lane] = f2bf(k00); k_cache[c0 + 64 + lane] = f2bf(k01);
        v_cache[c0 + lane] = v0a;       v_cache[c0 + 64 + lane] = v0b;
        k_cache[c1 + lane] = f2bf(k10); k_cache[c1 + 64 + lane] = f2bf(k11);
        v_cache[c1 + lane] = v1a;       v_cache[c1 + 64 + lane] = v1b;
    }
    const float qs = sm_scale * LOG2E;
    const float s0 = wave_sum(fmaf(q0 * qs, k00, (q1 * qs) * k01));
    const float s1 = wave_sum(fmaf(q0 * qs, k10, (q1 * qs) * k11));
    const float m = fmaxf(s0, s1);
    const float p0 = exp2f(s0 - m), p1 = exp2f(s1 - m);
    const float inv = 1.0f / (p0 + p1);
    const int width = q_heads * 128, col = h * 128 + lane;
    out[frag_off(row, col, width)] = v0a;
    out[frag_off(row, col + 64, width)] = v0b;
    out[frag_off(row1_off + row, col, width)] = f2bf(fmaf(p1, bf2f(v1a), p0 * bf2f(v0a)) * inv);
    out[frag_off(row1_off + row, col + 64, width)] = f2bf(fmaf(p1, bf2f(v1b), p0 * bf2f(v0b)) * inv);
}

int k_attn_pair01(const void* qkv, int row1_off, const void* qnorm_w, const void* knorm_w, const void* cos_sin, float eps,
                  void* k_cache, void* v_cache, void* out, int B, int q_heads, int kv_heads, int block_size, float sm_scale,
                  void* stream) {
    OMNI_CHECK_ARG(qkv && qnorm_w && knorm_w && cos_sin && k_cache && v_cache && out, "attn_pair01: null pointer");
    OMNI_CHECK_ARG(B >= 1 && kv_heads >= 1 && q_heads % kv_heads == 0 && block_size >= 2 && row1_off % 16 == 0 && row1_off >= B,
                   "attn_pair01: B=%d heads %d/%d block %d row1_off %d", B, q_heads, kv_heads, block_size, row1_off);
    const int npairs = B * q_heads;
    hipLaunchKernelGGL(attn_pair01_kernel, dim3((npairs + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, row1_off,
                       (const uint16_t*)qnorm_w, (const uint16_t*)knorm_w, (const uint16_t*)cos_sin, eps, (uint16_t*)k_cache,
                       (uint16_t*)v_cache, (uint16_t*)out, npairs, q_heads, kv_heads, block_size, sm_scale);
    OMNI_CHECK_LAUNCH("attn_pair01");
    return OMNI_OK;
}

// merge KV splits: one 128-thread block per (row, q-head)
__global__ __launch_bounds__(128) void paged_attn_merge_kernel(const float* __restrict__ partial,
                                                               uint16_t* __restrict__ out, int nsplit, float v_mul,
                                                               int q_heads, int out_frag, const float* __restrict__ v_mul_dev) {
    if (v_mul_dev) v_mul = v_mul_dev[1];           // fp8 KV with device-resident scales: {k_scale, v_scale}
    const size_t rh = blockIdx.x;
    const int d = threadIdx.x;
    const float* base = partial + rh * nsplit * PA_REC;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, base[s * PA_REC]);
    float L = 0.f, A = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* rec = base + s * PA_REC;
        const float w = (rec[0] == -INFINITY) ? 0.f : exp2f(rec[0] - M);
        L = fmaf(w, rec[1], L);           // (the shared factor FIRST: see the note on packed FMAs in pa_body.cuh)
        A = fmaf(w, rec[2 + d], A);
    }
    const int row = (int)(rh / q_heads), qh = (int)(rh % q_heads);
    out[out_frag ? frag_off(row, qh * 128 + d, q_heads * 128) : rh * 128 + d] = f2bf(L > 0.f ? (A / L) * v_mul : 0.f);
}

static int pick_nsplit(int rows, int kv_heads, int max_seq_len) {
    const int wgs = rows * kv_heads;
    if (wgs >= 384) return 1;                         // 1.5+ workgroups per CU already (and: the persistent backbone launch of
                                                      // bb_all.hip never splits -- 48+ rows x 8 heads take the same arithmetic on both paths)
    // as many KV splits as keep the grid within ONE round of the chip (512 workgroups of 4 waves = two per CU): rounding UP put 320
    // workgroups (40 rows x 8 heads) at 640 -- a second, quarter-filled round plus the merge launch: 0.45 ms per step for the 28 launches
    // against 0.33 unsplit (round 5, profiles/r05_other_configs.txt)
    int ns = 512 / wgs;
    const int cap = (max_seq_len + 255) / 256;        // >= 256 tokens per split
    if (ns > cap) ns = cap;
    if (ns > 16) ns = 16;
    return ns < 1 ? 1 : ns;
}

// workspace = PA_MERGE_CNT arrival counters (zero when handed over; every launch leaves them zero) + the splits' partial records
#define PA_MERGE_CNT 512
extern "C" int64_t omni_paged_attn_workspace_bytes(int B, int q_heads, int head_dim, int max_seq_len) {
    (void)head_dim;
    (void)max_seq_len;
    return (int64_t)PA_MERGE_CNT * 4 + (int64_t)B * q_heads * 16 * PA_REC * sizeof(float);
}

// g_pa_merge_inkernel (round 6): the KV splits merged by their last arriver inside the attention launch -- built, bit-identical
// (tests/test_gpu_ops.py), and measured against the merge launch in one process (profiles/r06_ab_kv_merge.txt): 1 row (16 splits) +0.087 ms per
// step, 16 / 32 rows a tie -- the merge launch spreads (row, head) pairs over 128-thread blocks of its own and costs ~3 us behind an 11 us
// attention; the last arriver walks its records alone.  Off; the debug library keeps the arm.
OMNI_KNOB g_pa_int8_max_g = 2, g_pa_tail_chunks = 1, g_pa_merge_inkernel = 0;
#ifdef OMNI_DEBUG_HOOKS
extern "C" void omni_debug_pa_merge(int inkernel) { g_pa_merge_inkernel = inkernel; }      // 1: KV splits merged by their last arriver (A/B arm)
extern "C" void omni_debug_int8_max_g(int g) { g_pa_int8_max_g = g; }
extern "C" void omni_debug_pa_tail(int on) { g_pa_tail_chunks = on; }      // A/B: the tail round as contiguous chunks (PAArgs::tail_chunks)
#endif
template <int KV, bool FUSED>
static int launch_pa(const PAArgs& a_in, int rows, hipStream_t st) {
    PAArgs a = a_in;
    int G = a.q_heads / a.kv_heads;
    a.kv_rep = 1;
    a.tail_chunks = g_pa_tail_chunks;
    // e.g. 16 q / 2 kv heads: 4 groups of 4 per kv head (int8: see g_pa_int8_max_g)
    const int max_g = (KV == OMNI_KV_INT8) ? g_pa_int8_max_g : 4;
    while (G > max_g && G % 2 == 0) { G /= 2; a.kv_rep *= 2; }
    if (rows * a.kv_heads * a.kv_rep >= 512) a.nsplit = 1;      // the virtual heads already fill the chip: no KV split
    // the workspace opens with the arrival counters of the in-kernel merge (a split launch has fewer than 512 (row, virtual head) pairs)
    float* ws = a.partial;
    a.merge_cnt = nullptr;
    if (ws != nullptr) {
        a.partial = ws + PA_MERGE_CNT;
        if (a.nsplit > 1 && g_pa_merge_inkernel && rows * a.kv_heads * a.kv_rep <= PA_MERGE_CNT) a.merge_cnt = reinterpret_cast<unsigned*>(ws);
    }
    dim3 grid(a.kv_heads * a.kv_rep, rows, a.nsplit), block(PA_THREADS);
    const size_t lds = (size_t)PA_LDS_FLOATS(G) * sizeof(float);
#define LAUNCH(GG) hipLaunchKernelGGL((paged_attn_decode_kernel<KV, GG, FUSED>), grid, block, lds, st, a)
    switch (G) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 4: LAUNCH(4); break;
        default: omni_set_error("omni_paged_attn: Hq/Hkv=%d unsupported (1, 2, 4 and their multiples by powers of two)", a.q_heads / a.kv_heads); return OMNI_EINVAL;
    }
#undef LAUNCH
    OMNI_CHECK_LAUNCH("omni_paged_attn_decode");
    if (a.nsplit > 1 && a.merge_cnt == nullptr) {
        hipLaunchKernelGGL(paged_attn_merge_kernel, dim3(rows * a.q_heads), dim3(128), 0, st, (const float*)a.partial,
                           a.out, a.nsplit, KV == OMNI_KV_FP8 ? a.v_scale : 1.0f, a.q_heads, a.out_frag,
                           KV == OMNI_KV_FP8 ? a.scale_dev : nullptr);
        OMNI_CHECK_LAUNCH("omni_paged_attn_merge");
    }
    return OMNI_OK;
}

static int pa_dispatch(PAArgs& a, int rows, int head_dim, int kv_dtype, bool fused, void* stream) {
    OMNI_CHECK_ARG(a.k_cache && a.v_cache && a.block_table && a.seq_lens && a.out, "omni_paged_attn: null pointer");
    OMNI_CHECK_ARG(fused ? (a.qkv && a.qnorm_w && a.knorm_w && a.positions && a.cos_sin) : (a.q != nullptr),
                   "omni_paged_attn: null q / qkv inputs");
    OMNI_CHECK_ARG(head_dim == 128, "omni_paged_attn: head_dim=%d (only 128)", head_dim);
    OMNI_CHECK_ARG(a.kv_heads > 0 && a.q_heads % a.kv_heads == 0, "omni_paged_attn: q_heads=%d kv_heads=%d", a.q_heads, a.kv_heads);
    OMNI_CHECK_ARG(a.bs > 0 && (a.bs & (a.bs - 1)) == 0 && a.bt_stride > 0, "omni_paged_attn: block_size=%d (a power of two) bt_stride=%d", a.bs, a.bt_stride);
    a.bs_shift = __builtin_ctz((unsigned)a.bs);
    OMNI_CHECK_ARG(kv_dtype != OMNI_KV_INT8 || (a.k_scales && a.v_scales), "omni_paged_attn: int8 needs scale arrays");
    OMNI_CHECK_ARG(a.nsplit == 1 || a.partial, "omni_paged_attn: workspace required for KV splits");
    OMNI_CHECK_ARG(a.k_scale > 0.f && a.v_scale > 0.f, "omni_paged_attn: scales must be > 0");
    if (rows <= 0) return OMNI_OK;
    hipStream_t st = (hipStream_t)stream;
#define GO(KVT) (fused ? launch_pa<KVT, true>(a, rows, st) : launch_pa<KVT, false>(a, rows, st))
    switch (kv_dtype) {
        case OMNI_KV_BF16: return GO(OMNI_KV_BF16);
        case OMNI_KV_FP8: return GO(OMNI_KV_FP8);
        case OMNI_KV_INT8: return GO(OMNI_KV_INT8);
        case OMNI_KV_FP16: return GO(OMNI_KV_FP16);
        default: omni_set_error("omni_paged_attn: kv_dtype=%d", kv_dtype); return OMNI_EINVAL;
    }
#undef GO
}

extern "C" int omni_paged_attn_decode(const void* q, const void* k_cache, const void* v_cache, const float* k_scales,
                                      const float* v_scales, const int32_t* block_table, int bt_stride,
                                      const int32_t* seq_lens, void* out, void* workspace, int B, int q_heads,
                                      int kv_heads, int head_dim, int block_size, int kv_dtype, float k_scale,
                                      float v_scale, float sm_scale, int max_seq_len, void* stream) {
    PAArgs a{};
    a.q = (const uint16_t*)q; a.k_cache = const_cast<void*>(k_cache); a.v_cache = const_cast<void*>(v_cache);
    a.k_scales = const_cast<float*>(k_scales); a.v_scales = const_cast<float*>(v_scales);
    a.block_table = block_table; a.bt_stride = bt_stride; a.seq_lens = seq_lens; a.out = (uint16_t*)out;
    a.partial = (float*)workspace; a.q_heads = q_heads; a.kv_heads = kv_heads; a.bs = block_size;
    a.k_scale = k_scale; a.v_scale = v_scale; a.sm_scale = sm_scale;
    a.nsplit = (workspace && kv_heads > 0 && B > 0) ? pick_nsplit(B, kv_heads, max_seq_len) : 1;
    return pa_dispatch(a, B, head_dim, kv_dtype, false, stream);
}

OMNI_KNOB g_small_splitq = 1, g_small_tiny = 1;
#ifdef OMNI_DEBUG_HOOKS
extern "C" void omni_debug_small_splitq(int on) { g_small_splitq = on; }
extern "C" void omni_debug_small_tiny(int on) { g_small_tiny = on; }
#endif

int k_attn_decode_fused(const void* qkv, const void* qnorm_w, const void* knorm_w, const int32_t* positions,
                        const void* cos_sin, float eps, void* k_cache, void* v_cache, float* k_scales, float* v_scales,
                        const int32_t* block_table, int bt_stride, const int32_t* seq_lens, int64_t* slot_out, void* out,
                        void* workspace, int B, int q_heads, int kv_heads, int head_dim, int block_size, int kv_dtype,
                        float k_scale, float v_scale, float sm_scale, int max_seq_len, int out_frag, int dense_pos,
                        void* stream, const int32_t* num_live, const int32_t* rope_delta, int rope_rows, const float* scale_dev) {
    PAArgs a{};
    a.rope_delta = rope_delta;
    a.rope_rows = rope_delta ? rope_rows : 0;
    a.scale_dev = kv_dtype == OMNI_KV_FP8 ? scale_dev : nullptr;
    a.out_frag = out_frag;
    a.dense_pos = dense_pos;
    a.num_live = num_live;
    a.qkv = (const uint16_t*)qkv; a.qnorm_w = (const uint16_t*)qnorm_w; a.knorm_w = (const uint16_t*)knorm_w;
    a.positions = positions; a.cos_sin = (const uint16_t*)cos_sin; a.eps = eps; a.slot_out = slot_out;
    a.k_cache = k_cache; a.v_cache = v_cache; a.k_scales = k_scales; a.v_scales = v_scales;
    a.block_table = block_table; a.bt_stride = bt_stride; a.seq_lens = seq_lens; a.out = (uint16_t*)out;
    a.partial = (float*)workspace; a.q_heads = q_heads; a.kv_heads = kv_heads; a.bs = block_size;
    a.k_scale = k_scale; a.v_scale = v_scale; a.sm_scale = sm_scale;
    a.nsplit = (workspace && kv_heads > 0 && B > 0) ? pick_nsplit(B, kv_heads, max_seq_len) : 1;
    const bool dense = dense_pos >= 0;
    OMNI_CHECK_ARG(!dense || (kv_dtype == OMNI_KV_BF16 && block_size >= 16 && dense_pos < 16 && dense_pos < block_size &&
                              head_dim == 128), "attn_decode_fused: dense mode needs bf16 KV, block_size >= 16, position < 16");
    if (kv_dtype == OMNI_KV_BF16 && (dense || max_seq_len <= 64) && head_dim == 128 && kv_heads > 0 && q_heads % kv_heads == 0 &&
        B > 0 && qkv && qnorm_w && knorm_w && cos_sin && k_cache && v_cache && out && block_size > 0 &&
        (dense || (positions && block_table && seq_lens && bt_stride > 0))) {
        const int G = q_heads / kv_heads;
        hipStream_t st = (hipStream_t)stream;
        a.nsplit = 1;
        if (dense && g_small_tiny) {
            const int npairs = B * q_heads;
            hipLaunchKernelGGL(attn_tiny_dense_kernel, dim3((npairs + 3) / 4), dim3(256), 0, st, a.qkv, a.qnorm_w, a.knorm_w, a.cos_sin, eps,
                               (uint16_t*)k_cache, (uint16_t*)v_cache, a.out, npairs, q_heads, kv_heads, block_size, dense_pos, sm_scale,
                               out_frag);
            OMNI_CHECK_LAUNCH("omni_attn_decode_fused(tiny dense)");
            return OMNI_OK;
        }
        if (g_small_splitq && G > 1) {
            const int npairs = B * q_heads;
            dim3 grid((npairs + 3) / 4), block(256);
            if (dense) hipLaunchKernelGGL((attn_small_fused_kernel<1, true, true>), grid, block, 0, st, a, npairs);
            else hipLaunchKernelGGL((attn_small_fused_kernel<1, false, true>), grid, block, 0, st, a, npairs);
            OMNI_CHECK_LAUNCH("omni_attn_decode_fused(small, per q head)");
            return OMNI_OK;
        }
        const int npairs = B * kv_heads;
        dim3 grid((npairs + 3) / 4), block(256);
#define SMALL(G_)                                                                                        \
        if (G == G_) {                                                                                   \
            if (dense) hipLaunchKernelGGL((attn_small_fused_kernel<G_, true, false>), grid, block, 0, st, a, npairs);  \
            else hipLaunchKernelGGL((attn_small_fused_kernel<G_, false, false>), grid, block, 0, st, a, npairs);       \
        }
        SMALL(1) SMALL(2) SMALL(4)
#undef SMALL
        if (G == 1 || G == 2 || G == 4) {
            OMNI_CHECK_LAUNCH("omni_attn_decode_fused(small)");
            return OMNI_OK;
        }
    }
    OMNI_CHECK_ARG(!dense, "attn_decode_fused: dense mode unsupported for q_heads / kv_heads = %d", kv_heads ? q_heads / kv_heads : 0);
    return pa_dispatch(a, B, head_dim, kv_dtype, true, stream);
}

extern "C" int omni_attn_decode_fused(const void* qkv, const void* qnorm_w, const void* knorm_w, const int32_t* positions,
                                      const void* cos_sin, float eps, void* k_cache, void* v_cache, float* k_scales,
                                      float* v_scales, const int32_t* block_table, int bt_stride, const int32_t* seq_lens,
                                      int64_t* slot_out, void* out, void* workspace, int B, int q_heads, int kv_heads,
                                      int head_dim, int block_size, int kv_dtype, float k_scale, float v_scale,
                                      float sm_scale, int max_seq_len, void* stream) {
    return k_attn_decode_fused(qkv, qnorm_w, knorm_w, positions, cos_sin, eps, k_cache, v_cache, k_scales, v_scales,
                               block_table, bt_stride, seq_lens, slot_out, out, workspace, B, q_heads, kv_heads, head_dim,
                               block_size, kv_dtype, k_scale, v_scale, sm_scale, max_seq_len, 0, -1, stream);
}

int k_paged_attn_prefill(const void* q, const void* k_cache, const void* v_cache, const float* k_scales,
                         const float* v_scales, const int32_t* block_table, int bt_stride, const int32_t* req_of_tok,
                         const int32_t* positions, void* out, int T, int q_heads, int kv_heads, int head_dim, int block_size,
                         int kv_dtype, float k_scale, float v_scale, float sm_scale, int out_frag, void* stream) {
    OMNI_CHECK_ARG(req_of_tok && positions, "omni_paged_attn_prefill: null pointer");
    if (k_prefill_mfma_supported(q_heads, kv_heads, head_dim))
        return k_prefill_mfma(q, k_cache, v_cache, k_scales, v_scales, block_table, bt_stride, req_of_tok, positions, out, T,
                              q_heads, kv_heads, block_size, kv_dtype, k_scale, v_scale, sm_scale, out_frag, stream);
    // other head ratios: every token is a decode row whose context is positions[t] + 1 keys of request req_of_tok[t]
    PAArgs a{};
    a.out_frag = out_frag;
    a.q = (const uint16_t*)q; a.k_cache = const_cast<void*>(k_cache); a.v_cache = const_cast<void*>(v_cache);
    a.k_scales = const_cast<float*>(k_scales); a.v_scales = const_cast<float*>(v_scales);
    a.block_table = block_table; a.bt_stride = bt_stride; a.seq_lens = positions; a.req_of_row = req_of_tok;
    a.seq_from_pos = 1; a.out = (uint16_t*)out; a.q_heads = q_heads; a.kv_heads = kv_heads; a.bs = block_size;
    a.k_scale = k_scale; a.v_scale = v_scale; a.sm_scale = sm_scale; a.nsplit = 1;
    return pa_dispatch(a, T, head_dim, kv_dtype, false, stream);
}

extern "C" int omni_paged_attn_prefill(const void* q, const void* k_cache, const void* v_cache, const float* k_scales,
                                       const float* v_scales, const int32_t* block_table, int bt_stride,
                                       const int32_t* req_of_tok, const int32_t* positions, void* out, int T,
                                       int q_heads, int kv_heads, int head_dim, int block_size, int kv_dtype,
                                       float k_scale, float v_scale, float sm_scale, void* stream) {
    return k_paged_attn_prefill(q, k_cache, v_cache, k_scales, v_scales, block_table, bt_stride, req_of_tok, positions, out, T,
                                q_heads, kv_heads, head_dim, block_size, kv_dtype, k_scale, v_scale, sm_scale, 0, stream);
}
