// The code predictor's layer stack as ONE persistent launch (reference: the 5-layer decoder that
// qwen3_tts_code_predictor_vllm.py:480-561 re-runs for each of the 15 residual code groups of a talker step).
//
// Launch-per-op, a predictor pass is 25 dependent launches (5 layers x { qkv, attention, o_proj, gate_up, down_proj }) of
// ~5.5 us each although a workgroup's share of a GEMM is 16-48 KB of weights and a few MFMAs: a dependent launch pays the
// kernel boundary (1.65 us), a cold weight fetch, a cold activation fetch and a store drain, one after the other.
// Here the 25 stages run inside one grid of 256 co-resident workgroups (one per CU):
//   * a stage's WEIGHT slice does not depend on its predecessor: every wave loads its whole slice (4-12 k-steps, 32-64 VGPRs)
//     before it looks at the stage flags, so the weight latency overlaps the predecessor's tail;
//   * stages hand activations over "coherent by access" (coherent.cuh): sc1 write-through stores, one flag word per
//     workgroup, sc1 loads behind the flag poll -- no cache write-back / invalidate, no kernel boundary;
//   * what a stage may read BEFORE the flags (besides weights): bytes that were final two or more stages earlier -- the
//     K / V history of the private cache, the old residual values of a residual epilogue.
// Tiles, k-step ownership of the 8 waves, accumulation order, LDS combine order and every rounding point are those of
// gemm_skinny_kernel / attn_tiny_dense_kernel: the chain is bit-identical to the launch-per-op path (tests/test_gpu_chain.py).
// Bounded spins: a grid that is not co-resident times out into an error word (never a hang).
#include "attn_common.cuh"
#include "coherent.cuh"
#include "common.cuh"
#include "gemm_frag.cuh"
#include "chain_gemm.cuh"
#include "kernels.h"
#include "sampler_body.cuh"

#define CH_MAX_LAYERS 8
#define CH_LDS_FLOATS ((CH_WAVES * 6 * 4 * 64) + CH_WAVES * 64)      // combine slots of the widest stage (NT * MT = 6) + rstd area

struct ChainLayer {
    const uint16_t *ln1, *wqkv, *qnorm, *knorm, *wo, *ln2, *wgu, *wdown;
    uint16_t *kc, *vc;                    // this layer's private K / V cache [B][bs][kv_heads][128] bf16
};
struct ChainArgs {
    ChainLayer layer[CH_MAX_LAYERS];
    int layers, B, bs, np_in;
    int g0, g1;                           // passes = buffer positions g0 .. g1 - 1 (1 <= g0, g1 <= 16); with_head: each pass ends in
    int with_head;                        // the group's head GEMM + sampler (+ gather of the next pass's input row)
    int q_heads, kv_heads;
    float eps, sm_scale;
    uint16_t* resid;                      // fragment-major residual stream [64][Hc]
    float* part;                          // sum(r^2) slabs [np][64]
    uint16_t *qkv, *attn, *act;           // stage outputs: row-major [B][4096], fragment-major [64][2048], fragment-major [64][3072]
    const uint16_t* cos_sin;
    // head + sampler of pass g (group g): logits row b at logits + b * logits_ld + (g - 1) * logits_pass
    const uint16_t *cp_norm, *lm_head;    // lm_head: [Q - 1][codebook][Hc] fragment-major per group
    float* logits; int logits_ld, logits_pass;
    int greedy, top_k, Q, codebook; float temperature, top_p; uint32_t seed;
    const int32_t* steps; const uint32_t* row_seed;
    int32_t* codes;                       // [B][Q]
    const uint16_t* ptab;                 // folded projection tables [Q - 1][codebook][Hc]: pass g < Q - 1 gathers row `code` of table g - 1
    uint32_t* flags;
    int32_t* err;
    // tail of the all-pass launch (round 6): input assembly of the backbone by the last sampler stage's row owners, then layer 0's qkv
    struct Tail {
        const int32_t* input_ids; const uint16_t* embed; int vocab; const uint16_t* cp_embed; const uint16_t* text_step;
        uint16_t* x_out; uint16_t* resid; float* part; int64_t* audio_codes; int H;
        const uint16_t* wqkv; const uint16_t* ln1; uint16_t* qkv; int NQ;
        int enabled;
    } tail;
    int Bp;                               // pair kernel: rows of position 1 start at Bp (B rounded up to 16) in the 128-row stream
    int skip;                             // debug library: ingest experiment (coherent.cuh ChainGate::skip)
    int dom, gu_narrow, nap;              // policy (run-time knobs in the debug library): flag domain (coherent.cuh), gate_up on the launch path's 32 x 24 tile, poll pause
    unsigned long long* stamps;           // debug library only: [stage][CH_NSTAMP][256] s_memrealtime ticks (100 MHz) of wave 0, or NULL
};

// ---- attention stage at buffer position pos <= 15 (dense private cache: row b owns block b): one wave per (row, q head),
// the work layout of attn_tiny_dense_kernel (paged_attn.hip).  Waves 0-3 of workgroup w take pairs 4 w .. 4 w + 3.
__device__ __forceinline__ void chain_attn(const ChainArgs& a, const ChainLayer& L, int pos, float* lds, ChainGate& g, int code,
                                           unsigned long long* stamps) {
    const int sidx = ((code & 255) >> 4) * 5 + (code & 15) - 1;
    CH_STAMP(stamps, sidx, 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q_heads = a.q_heads, kv_heads = a.kv_heads, bs = a.bs;
    if ((int)blockIdx.x * 4 >= a.B * q_heads) {                 // none of this workgroup's four pairs is a live row
        chain_gate_skip(g);
        return;
    }
    const int pair = blockIdx.x * 4 + (wave & 3);               // waves 4-7 do the work: wave 0 polls the flags, and a wave's
    const bool active = wave >= 4 && pair < a.B * q_heads;      // loads return in order -- history loads ahead of a poll delay it
    const int row = pair / q_heads, h = pair - row * q_heads;
    const int ratio = q_heads / kv_heads, kvh = h / ratio;
    const int nslots = q_heads + 2 * kv_heads;
    const int t = lane >> 2, qd = lane & 3;
    const coh_rsrc_t krs = coh_rsrc(L.kc), vrs = coh_rsrc(L.vc), qrs = coh_rsrc(a.qkv), ars = coh_rsrc(a.attn);
    // ---- before the flags: the history rows (written by earlier passes; rows >= pos hold stale bytes and are masked)
    u32x4 kq[4];
    uint32_t vq[16];
    if (active) {
        const uint32_t hrow = ((uint32_t)(row * bs + t) * kv_heads + kvh) * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) kq[j] = coh_ld16(krs, (hrow + qd * 32 + j * 8) * 2);
#pragma unroll
        for (int u = 0; u < 16; ++u) vq[u] = coh_ld4(vrs, (((uint32_t)(row * bs + u) * kv_heads + kvh) * 128 + 2 * lane) * 2);
    }
    CH_STAMP(stamps, sidx, 1);
    chain_gate_wait(g, code);
    CH_STAMP(stamps, sidx, 2);
    if (active) {
        // ---- behind the flags: this row's q / k / v heads out of the qkv GEMM's output, one dword per lane each; the
        // (l, l + 64) pairing of the norm + RoPE is restored through LDS
        const uint32_t qoff = ((uint32_t)row * nslots + h) * 128, koff = ((uint32_t)row * nslots + q_heads + kvh) * 128,
                       voff = ((uint32_t)row * nslots + q_heads + kv_heads + kvh) * 128;
        const uint32_t qw = coh_ld4(qrs, (qoff + 2 * lane) * 2), kw = coh_ld4(qrs, (koff + 2 * lane) * 2);
        const uint32_t vnew = coh_ld4(qrs, (voff + 2 * lane) * 2);
        float* sq = lds + (wave & 3) * 256;                   // [0,128): q in the score layout; [128,256): raw q | k dwords
        uint32_t* raw = reinterpret_cast<uint32_t*>(sq + 128);
        raw[lane] = qw;
        raw[64 + lane] = kw;
        __builtin_amdgcn_wave_barrier();
        const uint16_t* rq = reinterpret_cast<const uint16_t*>(raw);
        const uint16_t* rk = rq + 128;
        const uint16_t* cs = a.cos_sin + (size_t)pos * 128;
        float q0, q1, k0, k1;
        head_norm_rope_vals(bf2f(rq[lane]), bf2f(rq[lane + 64]), L.qnorm, cs, a.eps, lane, q0, q1);
        head_norm_rope_vals(bf2f(rk[lane]), bf2f(rk[lane + 64]), L.knorm, cs, a.eps, lane, k0, k1);
        if (h % ratio == 0) {
            // the new token's K / V into the private cache: one dword per lane (even lanes elements l, l + 1 of the first
            // half, odd lanes l + 63, l + 64 of the second), written through for the later passes
            const uint32_t crow = ((uint32_t)(row * bs + pos) * kv_heads + kvh) * 128;
            const uint32_t kb0 = f2bf(k0), kb1 = f2bf(k1);
            const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kb0, OMNI_DPP_XOR1, 0xF, 0xF, true);
            const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kb1, OMNI_DPP_XOR1, 0xF, 0xF, true);
            const bool odd = (lane & 1) != 0;
            const uint32_t word = odd ? (n1 | (kb1 << 16)) : (kb0 | (n0 << 16));
            const uint32_t elem = odd ? 64 + lane - 1 : lane;
            coh_st4(krs, (crow + elem) * 2, word);
            coh_st4(vrs, (crow + 2 * lane) * 2, vnew);
        }
        const float qs = a.sm_scale * LOG2E;
        const float s_new = wave_sum(fmaf(q0 * qs, k0, (q1 * qs) * k1));
        sq[lane] = q0 * qs;
        sq[lane + 64] = q1 * qs;
        __builtin_amdgcn_wave_barrier();
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 qa = *reinterpret_cast<const f32x4*>(&sq[qd * 32 + j * 8]);
            const f32x4 qb = *reinterpret_cast<const f32x4*>(&sq[qd * 32 + j * 8 + 4]);
            d = fmaf(qa[0], bf_lo(kq[j][0]), d); d = fmaf(qa[1], bf_hi(kq[j][0]), d);
            d = fmaf(qa[2], bf_lo(kq[j][1]), d); d = fmaf(qa[3], bf_hi(kq[j][1]), d);
            d = fmaf(qb[0], bf_lo(kq[j][2]), d); d = fmaf(qb[1], bf_hi(kq[j][2]), d);
            d = fmaf(qb[2], bf_lo(kq[j][3]), d); d = fmaf(qb[3], bf_hi(kq[j][3]), d);
        }
        d += dpp_f<OMNI_DPP_XOR1>(d);
        d += dpp_f<OMNI_DPP_XOR2>(d);
        const float s = t < pos ? d : -INFINITY;
        float m = fmaxf(s, dpp_f<OMNI_DPP_HALF_MIRROR>(s));
        m = fmaxf(m, dpp_f<OMNI_DPP_MIRROR>(m));
        m = xor32_max(xor16_max(m));
        m = fmaxf(m, s_new);
        const float p = exp2f(s - m);
        const float p_new = exp2f(s_new - m);
        float lsum = p + dpp_f<OMNI_DPP_HALF_MIRROR>(p);
        lsum += dpp_f<OMNI_DPP_MIRROR>(lsum);
        lsum = xor32_sum(xor16_sum(lsum));
        const float inv = 1.0f / (lsum + p_new);
        float o0 = p_new * bf_lo(vnew), o1 = p_new * bf_hi(vnew);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float pu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 4 * u));
            o0 = fmaf(pu, bf_lo(vq[u]), o0);
            o1 = fmaf(pu, bf_hi(vq[u]), o1);
        }
        coh_st4(ars, (uint32_t)frag_off(row, h * 128 + 2 * lane, q_heads * 128) * 2, pack_bf2(o0 * inv, o1 * inv));
    }
    CH_STAMP(stamps, sidx, 6);
    chain_gate_arrive(g);
    CH_STAMP(stamps, sidx, 7);
}

// ---- the backbone's input row of batch row b (mtp_finalize_kernel, capi.hip: x[t+1] = bf16(bf16(sum_fp32(e0, emb_1 .. emb_{Q-1})) + text_step), an
// invalid layer-0 id -> a frame of zeros; qwen3_tts_talker.py:1630-1641) by the workgroup that just drew the row's last code: the same 256 threads,
// the same order of additions, the same slab value.  Row-major copy (inputs_embeds: the step's output) by plain stores, the residual stream and its
// slab by write-through stores (the qkv stage behind the flags reads them)
// (a real call, not inlined: compiled into the sampler stage its code changed the register allocation of the whole pass loop -- the all-pass
//  launch ran 4.5 us per pass slower with the tail inlined, profiles/r06 NOTEBOOK)
struct ChainFinArgs { ChainArgs::Tail T; const int32_t* codes; int Q, codebook; };
__device__ __attribute__((noinline)) void chain_finalize_row(const ChainFinArgs fa, int b, int last_code, float* lds) {
    const ChainArgs::Tail& T = fa.T;
    struct { const int32_t* codes; int Q, codebook; } a{fa.codes, fa.Q, fa.codebook};
    const int Q = a.Q, H = T.H, tid = threadIdx.x;
    int* cg = reinterpret_cast<int*>(lds + CH_LDS_FLOATS - 128);          // [64] codes of the row | [4] wave sums (past the sampler's carve)
    float* red = lds + CH_LDS_FLOATS - 64;
    const int c0 = T.input_ids[b];
    const bool invalid0 = c0 < 0 || c0 >= a.codebook;
    if (tid < Q) {
        int c = tid == 0 ? c0 : (tid == Q - 1 ? last_code : (int)coh_ld4(coh_rsrc(a.codes), ((uint32_t)b * Q + tid) * 4));
        if (invalid0) c = 0;
        cg[tid] = c;
        T.audio_codes[(size_t)b * Q + tid] = (int64_t)c;
    }
    __syncthreads();
    float ss = 0.f;
    if (tid < 256) {
        const coh_rsrc_t rrs = coh_rsrc(T.resid);
        for (int v = tid; v < H / 8; v += 256) {
            float s[8];
            {
                const bool ok = c0 >= 0 && c0 < T.vocab;
                const u32x4 e = ok ? ld16(T.embed + (size_t)c0 * H + v * 8) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[2 * j] = bf_lo(e[j]); s[2 * j + 1] = bf_hi(e[j]); }
            }
#pragma unroll 4
            for (int g = 1; g < Q; ++g) {
                const u32x4 e = ld16(T.cp_embed + ((size_t)(g - 1) * a.codebook + cg[g]) * H + v * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[2 * j] += bf_lo(e[j]); s[2 * j + 1] += bf_hi(e[j]); }
            }
            const u32x4 tx = ld16(T.text_step + (size_t)b * H + v * 8);
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = pack_bf2(bfround(s[2 * j]) + bf_lo(tx[j]), bfround(s[2 * j + 1]) + bf_hi(tx[j]));
            *reinterpret_cast<u32x4*>(T.x_out + (size_t)b * H + v * 8) = o;
#pragma unroll
            for (int j = 0; j < 4; ++j) ss += bf_lo(o[j]) * bf_lo(o[j]) + bf_hi(o[j]) * bf_hi(o[j]);
            coh_st16(rrs, (uint32_t)frag_off(b, v * 8, H) * 2, o);
        }
        ss = wave_sum(ss);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
    }
    __syncthreads();
    if (tid == 0) coh_st4(coh_rsrc(T.part), (uint32_t)b * 4, __float_as_uint(red[0] + red[1] + red[2] + red[3]));
}

// ---- sampler stage of pass g: the workgroup that owns batch row b (inside b's flag domain) draws code g of row b from the
// head GEMM's logits and, unless this was the last group, gathers the folded embedding -> projection row of the drawn code
// into the residual stream as the next pass's input (slab 0 = its sum of squares).  sample_kernel's arithmetic (sampler_body.cuh):
// waves 0-3 are the row's 256 threads, waves 4-7 only keep the barriers.
template <bool TAIL = false>
__device__ __forceinline__ void chain_sample(const ChainArgs& a, int g, float* lds, ChainGate& gate, int code, unsigned long long* stamps) {
    const int sidx = 26;
    int* const last_code = reinterpret_cast<int*>(lds + CH_LDS_FLOATS - 136);      // the drawn code, for the whole workgroup (tail: chain_finalize_row)
    CH_STAMP(stamps, sidx, 0);
    const int rows_per_dom = 16 << (gate.dom - 6);
    const int wi = blockIdx.x & ((1 << gate.dom) - 1);
    const int b = (blockIdx.x >> gate.dom) * rows_per_dom + wi;
    const bool has_row = wi < rows_per_dom && b < a.B;              // workgroup-uniform
    const bool live = threadIdx.x < SMP_THREADS;
    const bool more = g < a.Q - 1 && a.ptab != nullptr;
    if (!has_row) {
        chain_gate_skip(gate);
        return;
    }
    CH_STAMP(stamps, sidx, 1);
    chain_gate_wait(gate, code);
    CH_STAMP(stamps, sidx, 2);
    // the pick is wave 0's alone, without a workgroup barrier (sampler_body.cuh smp_pick_wave: round 6, 8.4 -> ~2.5 us per pass); rows that ask
    // for top-p keep the 4-wave pick.  Workgroup-uniform: kernel arguments only
    const bool wave_path = a.greedy || !(a.top_p > 0.f && a.top_p < 1.f);
    if (wave_path) {
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const SmpLds S = smp_carve<8>(lds);
            const coh_rsrc_t lrs = coh_rsrc(a.logits);
            const int V = a.codebook;
            const uint32_t base = (uint32_t)b * a.logits_ld + (uint32_t)(g - 1) * a.logits_pass;
            // a moderate temperature keeps the head GEMM's bf16-exact logits strictly ordered under the division: select first, divide the kept
            const bool late = !a.greedy && a.temperature >= 1.0f / 64 && a.temperature <= 64.0f;
            float xr[32];
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8) {
                const u32x4 v4 = coh_ld16(lrs, (base + j8 * 256 + 4 * lane) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) xr[j8 * 4 + c] = __uint_as_float(v4[c]);
            }
            // (both fix-ups behind wave-uniform branches the optimiser cannot turn into 32 selects: as straight-line code they were 32 IEEE
            //  divisions on every pick)
            if (V < 2048) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int e = 0; e < 32; ++e)
                    if ((e >> 2) * 256 + 4 * lane + (e & 3) >= V) xr[e] = -INFINITY;
            }
            if (!a.greedy && !late) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int e = 0; e < 32; ++e) xr[e] = xr[e] / a.temperature;
            }
            const uint32_t seed = a.row_seed ? a.row_seed[b] : a.seed;
            const uint32_t step = (uint32_t)(a.steps ? a.steps[b] * a.Q + g : g);
            const int pick = smp_pick_wave<32>(xr, V, a.greedy, a.top_k, late, a.temperature, seed, step, S.ckey);
            CH_STAMP(stamps, sidx, 4);
            if (more) {
                // the folded embedding -> projection row of the drawn code: lane l moves 16-byte pieces l and l + 64 of the 2 KB row; the
                // slab value in sample_kernel's order of additions (piece sums, one wave_sum per 64 pieces, then the two halves)
                const int Hc = 1024;
                const uint16_t* tab = a.ptab + ((size_t)(g - 1) * a.codebook + pick) * Hc;
                const coh_rsrc_t rrs = coh_rsrc(a.resid);
                float ss2[2];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int v = lane + 64 * h2;
                    const u32x4 w4 = ld16(tab + v * 8);
                    float ss = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ss += bf_lo(w4[j]) * bf_lo(w4[j]) + bf_hi(w4[j]) * bf_hi(w4[j]);
                    coh_st16(rrs, (uint32_t)frag_off(b, v * 8, Hc) * 2, w4);
                    ss2[h2] = wave_sum(ss);
                }
                if (lane == 0) coh_st4(coh_rsrc(a.part), (uint32_t)b * 4, __float_as_uint(((ss2[0] + ss2[1]) + 0.f) + 0.f));
            }
            if (lane == 0) coh_st4(coh_rsrc(a.codes), ((uint32_t)b * a.Q + g) * 4, (uint32_t)pick);
            if (lane == 0) last_code[0] = pick;
        }
    } else {
        constexpr int NPT = 8;
        const SmpLds S = smp_carve<NPT>(lds);
        const coh_rsrc_t lrs = coh_rsrc(a.logits);
        const int V = a.codebook;
        const uint32_t base = (uint32_t)b * a.logits_ld + (uint32_t)(g - 1) * a.logits_pass;
        float xr[NPT];
#pragma unroll
        for (int e = 0; e < NPT; ++e) {
            const int i = threadIdx.x + e * SMP_THREADS;
            float x = -INFINITY;
            if (live && i < V) {
                x = coh_ldf(lrs, (base + i) * 4);
                if (!a.greedy) x = x / a.temperature;
            }
            xr[e] = x;
        }
        const uint32_t seed = a.row_seed ? a.row_seed[b] : a.seed;
        const uint32_t step = (uint32_t)(a.steps ? a.steps[b] * a.Q + g : g);
        const int pick = smp_pick<NPT>(xr, V, a.greedy, a.top_k, a.top_p, seed, step, S, live, stamps);
        CH_STAMP(stamps, sidx, 4);
        if (more) {
            const int Hc = 1024;
            const uint16_t* tab = a.ptab + ((size_t)(g - 1) * a.codebook + pick) * Hc;
            const coh_rsrc_t rrs = coh_rsrc(a.resid);
            float ss = 0.f;
            if (live) {
                for (int v = threadIdx.x; v < Hc / 8; v += SMP_THREADS) {
                    const u32x4 w4 = ld16(tab + v * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) ss += bf_lo(w4[j]) * bf_lo(w4[j]) + bf_hi(w4[j]) * bf_hi(w4[j]);
                    coh_st16(rrs, (uint32_t)frag_off(b, v * 8, Hc) * 2, w4);
                }
            }
            ss = wave_sum(ss);
            if ((threadIdx.x & 63) == 0) S.sval[threadIdx.x >> 6] = ss;
            __syncthreads();
            if (threadIdx.x == 0) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < SMP_WAVES; ++w) t += S.sval[w];
                coh_st4(coh_rsrc(a.part), (uint32_t)b * 4, __float_as_uint(t));
            }
        }
        if (threadIdx.x == 0) coh_st4(coh_rsrc(a.codes), ((uint32_t)b * a.Q + g) * 4, (uint32_t)pick);
        if (threadIdx.x == 0) last_code[0] = pick;
    }
    if (TAIL && g == a.Q - 1 && a.tail.enabled) {
        // the row's frame is complete: its owner assembles the backbone's input row right here (no hand-off: the last code is its own)
        __syncthreads();
        chain_finalize_row(ChainFinArgs{a.tail, a.codes, a.Q, a.codebook}, b, last_code[0], lds);
    }
    CH_STAMP(stamps, sidx, 6);
    chain_gate_arrive(gate);
    CH_STAMP(stamps, sidx, 7);
}

// ---- positions 0 and 1 of the predictor as stages too (round 4; they were 25 launches + the group-1 head and sampler launches):
// the two-block stream of cp_forward_pair01 (capi.hip) -- rows [0, B) = position 0 (the talker's last hidden state), rows [Bp, Bp + B)
// = position 1 (the layer-0 code embedding), slabs 128 rows wide -- through 5 x { qkv, pair attention, o_proj, gate_up, down_proj },
// then the group-1 head GEMM on the position-1 rows and the sampler (which gathers the input of pass 2 into rows [0, B)).  Tiles =
// the launch path's at 128 rows (pick_tile: qkv 32 x 64, o / down 32 x 16, gate_up 64 x 24): same bits.
// The pair attention: one wave per (row, q head) as attn_pair01_kernel (paged_attn.hip) -- position 0 attends to itself (output = its
// V row), position 1 to both; q / k / v cross from the qkv stage with sc1 dword loads (elements 2 l, 2 l + 1 per lane: V needs no
// other layout; the (l, l + 64) pairing of the norm + RoPE is restored through LDS), K / V of both positions go to the private cache.
__device__ __forceinline__ void chain_attn_pair(const ChainArgs& a, const ChainLayer& L, float* lds, ChainGate& g, int code,
                                                unsigned long long* stamps) {
    const int sidx = ((code & 255) >> 4) * 5 + (code & 15) - 1;
    CH_STAMP(stamps, sidx, 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q_heads = a.q_heads, kv_heads = a.kv_heads, bs = a.bs;
    if ((int)blockIdx.x * 4 >= a.B * q_heads) {
        chain_gate_skip(g);
        return;
    }
    const int pair = blockIdx.x * 4 + (wave & 3);               // waves 4-7 work, wave 0 polls (chain_attn)
    const bool active = wave >= 4 && pair < a.B * q_heads;
    const int row = pair / q_heads, h = pair - row * q_heads;
    const int ratio = q_heads / kv_heads, kvh = h / ratio;
    const int nslots = q_heads + 2 * kv_heads;
    const coh_rsrc_t krs = coh_rsrc(L.kc), vrs = coh_rsrc(L.vc), qrs = coh_rsrc(a.qkv), ars = coh_rsrc(a.attn);
    CH_STAMP(stamps, sidx, 1);
    chain_gate_wait(g, code);
    CH_STAMP(stamps, sidx, 2);
    if (active) {
        const uint32_t r0 = (uint32_t)row * nslots * 128, r1 = (uint32_t)(a.Bp + row) * nslots * 128;
        const uint32_t ko = (uint32_t)(q_heads + kvh) * 128, vo = (uint32_t)(q_heads + kv_heads + kvh) * 128;
        const uint32_t k0w = coh_ld4(qrs, (r0 + ko + 2 * lane) * 2), k1w = coh_ld4(qrs, (r1 + ko + 2 * lane) * 2);
        const uint32_t q1w = coh_ld4(qrs, (r1 + (uint32_t)h * 128 + 2 * lane) * 2);
        const uint32_t v0w = coh_ld4(qrs, (r0 + vo + 2 * lane) * 2), v1w = coh_ld4(qrs, (r1 + vo + 2 * lane) * 2);
        uint32_t* raw = reinterpret_cast<uint32_t*>(lds + (wave & 3) * 256);        // three heads of raw dwords: k(pos 0) | k(pos 1) | q(pos 1)
        raw[lane] = k0w; raw[64 + lane] = k1w; raw[128 + lane] = q1w;
        __builtin_amdgcn_wave_barrier();
        const uint16_t* rh = reinterpret_cast<const uint16_t*>(raw);
        float k00, k01, k10, k11, q0, q1;
        head_norm_rope_vals(bf2f(rh[lane]), bf2f(rh[lane + 64]), L.knorm, a.cos_sin, a.eps, lane, k00, k01);
        head_norm_rope_vals(bf2f(rh[128 + lane]), bf2f(rh[128 + lane + 64]), L.knorm, a.cos_sin + 128, a.eps, lane, k10, k11);
        head_norm_rope_vals(bf2f(rh[256 + lane]), bf2f(rh[256 + lane + 64]), L.qnorm, a.cos_sin + 128, a.eps, lane, q0, q1);
        if (h % ratio == 0) {
            // K / V of both positions into the private cache, one dword per lane (chain_attn's pairing of the two halves)
            const bool odd = (lane & 1) != 0;
            const uint32_t elem = odd ? 64 + lane - 1 : lane;
#pragma unroll
            for (int pz = 0; pz < 2; ++pz) {
                const uint32_t crow = ((uint32_t)(row * bs + pz) * kv_heads + kvh) * 128;
                const uint32_t kb0 = f2bf(pz ? k10 : k00), kb1 = f2bf(pz ? k11 : k01);
                const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kb0, OMNI_DPP_XOR1, 0xF, 0xF, true);
                const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kb1, OMNI_DPP_XOR1, 0xF, 0xF, true);
                coh_st4(krs, (crow + elem) * 2, odd ? (n1 | (kb1 << 16)) : (kb0 | (n0 << 16)));
                coh_st4(vrs, (crow + 2 * lane) * 2, pz ? v1w : v0w);
            }
        }
        const float qs = a.sm_scale * LOG2E;
        const float s0 = wave_sum(fmaf(q0 * qs, k00, (q1 * qs) * k01));
        const float s1 = wave_sum(fmaf(q0 * qs, k10, (q1 * qs) * k11));
        const float m = fmaxf(s0, s1);
        const float p0 = exp2f(s0 - m), p1 = exp2f(s1 - m);
        const float inv = 1.0f / (p0 + p1);
        const int width = q_heads * 128, col = h * 128 + 2 * lane;
        coh_st4(ars, (uint32_t)frag_off(row, col, width) * 2, v0w);
        coh_st4(ars, (uint32_t)frag_off(a.Bp + row, col, width) * 2,
                pack_bf2(fmaf(p1, bf_lo(v1w), p0 * bf_lo(v0w)) * inv, fmaf(p1, bf_hi(v1w), p0 * bf_hi(v0w)) * inv));
    }
    CH_STAMP(stamps, sidx, 6);
    chain_gate_arrive(g);
    CH_STAMP(stamps, sidx, 7);
}

__global__ __launch_bounds__(CH_THREADS) void cp_pair_kernel(const ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[CH_LDS_FLOATS];
    ChainGate g;
    chain_gate_init(g, a.flags, a.err);
    g.dom = 8;                            // 64-row gate_up tiles and the (row, head) pairs tie every row together
    g.nap = a.nap & 31;
    if (a.nap >> 5) g.skip_units = (a.nap >> 5) - 1;
    const int wg = blockIdx.x;
    const int Hc = 1024, NQ = 4096, NI = 3072;
    const int M2 = a.Bp + a.B;
    int np = a.np_in;
    for (int l = 0; l < a.layers; ++l) {
        const ChainLayer& L = a.layer[l];
        chain_gemm<2, 4, 4, 3, OMNI_EPI_BF16, 0, 0, ChainNoPrefetch, false, 128>(L.wqkv, L.ln1, a.resid, a.part, np, a.qkv, NQ, nullptr, M2, NQ, a.eps, wg & 63,
                                                                                 wg >> 6, lds, g, l > 0, 0x0100 | (16 * l + 1), a.stamps);
        chain_attn_pair(a, L, lds, g, 0x0100 | (16 * l + 2), a.stamps);
        chain_gemm<2, 1, 8, 0, OMNI_EPI_RESID, 0, 0, ChainNoPrefetch, false, 128>(L.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, M2, Hc, a.eps, wg & 63,
                                                                                  wg >> 6, lds, g, true, 0x0100 | (16 * l + 3), a.stamps);
        np = Hc / 16;
        chain_gemm<4, 3, 4, 3, OMNI_EPI_SILU_MUL_GU8, 0, 0, ChainNoPrefetch, false, 128>(L.wgu, L.ln2, a.resid, a.part, np, a.act, 0, nullptr, M2, NI, a.eps,
                                                                                         wg & 127, wg >> 7, lds, g, true, 0x0100 | (16 * l + 4), a.stamps);
        chain_gemm<2, 1, 12, 0, OMNI_EPI_RESID, 0, 0, ChainNoPrefetch, false, 128>(L.wdown, nullptr, a.act, nullptr, 0, a.resid, 0, a.part, M2, Hc, a.eps,
                                                                                   wg & 63, wg >> 6, lds, g, true, 0x0100 | (16 * l + 5), a.stamps);
    }
    // group 1: head GEMM on the position-1 rows (fragment-major tiles Bp / 16 onwards; their slabs start at row Bp), then the sampler
    chain_gemm<1, 2, 4, 2, OMNI_EPI_F32_BF16RND, 0, 0, ChainNoPrefetch, false, 128>(a.lm_head, a.cp_norm, a.resid + (size_t)a.Bp * Hc, a.part + a.Bp, np,
                                                                                    a.logits, a.logits_ld, nullptr, a.B, a.codebook, a.eps, wg & 63, wg >> 6,
                                                                                    lds, g, true, 0x0156, a.stamps);
    chain_sample(a, 1, lds, g, 0x01F2, a.stamps);
}

// DEFER: rstd of the qkv / gate_up stages applied in their epilogues (chain_gemm PRO 3; the head GEMM keeps the exact norm) -- round 4's
// A/B arm of VERDICT r3 item 1b, the product form since round 6 (the launch path's gemm_skinny_kernel takes it for the same GEMMs: same bits);
// <.., false> = round 5's exact-rstd stages, debug library only
// the tail's qkv stage as a real call on a copy of the gate (the launch ends behind it: nothing reads the gate afterwards)
template <int KHT>
__device__ __attribute__((noinline)) void chain_tail_qkv(const ChainArgs::Tail T, int B, float eps, float* lds, ChainGate g, int code) {
    g.dom = 8;
    const int nc_q = T.NQ >> 5;
    chain_gemm<2, 2, KHT, 3, OMNI_EPI_BF16, 4>(T.wqkv, T.ln1, T.resid, T.part, 1, T.qkv, T.NQ, nullptr, B, T.NQ, eps, blockIdx.x % nc_q, blockIdx.x / nc_q, lds, g,
                                               true, code, nullptr);
}

// KHT: 0, or hidden / 256 of the backbone whose first qkv stage closes the launch (the tail: round 6)
template <bool GU_NARROW, bool DEFER = true, int KHT = 0>
__global__ __launch_bounds__(CH_THREADS) void cp_chain_kernel(const ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[CH_LDS_FLOATS];
    static_assert(CH_LDS_FLOATS * 4 >= SMP_LDS_BYTES(8), "sampler working set must fit the chain's LDS");
    ChainGate g;
    chain_gate_init(g, a.flags, a.err);
    g.dom = a.dom;
    g.nap = a.nap & 31;
    if (a.nap >> 5) g.skip_units = (a.nap >> 5) - 1;      // debug library: nap + 32 (units + 1) selects the run-ahead sleep (coherent.cuh)
    g.skip = a.skip;
    const int wg = blockIdx.x;
    {
        // a partly filled batch: a workgroup that owns no live row in ANY stage of the launch (its 16-row group, its 32-row gate_up
        // pair, its four attention pairs, its sampler row) publishes the launch's final stage count at once and leaves -- the others
        // find its flag ahead of every stage they wait for, and it never polls
        const int rows_per_dom = 16 << (a.dom - 6), wi = wg & ((1 << a.dom) - 1);
        const bool gemm16 = (wg >> 6) * 16 < a.B, gemm32 = GU_NARROW && (wg >> 7) * 32 < a.B, attn_ = wg * 4 < a.B * a.q_heads;
        const bool smp = a.with_head && wi < rows_per_dom && (wg >> a.dom) * rows_per_dom + wi < a.B;
        if (!(gemm16 || gemm32 || attn_ || smp)) {
            const uint32_t total = (uint32_t)(a.g1 - a.g0) * (uint32_t)(a.layers * 5 + (a.with_head ? 2 : 0)) + ((KHT > 0 && a.tail.enabled) ? 1u : 0u);
            if (threadIdx.x < 64) chain_flag_publish(g.frs, wg, g.epoch + total);
            return;
        }
    }
    int np = a.np_in;
    const int Hc = 1024, NQ = 4096, NI = 3072;
    for (int pass = a.g0; pass < a.g1; ++pass) {
        const int pc = pass << 8;            // error-word stage codes: (pass << 8) | (16 * layer + stage + 1); head 0xF1, sampler 0xF2
        // debug library: every pass stamps its own block of CH_STAMP_PASS words (round 6: the cold tail of the passes in the real step)
        unsigned long long* const st = a.stamps ? a.stamps + (size_t)pass * CH_STAMP_PASS : nullptr;
        for (int l = 0; l < a.layers; ++l) {
            const ChainLayer& L = a.layer[l];
            // the workgroup index, opaque per layer: hipcc otherwise hoists every tile / k-step offset of every stage (they depend on
            // blockIdx only) out of both loops and parks ~450 of them in VGPR lanes for the whole launch (v_writelane / v_readlane)
            int wg = blockIdx.x;
            asm volatile("" : "+s"(wg));
            chain_gemm<1, 4, 4, DEFER ? 3 : 2, OMNI_EPI_BF16>(L.wqkv, L.ln1, a.resid, a.part, np, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg & 63, wg >> 6, lds, g,
                                                              l > 0 || pass > a.g0, pc | (16 * l + 1), st);
            chain_attn(a, L, pass, lds, g, pc | (16 * l + 2), st);
            chain_gemm<1, 1, 8, 0, OMNI_EPI_RESID>(L.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, a.B, Hc, a.eps, wg & 63, wg >> 6, lds, g,
                                                   true, pc | (16 * l + 3), st);
            np = Hc / 16;
            if (GU_NARROW)
                chain_gemm<2, 3, 4, DEFER ? 3 : 2, OMNI_EPI_SILU_MUL_GU8>(L.wgu, L.ln2, a.resid, a.part, np, a.act, 0, nullptr, a.B, NI, a.eps, wg & 127, wg >> 7,
                                                                          lds, g, true, pc | (16 * l + 4), st);
            else   // 16 rows x 48 act columns: half the in-register RMSNorm per workgroup, a weight slice twice as wide (prefetched)
                chain_gemm<1, 6, 4, DEFER ? 3 : 2, OMNI_EPI_SILU_MUL_GU8>(L.wgu, L.ln2, a.resid, a.part, np, a.act, 0, nullptr, a.B, NI, a.eps, wg & 63, wg >> 6,
                                                                          lds, g, true, pc | (16 * l + 4), st);
            chain_gemm<1, 1, 12, 0, OMNI_EPI_RESID>(L.wdown, nullptr, a.act, nullptr, 0, a.resid, 0, a.part, a.B, Hc, a.eps, wg & 63, wg >> 6, lds, g,
                                                    true, pc | (16 * l + 5), st);
        }
        if (!a.with_head) continue;
        int wg = blockIdx.x;
        asm volatile("" : "+s"(wg));
        // group `pass`: final norm folded into the head GEMM (2048 logits per row), then one workgroup per row samples
        chain_gemm<1, 2, 4, 2, OMNI_EPI_F32_BF16RND>(a.lm_head + (size_t)(pass - 1) * a.codebook * Hc, a.cp_norm, a.resid, a.part, np,
                                                     a.logits + (size_t)(pass - 1) * a.logits_pass, a.logits_ld, nullptr, a.B, a.codebook, a.eps,
                                                     wg & 63, wg >> 6, lds, g, true, pc | 0x56, st);
        chain_sample<(KHT > 0)>(a, pass, lds, g, pc | 0xF2, st);
        np = 1;
    }
    if constexpr (KHT > 0) {
        if (a.tail.enabled) {
            // layer 0's qkv rows of the backbone, behind EVERY row's input assembly (flag domain: all 256 workgroups); gemm_skinny_kernel's
            // norm-fused arithmetic on 32 x 32 tiles (the bits do not depend on the tile shape: gemm_frag.cuh SlabOrder)
            chain_tail_qkv<KHT>(a.tail, a.B, a.eps, lds, g, ((a.g1 - 1) << 8) | 0x62);
        }
    }
}

// ---- host
OMNI_KNOB g_cp_chain = 1, g_chain_dom = 7, g_chain_gu_narrow = 1, g_chain_nap = 1, g_chain_span = 2, g_chain_skip = 0, g_chain_defer = 1, g_chain_pair = 1;
#ifdef OMNI_DEBUG_HOOKS
extern "C" void omni_debug_chain_mode(int dom, int gu_narrow, int nap) { g_chain_dom = dom; g_chain_gu_narrow = gu_narrow; g_chain_nap = nap; }
extern "C" void omni_debug_chain_skip(int mode) { g_chain_skip = mode; }
extern "C" void omni_debug_chain_defer(int on) { g_chain_defer = on; }
extern "C" void omni_debug_chain_pair(int on) { g_chain_pair = on; }            // positions 0 / 1 + group 1 as a persistent launch (0: the launch path's pair pass)          // timing arm: rstd in the epilogues (not the reference's rounding)      // timing experiment: 1 = half the weight bytes, 2 = half the activation bytes (results garbage)
static unsigned long long* g_chain_stamps = nullptr;
// 0: launch per op; 1: one persistent launch per pass (layer stack only); 2: one persistent launch for all passes incl. heads + samplers
extern "C" void omni_debug_cp_chain(int on) { g_cp_chain = on != 0; g_chain_span = on; }
// device buffer of 16 blocks (one per predictor pass; block 0: the pair kernel) of 40 * CH_NSTAMP * 256 uint64 that every following chain
// launch overwrites (NULL: off)
extern "C" void omni_debug_chain_stamps(void* buf) { g_chain_stamps = (unsigned long long*)buf; }
#endif

bool k_cp_chain_supported(const omni_talker_desc& d, int pos) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 0;
    }
    return g_cp_chain && cus >= OMNI_CHAIN_WGS && d.cp_fused_norm && d.frag_layout && d.cp_hidden == 1024 && d.cp_head_dim == 128 &&
           d.cp_q_heads * 128 == 2048 && (d.cp_q_heads + 2 * d.cp_kv_heads) * 128 == 4096 && d.cp_inter == 3072 &&
           d.cp_layers >= 1 && d.cp_layers <= CH_MAX_LAYERS && pos >= 1 && pos <= 15 && d.num_code_groups + 1 >= 16 &&
           d.max_batch <= 64;
}
// all passes g0 .. Q - 1 with their heads and samplers in ONE launch: additionally the 2048-entry codebook the in-chain sampler is
// built for, the folded projection tables (the next pass's input is a gather) and top-k within the sampler's fast path
bool k_cp_chain_all_supported(const omni_talker_desc& d, int g0, int greedy, int top_k, float top_p) {
    return g_chain_span >= 2 && k_cp_chain_supported(d, g0) && d.codebook == 2048 && d.cp_proj_table != nullptr && d.cp_lm_head != nullptr &&
           d.num_code_groups <= 16 && (greedy || !(top_p > 0.f && top_p < 1.f) || (top_k > 0 && top_k <= SMP_PCAP));
}

// positions 0 / 1 of every row + group 1's head and sampler as ONE persistent launch (cp_pair_kernel); supported where the all-pass
// chain is, with the projected inputs already in the two-block stream (capi.hip run_code_predictor)
bool k_cp_pair_supported(const omni_talker_desc& d, int B, int greedy, int top_k, float top_p) {
    return g_chain_pair && k_cp_chain_all_supported(d, 1, greedy, top_k, top_p) && d.cp_layers <= CH_MAX_LAYERS && B >= 1 && ((B + 15) & ~15) + B <= 128;
}
int k_cp_pair(const omni_talker_desc& d, const omni_layer_weights* layers, uint16_t* const* k_cache, uint16_t* const* v_cache, int B, int np_in,
              uint16_t* resid, float* part, uint16_t* qkv, uint16_t* attn, uint16_t* act, uint32_t* flags, int32_t* err, const omni_chain_head* head,
              void* stream) {
    ChainArgs a{};
    for (int l = 0; l < d.cp_layers; ++l) {
        const omni_layer_weights& w = layers[l];
        a.layer[l] = ChainLayer{(const uint16_t*)w.ln1, (const uint16_t*)w.wqkv, (const uint16_t*)w.qnorm, (const uint16_t*)w.knorm,
                                (const uint16_t*)w.wo, (const uint16_t*)w.ln2, (const uint16_t*)w.wgu, (const uint16_t*)w.wdown,
                                k_cache[l], v_cache[l]};
    }
    a.layers = d.cp_layers; a.B = B; a.Bp = (B + 15) & ~15; a.g0 = 0; a.g1 = 2; a.bs = d.num_code_groups + 1; a.np_in = np_in;
    a.q_heads = d.cp_q_heads; a.kv_heads = d.cp_kv_heads;
    a.eps = d.eps; a.sm_scale = 1.0f / sqrtf((float)d.cp_head_dim);
    a.resid = resid; a.part = part; a.qkv = qkv; a.attn = attn; a.act = act;
    a.cos_sin = (const uint16_t*)d.cp_cos_sin;
    a.flags = flags; a.err = err;
    a.with_head = 1;
    a.cp_norm = (const uint16_t*)d.cp_norm; a.lm_head = (const uint16_t*)d.cp_lm_head;
    a.logits = head->logits; a.logits_ld = head->logits_ld; a.logits_pass = head->logits_pass;
    a.greedy = head->greedy; a.top_k = head->top_k; a.Q = d.num_code_groups; a.codebook = d.codebook;
    a.temperature = head->temperature; a.top_p = head->top_p; a.seed = head->seed;
    a.steps = head->steps; a.row_seed = head->row_seed; a.codes = head->codes;
    a.ptab = (const uint16_t*)d.cp_proj_table;
    a.dom = 8; a.nap = g_chain_nap;
#ifdef OMNI_DEBUG_HOOKS
    a.stamps = g_chain_stamps;            // block 0 of the stamp buffer (the chain's passes 2 .. 15 stamp blocks 2 .. 15)
#endif
    hipLaunchKernelGGL(cp_pair_kernel, dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), 0, (hipStream_t)stream, a);
    OMNI_CHECK_LAUNCH("cp_pair");
    return OMNI_OK;
}

// backbone widths (hidden / 256) the tail's qkv stage is instantiated for: those of bb_chain.hip's BB_SHAPES
#define CP_TAIL_WIDTHS(X) X(8) X(6)
OMNI_KNOB g_chain_tail = 1;
#ifdef OMNI_DEBUG_HOOKS
extern "C" void omni_debug_chain_tail(int on) { g_chain_tail = on; }      // 0: mtp_finalize and the first qkv as launches of their own (round 5)
#endif
bool k_cp_chain_tail_supported(const omni_talker_desc& d, int B) {
    bool listed = false;
#define X(KH_) listed = listed || d.hidden == KH_ * 256;
    CP_TAIL_WIDTHS(X)
#undef X
    const int NQ = (d.q_heads + 2 * d.kv_heads) * 128;
    // every row group live (B > 48: no workgroup leaves at launch start, all of them own a qkv tile with rows), the norm-free stream, the
    // qkv stage's 32 x 32 tile grid inside the 256 workgroups
    return g_chain_tail && g_chain_defer && g_chain_gu_narrow && g_cp_chain && g_chain_span >= 2 && listed && B > 48 && B <= 64 && d.fused_norm && d.frag_layout && d.moe_experts == 0 && d.head_dim == 128 &&
           NQ % 32 == 0 && (NQ / 32) * 2 <= OMNI_CHAIN_WGS && d.hidden % 8 == 0;
}

int k_cp_chain(const omni_talker_desc& d, const omni_layer_weights* layers, uint16_t* const* k_cache, uint16_t* const* v_cache, int B,
               int g0, int g1, int np_in, uint16_t* resid, float* part, uint16_t* qkv, uint16_t* attn, uint16_t* act, uint32_t* flags,
               int32_t* err, const omni_chain_head* head, void* stream) {
    ChainArgs a{};
    for (int l = 0; l < d.cp_layers; ++l) {
        const omni_layer_weights& w = layers[l];
        a.layer[l] = ChainLayer{(const uint16_t*)w.ln1, (const uint16_t*)w.wqkv, (const uint16_t*)w.qnorm, (const uint16_t*)w.knorm,
                                (const uint16_t*)w.wo, (const uint16_t*)w.ln2, (const uint16_t*)w.wgu, (const uint16_t*)w.wdown,
                                k_cache[l], v_cache[l]};
    }
    a.layers = d.cp_layers; a.B = B; a.g0 = g0; a.g1 = g1; a.bs = d.num_code_groups + 1; a.np_in = np_in;
    a.q_heads = d.cp_q_heads; a.kv_heads = d.cp_kv_heads;
    a.eps = d.eps; a.sm_scale = 1.0f / sqrtf((float)d.cp_head_dim);
    a.resid = resid; a.part = part; a.qkv = qkv; a.attn = attn; a.act = act;
    a.cos_sin = (const uint16_t*)d.cp_cos_sin;
    a.flags = flags; a.err = err;
    if (head) {
        a.with_head = 1;
        a.cp_norm = (const uint16_t*)d.cp_norm; a.lm_head = (const uint16_t*)d.cp_lm_head;
        a.logits = head->logits; a.logits_ld = head->logits_ld; a.logits_pass = head->logits_pass;
        a.greedy = head->greedy; a.top_k = head->top_k; a.Q = d.num_code_groups; a.codebook = d.codebook;
        a.temperature = head->temperature; a.top_p = head->top_p; a.seed = head->seed;
        a.steps = head->steps; a.row_seed = head->row_seed; a.codes = head->codes;
        a.ptab = (const uint16_t*)d.cp_proj_table;
    }
    a.dom = g_chain_dom < 6 ? 6 : (g_chain_dom > 8 ? 8 : g_chain_dom);
    // (32-row gate_up tiles from 49 rows: at 33-48 rows the launch path takes 16-row tiles too since round 5 -- gemm.hip pick_tile -- so
    //  that the fourth row group has no work in any stage and its workgroups leave at the start of the launch)
    const bool narrow = g_chain_gu_narrow == 2 || (g_chain_gu_narrow && B > 48);      // 2 (debug): the 32-row tile at any batch size
    if (narrow && a.dom < 7) a.dom = 7;  // the 32-row tile ties two row groups together
    a.skip = g_chain_skip;
    // gate_up's RMSNorm statistics are summed in an order that depends on the rows per tile: follow the launch path's tile
    // policy (32-row tiles only above 32 rows) so that both schedules produce the same bits
    a.gu_narrow = narrow;
    a.nap = g_chain_nap;
#ifdef OMNI_DEBUG_HOOKS
    a.stamps = g_chain_stamps;
#endif
#ifdef OMNI_DEBUG_HOOKS
    if (!g_chain_defer) {      // timing arm: round 5's exact rstd (its bits are no longer the launch path's)
        if (a.gu_narrow) hipLaunchKernelGGL((cp_chain_kernel<true, false>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((cp_chain_kernel<false, false>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), 0, (hipStream_t)stream, a);
        OMNI_CHECK_LAUNCH("cp_chain(exact rstd)");
        return OMNI_OK;
    }
#endif
    if (head && head->tail && g1 == d.num_code_groups && a.gu_narrow && k_cp_chain_tail_supported(d, B)) {
        // the launch ends in the step's input assembly + layer 0's qkv (instantiated per backbone width: CP_TAIL_WIDTHS)
        const omni_chain_tail& T = *head->tail;
        a.tail = ChainArgs::Tail{T.input_ids, (const uint16_t*)T.embed, T.vocab, (const uint16_t*)T.cp_embed, (const uint16_t*)T.text_step,
                                 (uint16_t*)T.x_out, (uint16_t*)T.resid, T.part, T.audio_codes, T.H,
                                 (const uint16_t*)T.wqkv, (const uint16_t*)T.ln1, (uint16_t*)T.qkv, T.NQ, 1};
#define X(KH_) if (d.hidden == KH_ * 256) { hipLaunchKernelGGL((cp_chain_kernel<true, true, KH_>), dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), 0, (hipStream_t)stream, a); OMNI_CHECK_LAUNCH("cp_chain(tail)"); return OMNI_OK; }
        CP_TAIL_WIDTHS(X)
#undef X
    }
    if (a.gu_narrow) hipLaunchKernelGGL(cp_chain_kernel<true>, dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(cp_chain_kernel<false>, dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), 0, (hipStream_t)stream, a);
    OMNI_CHECK_LAUNCH("cp_chain");
    return OMNI_OK;
}
