// The backbone segment o_proj -> gate_up -> down_proj -> next qkv (bb_chain.hip) with the two operand streams of a stage on
// DIFFERENT waves of the workgroup:
//   * 8 compute waves: weights only.  A ring of G k-steps of the wave's weight share in registers (the k-step ownership of the
//     launch path: wave w owns k-steps w, w + 8, ...), refilled as the MFMAs retire them; activation fragments come from LDS.
//     A compute wave never issues an activation load, a poll or a store to global memory: its in-order vector-memory queue
//     holds nothing but the HBM stream, and the next stage's ring goes out the moment its MFMAs are done.
//   * 4 service waves: everything else.  Poll the stage flags, reduce the sum(r^2) slabs to rstd, fetch the activation
//     fragments (sc1 loads, L2-served: a queue with nothing slow in it), apply the RMSNorm once per fragment (the compute
//     waves' MFMA loop carries no VALU normalisation), publish them round by round (8 k-steps) into an LDS ring; afterwards
//     combine the compute waves' K-partials from LDS, run the epilogue, drain the stores and publish the workgroup's flag.
// Why: in the plain chain a wave's activation loads (L2 hits, ~0.7 us) sit in the same in-order queue as its weight loads (HBM,
// ~2 us loaded), so everything returns at HBM latency: ~112 KB in flight per CU / 2 us = 56 GB/s per CU for W + x together
// (1.47 MB per CU and layer: 27 us), and the stream stops at every hand-off.  (bb_engine.hip moved the WEIGHTS to loader waves
// and an LDS FIFO -- the compute waves' queues then still carried activation loads, polls and stores; bb_pp.hip alternated two
// groups with complete stages each.)
// Arithmetic = the launch path's: same k-step ownership, MFMA order, combine order over the 8 compute waves, per-fragment
// normalisation (gemm_frag.cuh), slab reduction in the launch kernel's channel order.  Synchronisation inside the workgroup is
// LDS words only (s_barrier would stop both roles): monotonic counters, relaxed polls, every wait bounded.
#include "chain_gemm.cuh"
#include "common.cuh"
#include "kernels.h"

#define XW_CW 8                         // compute waves
#define XW_SW 4                         // service waves
#define XW_ST (XW_SW * 64)              // service threads
#define XW_THREADS ((XW_CW + XW_SW) * 64)
#define XW_RING_BYTES (96 * 1024)       // activation ring; the K-partials of a stage reuse it once its activations are consumed
#define XW_SPIN_BOUND (1u << 24)

struct XwSync {
    unsigned bar_c, bar_s;              // barrier arrivals: compute group, service group
    unsigned x_ready;                   // activation rounds published (all stages of the launch, monotonic)
    unsigned part_ready;                // compute waves that have written their K-partials (monotonic: 8 per stage)
    unsigned x_done[XW_CW];             // per compute wave: activation rounds it has read
    unsigned dead;
    unsigned pad[3];
};

struct XwCtx {
    XwSync* s;
    uint8_t* ring;
    float* red;                         // [8][64] slab-reduction scratch (service waves)
    unsigned bgen;                      // barriers of MY group passed
    bool service;
    // cross-workgroup gate (service waves)
    coh_rsrc_t frs; uint32_t base; int32_t* err; bool dead; int nap;
};

__device__ __forceinline__ unsigned xw_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void xw_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// wait until *p has reached `target` (monotonic, wrap-safe); LDS words, relaxed
__device__ __forceinline__ void xw_wait(XwCtx& c, const unsigned* p, unsigned target) {
    unsigned spins = 0;
    while ((int)(xw_ld(p) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > XW_SPIN_BOUND) {
            xw_st(&c.s->dead, 1u);
            break;
        }
    }
    asm volatile("" ::: "memory");
}

// barrier among the waves of my group (LDS operations of a wave execute in order: only the compiler must keep the order)
__device__ __forceinline__ void xw_barrier(XwCtx& c) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    c.bgen += 1;
    unsigned* b = c.service ? &c.s->bar_s : &c.s->bar_c;
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    xw_wait(c, b, c.bgen * (c.service ? XW_SW : XW_CW));
}

// ---- compute waves: one stage.  NTW k-steps per wave, ring of G; rounds_base = activation rounds published before this stage
template <int MT, int NT, int NTW, int G>
__device__ __forceinline__ void xw_compute(const uint16_t* __restrict__ W, int bx, XwCtx& c, unsigned rounds_base, unsigned stage_idx) {
    constexpr int nsteps = NTW * XW_CW;
    constexpr int RING = XW_RING_BYTES / (XW_CW * MT * 1024);         // rounds the activation ring holds
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane16 = lane * 16;
    const coh_rsrc_t wrs = coh_rsrc(W);
    u32x4 Wq[G][NT];
    auto load_w = [&](int d) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
            Wq[d % G][j] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (uint32_t)(((bx * NT + j) * nsteps + wave + d * XW_CW) * 1024), 0);
    };
#pragma unroll
    for (int d = 0; d < (G < NTW ? G : NTW); ++d) load_w(d);
    f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < NTW; ++d) {
        xw_wait(c, &c.s->x_ready, rounds_base + d + 1);                  // round d of this stage is in the ring
        u32x4 Xn[MT], Wn[NT];
        const uint8_t* slot = c.ring + (size_t)((d % RING) * XW_CW + wave) * (MT * 1024);
#pragma unroll
        for (int i = 0; i < MT; ++i) Xn[i] = *reinterpret_cast<const u32x4*>(slot + i * 1024 + lane16);
#pragma unroll
        for (int j = 0; j < NT; ++j) Wn[j] = Wq[d % G][j];
        if (d + G < NTW) load_w(d + G);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the fragments are in registers: the slot is free for me
        if (lane == 0) xw_st(&c.s->x_done[wave], rounds_base + d + 1);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[j][i] = mfma16(Wn[j], Xn[i], acc[j][i]);
    }
    // K-partials into the ring area (its activations are consumed once EVERY compute wave is here)
    xw_barrier(c);
    f32x4* p4 = reinterpret_cast<f32x4*>(c.ring);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) p4[(wave * (NT * MT) + j * MT + i) * 64 + lane] = acc[j][i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&c.s->part_ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    (void)stage_idx;
}

// ---- service waves: one stage
template <int MT, int NT, int NTW, int PRO, int EPI>
__device__ __forceinline__ void xw_service(const uint16_t* __restrict__ norm_w, const uint16_t* x, const float* part_in, void* out, int ldo,
                                           float* part_out, int M, int N, float eps, int bx, int by, XwCtx& c, unsigned rounds_base,
                                           unsigned stage_idx, uint32_t wait_stages, uint32_t done_stages, int code) {
    constexpr int nsteps = NTW * XW_CW, K = nsteps * 32;
    constexpr int RING = XW_RING_BYTES / (XW_CW * MT * 1024);
    constexpr bool GU8 = EPI == OMNI_EPI_SILU_MUL_GU8;
    constexpr int FPW = XW_CW * MT / XW_SW;                              // fragments per service wave and round
    static_assert(XW_CW * MT % XW_SW == 0 && (EPI == OMNI_EPI_BF16 || EPI == OMNI_EPI_RESID || GU8), "xw_service");
    const int lane = threadIdx.x & 63;
    const int sw = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) - XW_CW);
    const int tg = threadIdx.x - XW_CW * 64;
    const int q = lane >> 4;
    const int m_base = by * (MT * 16);
    const int Mloc = min(M - m_base, MT * 16);
    const coh_rsrc_t xrs = coh_rsrc(x), ors = coh_rsrc(out), nrs = coh_rsrc(PRO == 2 ? norm_w : x);
    const uint32_t lane16 = lane * 16;

    // ---- ahead of the flags: the old residual values of this thread's epilogue item
    u32x2 r_old = (u32x2){0u, 0u};
    if (EPI == OMNI_EPI_RESID && tg < MT * 64) {
        const int ml = (tg >> 6) * 16 + (lane & 15);
        if (ml < Mloc) r_old = coh_ld8(ors, (uint32_t)frag_off(m_base + ml, bx * 16 + 4 * (lane >> 4), N) * 2);
    }
    // ---- the stage's flags
    if (wait_stages) {
        if (sw == 0 && !c.dead) {
            const uint32_t want = c.base + wait_stages;
            unsigned spins = 0;
            for (;;) {
                asm volatile("" ::: "memory");
                const u32x4 f = coh_ld16(c.frs, lane16);
                const bool behind = (int)(f[0] - want) < 0 || (int)(f[1] - want) < 0 || (int)(f[2] - want) < 0 || (int)(f[3] - want) < 0;
                if (__builtin_amdgcn_ballot_w64(behind) == 0) break;
                if (c.nap <= 1) __builtin_amdgcn_s_sleep(1);
                else __builtin_amdgcn_s_sleep(2);
                if (++spins > OMNI_CHAIN_SPIN_BOUND) {
                    if (lane == 0) atomicCAS(c.err, 0, code);
                    c.dead = true;
                    break;
                }
            }
        }
        xw_barrier(c);
    }
    // ---- slabs -> rstd, in the launch kernel's order: channel ch sums slabs ch, ch + NCH, ...; channels of one launch wave are
    // added (XROWS <= 32), then the 8 wave sums in wave order
    float rstd_f[FPW];
    if (PRO == 2) {
        constexpr int XROWS = MT * 16, NCH = 512 / XROWS, PE = 128 / NCH, CPT = NCH / (XW_ST / XROWS);   // channels per service thread
        static_assert(CPT == 2, "xw_service: two launch channels per service thread");
        const coh_rsrc_t prs = coh_rsrc(part_in);
        const int row = tg % XROWS, g_ = tg / XROWS;
        float pv[2][PE];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ch = XROWS == 64 ? g_ + 4 * h : 2 * g_ + h;        // 64 rows: channels g, g + 4; 32 rows: the pair 2 g, 2 g + 1
#pragma unroll
            for (int e = 0; e < PE; ++e) pv[h][e] = coh_ldf(prs, (uint32_t)((ch + e * NCH) * 64 + m_base + row) * 4);
        }
        float s_[2] = {0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < PE; ++e) s_[h] += pv[h][e];
        if (XROWS == 64) {
            c.red[g_ * 64 + row] = s_[0];
            c.red[(g_ + 4) * 64 + row] = s_[1];
        } else {
            c.red[g_ * 64 + row] = s_[0] + s_[1];                        // = xor32_sum of the launch kernel's wave g
        }
        xw_barrier(c);
        // fragment f of a round (k-step f / MT, m-tile f % MT): this lane's row is (f % MT) * 16 + (lane & 15)
#pragma unroll
        for (int n = 0; n < FPW; ++n) {
            const int i = (sw + n * XW_SW) % MT;
            const int r = i * 16 + (lane & 15);
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += c.red[w * 64 + r];
            rstd_f[n] = 1.0f / sqrtf(t / (float)K + eps);
        }
    }
    // ---- activation rounds: wave sw moves fragments f = sw, sw + 4, ... of every round (8 k-steps x MT m-tiles), two rounds of
    // loads in flight
    u32x4 Xa[2][FPW], Na[2][FPW];
    auto issue = [&](int r) {
#pragma unroll
        for (int n = 0; n < FPW; ++n) {
            const int f = sw + n * XW_SW, ks = r * XW_CW + f / MT, i = f % MT;
            Xa[r & 1][n] = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane16, (uint32_t)((((m_base >> 4) + i) * nsteps + ks) * 1024), OMNI_AUX_SC1);
            if (PRO == 2) Na[r & 1][n] = __builtin_amdgcn_raw_buffer_load_b128(nrs, q * 16, ks * 64, 0);
        }
    };
    issue(0);
#pragma unroll
    for (int r = 0; r < NTW; ++r) {
        if (r + 1 < NTW) issue(r + 1);
        if (r >= RING) {                                                 // the slot's previous round has been read by every compute wave
#pragma unroll
            for (int w = 0; w < XW_CW; ++w) xw_wait(c, &c.s->x_done[w], rounds_base + r - RING + 1);
        }
        uint8_t* slot = c.ring + (size_t)(r % RING) * (XW_CW * MT * 1024);
#pragma unroll
        for (int n = 0; n < FPW; ++n) {
            const int f = sw + n * XW_SW;
            u32x4 v = Xa[r & 1][n];
            if (PRO == 2) v = xnorm_frag(v, Na[r & 1][n], rstd_f[n]);
            *reinterpret_cast<u32x4*>(slot + (size_t)f * 1024 + lane16) = v;
        }
        xw_barrier(c);                                                   // all four waves' fragments of the round are written
        if (tg == 0) xw_st(&c.s->x_ready, rounds_base + r + 1);
    }
    // ---- combine the 8 K-partials (wave order), epilogue with write-through stores
    xw_wait(c, &c.s->part_ready, (stage_idx + 1) * XW_CW);
    const f32x4* p4 = reinterpret_cast<const f32x4*>(c.ring);
    constexpr int TT = NT * MT;
    constexpr int LN = GU8 ? 32 : 64;
    constexpr int ITEMS = TT * LN;
    for (int it = tg; it < ITEMS; it += XW_ST) {
        const int l = it % LN;
        const int t = it / LN;
        const int i = t % MT, j = t / MT;
        const int ml = i * 16 + (l & 15);
        if (ml >= Mloc) continue;
        const int m = m_base + ml;
        f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f}, sum2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < XW_CW; ++w) {
            sum += p4[(w * TT + t) * 64 + l];
            if (GU8) sum2 += p4[(w * TT + t) * 64 + l + 32];
        }
        if (GU8) {
            const int n = (bx * NT + j) * 8 + 4 * (l >> 4);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = silu_mul_bf16(sum[e], sum2[e]);
            coh_st8(ors, (uint32_t)frag_off(m, n, N) * 2, (u32x2){pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])});
        } else if (EPI == OMNI_EPI_RESID) {
            const int n = bx * 16 + 4 * (l >> 4);
            float rv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) rv[e] = bfround(sum[e]);
            rv[0] = bfround(bf_lo(r_old[0]) + rv[0]); rv[1] = bfround(bf_hi(r_old[0]) + rv[1]);
            rv[2] = bfround(bf_lo(r_old[1]) + rv[2]); rv[3] = bfround(bf_hi(r_old[1]) + rv[3]);
            coh_st8(ors, (uint32_t)frag_off(m, n, N) * 2, (u32x2){pack_bf2(rv[0], rv[1]), pack_bf2(rv[2], rv[3])});
            float ss = rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2] + rv[3] * rv[3];
            ss = xor32_sum(xor16_sum(ss));
            if (l < 16) coh_st4(coh_rsrc(part_out), (uint32_t)(bx * 64 + m) * 4, __float_as_uint(ss));
        } else {
            const int n = (bx * NT + j) * 16 + 4 * (l >> 4);
            coh_st8(ors, (uint32_t)((size_t)m * ldo + n) * 2, (u32x2){pack_bf2(sum[0], sum[1]), pack_bf2(sum[2], sum[3])});
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    xw_barrier(c);                                                       // every service wave's stores are out; the partials are read
    if (tg < 64) chain_flag_publish(c.frs, blockIdx.x, c.base + done_stages);
}

struct XwArgs {
    const uint16_t *wo, *ln2, *wgu, *wdown, *ln1_next, *wqkv_next;      // *_next == NULL: last layer, no qkv stage
    const uint16_t* attn; uint16_t* resid; float* part; uint16_t* act; uint16_t* qkv;
    int B, nap; float eps;
    uint32_t* flags; int32_t* err;
};

#define XW_LDS_BYTES (XW_RING_BYTES + 8 * 64 * 4 + 128)

__global__ __launch_bounds__(XW_THREADS) void bb_xw_kernel(const XwArgs a) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    XwCtx c;
    c.ring = lds;
    c.red = reinterpret_cast<float*>(lds + XW_RING_BYTES);
    c.s = reinterpret_cast<XwSync*>(lds + XW_RING_BYTES + 8 * 64 * 4);
    if (threadIdx.x < sizeof(XwSync) / 4) reinterpret_cast<unsigned*>(c.s)[threadIdx.x] = 0u;
    __syncthreads();                                   // the only s_barrier
    c.bgen = 0;
    c.service = threadIdx.x >= XW_CW * 64;
    c.frs = coh_rsrc(a.flags);
    c.err = a.err;
    c.base = __builtin_amdgcn_readfirstlane(coh_ld4(c.frs, blockIdx.x * 4));
    c.dead = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0;
    c.nap = a.nap;
    const int wg = blockIdx.x;
    constexpr int H = 2048, I = 6144, NQ = 4096;
    const bool has_qkv = a.wqkv_next != nullptr;
    // activation rounds per stage: o 8, gate_up 8, down 24, qkv 8
    if (!c.service) {
        xw_compute<2, 1, 8, 8>(a.wo, wg & 127, c, 0u, 0u);
        xw_compute<4, 3, 8, 4>(a.wgu, wg, c, 8u, 1u);
        xw_compute<2, 1, 24, 12>(a.wdown, wg & 127, c, 16u, 2u);
        if (has_qkv) xw_compute<2, 2, 8, 8>(a.wqkv_next, wg & 127, c, 40u, 3u);
    } else {
        xw_service<2, 1, 8, 0, OMNI_EPI_RESID>(nullptr, a.attn, nullptr, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, c, 0u, 0u, 0u, 1u, 0x3001);
        xw_service<4, 3, 8, 2, OMNI_EPI_SILU_MUL_GU8>(a.ln2, a.resid, a.part, a.act, 0, nullptr, a.B, I, a.eps, wg, 0, c, 8u, 1u, 1u, 2u, 0x3002);
        xw_service<2, 1, 24, 0, OMNI_EPI_RESID>(nullptr, a.act, nullptr, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, c, 16u, 2u, 2u, 3u, 0x3003);
        if (has_qkv)
            xw_service<2, 2, 8, 2, OMNI_EPI_BF16>(a.ln1_next, a.resid, a.part, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg & 127, wg >> 7, c, 40u, 3u, 3u, 4u,
                                                  0x3004);
        if (xw_ld(&c.s->dead) && threadIdx.x == XW_CW * 64) atomicCAS(a.err, 0, 0x30ff);
    }
}

OMNI_KNOB g_bb_xw = 0;
#ifdef OMNI_DEBUG_HOOKS
extern "C" void omni_debug_bb_xw(int on) { g_bb_xw = on != 0; }
#endif

bool k_bb_xw_enabled() { return g_bb_xw != 0; }

int k_bb_xw(const omni_layer_weights& w, const omni_layer_weights* next, const void* attn, void* resid, float* part, void* act, void* qkv,
            int B, float eps, uint32_t* flags, int32_t* err, void* stream) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)bb_xw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, XW_LDS_BYTES);
        attr = true;
    }
    XwArgs a{};
    a.wo = (const uint16_t*)w.wo; a.ln2 = (const uint16_t*)w.ln2; a.wgu = (const uint16_t*)w.wgu; a.wdown = (const uint16_t*)w.wdown;
    a.ln1_next = next ? (const uint16_t*)next->ln1 : nullptr;
    a.wqkv_next = next ? (const uint16_t*)next->wqkv : nullptr;
    a.attn = (const uint16_t*)attn; a.resid = (uint16_t*)resid; a.part = part; a.act = (uint16_t*)act; a.qkv = (uint16_t*)qkv;
    a.B = B; a.nap = 1; a.eps = eps; a.flags = flags; a.err = err;
    hipLaunchKernelGGL(bb_xw_kernel, dim3(OMNI_CHAIN_WGS), dim3(XW_THREADS), XW_LDS_BYTES, (hipStream_t)stream, a);
    OMNI_CHECK_LAUNCH("bb_xw");
    return OMNI_OK;
}
