// The backbone segment o_proj -> gate_up -> down_proj -> next qkv (bb_chain.hip) with its stages dealt to TWO wave groups of one
// workgroup that alternate ("ping-pong"): waves 0-3 run o_proj and down_proj, waves 4-7 run gate_up and the next qkv.
//
// Why: a wave's vector-memory operations return in order.  In the plain chain the 8 waves of a workgroup issue a stage's weight
// loads at the stage's entry and then the flag poll and the activation loads BEHIND them, so a stage is
//     (its weight slice lands) -> flags -> activations -> MFMA -> stores -> flag,
// serial, and the HBM stream stops for every hand-off (~3 us x 4 stages of a 39 us layer).  Issuing the NEXT stage's slice from
// the same waves only moves the queue: the next hand-off's poll and activation loads then wait behind that slice
// (bb_chain.hip `pf`: measured slower).  Here the group that is NOT computing has nothing in its queue but its next stage's
// weight ring: those loads land while the other group computes, and when the other group's flags come up its poll and its
// activation loads return at once.  One group's stream covers the other group's hand-off.
//
// Inside a group: 4 waves, k-step ownership gw, gw + 4, ...; a ring of G k-steps (weights + norm weights + activations) per
// wave, the weight part of the first G steps fetched ahead of the flags; K-partials combined through the group's own LDS area in
// wave order; group barriers are arrivals on an LDS counter (s_barrier would stop the other group).  Cross-workgroup protocol =
// the chains' (coherent.cuh): sc1 write-through stores, one flag word per workgroup = stages completed, bounded polls, sticky
// error word.  Sums are taken in a different order than the launch path's 8-wave kernels (4 K-partials instead of 8; the slab
// reduction by 4 waves): results agree with the launch path to accumulation-order rounding, not bit for bit -- parity is against
// the oracle (tests/test_gpu_chain.py).  OFF by default: see the note at g_bb_pp.
#include "chain_gemm.cuh"
#include "common.cuh"
#include "kernels.h"

#define PP_GW 4                        // waves per group
#define PP_GT (PP_GW * 64)             // threads per group
#define PP_SPIN_BOUND (1u << 24)

struct PpSync {
    unsigned bar[2];                   // per group: barrier arrivals
    unsigned dead;                     // a bounded wait ran out
    unsigned pad;
};

struct PpGate {
    coh_rsrc_t frs;                    // flag words [OMNI_CHAIN_WGS]
    uint32_t base;                     // this workgroup's flag at entry (= every workgroup's)
    int32_t* err;
    bool dead;
    int nap, gid;
    PpSync* ps;
    unsigned bgen;
};

__device__ __forceinline__ void pp_barrier(PpGate& g) {
    // LDS operations of a wave execute in order; the words are relaxed atomics: only the COMPILER must keep the order
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    g.bgen += 1;
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&g.ps->bar[g.gid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned target = g.bgen * PP_GW;
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(&g.ps->bar[g.gid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > PP_SPIN_BOUND) {
            __hip_atomic_store(&g.ps->dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
    }
    asm volatile("" ::: "memory");
}

// every workgroup has completed `stages` stages of this launch: wave 0 of the group polls all 256 flags (64 lanes x 4 words)
__device__ __forceinline__ void pp_gate_wait(PpGate& g, uint32_t stages, int code) {
    if ((threadIdx.x & (PP_GT - 1)) < 64 && !g.dead) {
        const uint32_t want = g.base + stages;
        const uint32_t off = (threadIdx.x & 63) * 16;
        unsigned spins = 0;
        for (;;) {
            asm volatile("" ::: "memory");
            const u32x4 f = coh_ld16(g.frs, off);
            const bool behind = (int)(f[0] - want) < 0 || (int)(f[1] - want) < 0 || (int)(f[2] - want) < 0 || (int)(f[3] - want) < 0;
            if (__builtin_amdgcn_ballot_w64(behind) == 0) break;
            if (g.nap <= 1) __builtin_amdgcn_s_sleep(1);
            else if (g.nap == 2) __builtin_amdgcn_s_sleep(2);
            else __builtin_amdgcn_s_sleep(4);
            if (++spins > OMNI_CHAIN_SPIN_BOUND) {
                if ((threadIdx.x & 63) == 0) atomicCAS(g.err, 0, code);
                g.dead = true;
                break;
            }
        }
    }
    pp_barrier(g);
}

// the group's stores of the stage are out (every wave drains its own queue), then one lane publishes the workgroup's count
__device__ __forceinline__ void pp_gate_arrive(PpGate& g, uint32_t stages) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pp_barrier(g);
    if ((threadIdx.x & (PP_GT - 1)) < 64) chain_flag_publish(g.frs, blockIdx.x, g.base + stages);
}

// One skinny-GEMM stage on a 4-wave group: K = KS * 128 (wave gw owns k-steps gw, gw + 4, ...), tile = NT n-tiles x MT m-tiles
// at (bx, by); PRO 2 = RMSNorm folded into the x fragments; EPI as gemm.hip.  G = ring depth in k-steps per wave.
template <int MT, int NT, int KS, int PRO, int EPI, int G>
__device__ __forceinline__ void pp_gemm(const uint16_t* __restrict__ W, const uint16_t* __restrict__ norm_w, const uint16_t* x,
                                        const float* part_in, int np_in, void* out, int ldo, float* part_out, int M, int N, float eps,
                                        int bx, int by, float* lds, PpGate& g, uint32_t wait_stages, uint32_t done_stages, int code,
                                        unsigned long long* stamps) {
    const int sidx = (int)done_stages - 1;
#ifdef OMNI_DEBUG_HOOKS
#define PP_STAMP(k)                                                                                                              \
    do {                                                                                                                         \
        if (stamps != nullptr && (threadIdx.x & (PP_GT - 1)) == 0)                                                               \
            stamps[((size_t)sidx * CH_NSTAMP + (k)) * OMNI_CHAIN_WGS + blockIdx.x] = __builtin_amdgcn_s_memrealtime();           \
    } while (0)
#else
#define PP_STAMP(k) do { (void)stamps; (void)sidx; } while (0)
#endif
    PP_STAMP(0);
    constexpr int nsteps = KS * PP_GW, K = nsteps * 32;
    constexpr bool GU8 = EPI == OMNI_EPI_SILU_MUL_GU8;
    static_assert(G <= KS, "pp_gemm: ring deeper than the wave's share");
    static_assert(EPI == OMNI_EPI_BF16 || EPI == OMNI_EPI_RESID || GU8, "pp_gemm: epilogue");
    static_assert(EPI != OMNI_EPI_RESID || (NT == 1 && PRO == 0), "pp_gemm: residual epilogue = one n-tile, plain x");
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & (PP_GW - 1));
    const int tg = threadIdx.x & (PP_GT - 1);
    const int q = lane >> 4;
    const int m_base = by * (MT * 16);
    const int Mloc = min(M - m_base, MT * 16);
    const coh_rsrc_t xrs = coh_rsrc(x), ors = coh_rsrc(out), wrs = coh_rsrc(W), nrs = coh_rsrc(PRO == 2 ? norm_w : W);
    const uint32_t lane16 = lane * 16;

    // ---- ahead of the flags: the weight part of the ring's first G k-steps
    u32x4 Wq[G][NT], NWq[G], Xq[G][MT];
    auto load_w = [&](int d) {
        const int ks = gw + d * PP_GW;
#pragma unroll
        for (int j = 0; j < NT; ++j)
            Wq[d % G][j] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (uint32_t)(((bx * NT + j) * nsteps + ks) * 1024), 0);
        if (PRO == 2) NWq[d % G] = __builtin_amdgcn_raw_buffer_load_b128(nrs, q * 16, ks * 64, 0);
    };
    auto load_x = [&](int d) {
        const int ks = gw + d * PP_GW;
#pragma unroll
        for (int i = 0; i < MT; ++i)
            Xq[d % G][i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane16, (uint32_t)((((m_base >> 4) + i) * nsteps + ks) * 1024), OMNI_AUX_SC1);
    };
#pragma unroll
    for (int d = 0; d < G; ++d) load_w(d);
    u32x2 r_old = (u32x2){0u, 0u};
    if (EPI == OMNI_EPI_RESID && tg < MT * 64) {
        const int ml = (tg >> 6) * 16 + (lane & 15);
        if (ml < Mloc) r_old = coh_ld8(ors, (uint32_t)frag_off(m_base + ml, bx * 16 + 4 * (lane >> 4), N) * 2);
    }
    PP_STAMP(1);
    if (wait_stages) pp_gate_wait(g, wait_stages, code);
    PP_STAMP(2);

    // ---- behind the flags: slabs, then the activation part of the ring
    constexpr int XROWS = MT * 16, NCH = PP_GT / XROWS, PE = 128 / NCH;
    float pv[PE];
    if (PRO == 2) {
        const coh_rsrc_t prs = coh_rsrc(part_in);
        const int row = tg % XROWS, ch = tg / XROWS;
#pragma unroll
        for (int e = 0; e < PE; ++e) {
            const int p = ch + e * NCH;
            pv[e] = coh_ldf(prs, (uint32_t)(min(p, np_in - 1) * 64 + m_base + row) * 4);
            if (p >= np_in) pv[e] = 0.f;
        }
    }
#pragma unroll
    for (int d = 0; d < G; ++d) load_x(d);

    float rstd[MT];
    if (PRO == 2) {
        float s_ = 0.f;
#pragma unroll
        for (int e = 0; e < PE; ++e) s_ += pv[e];
        if (XROWS <= 32) s_ = xor32_sum(s_);
        if (XROWS <= 16) s_ = xor16_sum(s_);
        float* red = lds + PP_GW * (NT * MT > 6 ? NT * MT / 2 : NT * MT) * 4 * 64;       // behind the combine slots
        red[gw * 64 + lane] = s_;
        pp_barrier(g);
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < PP_GW; ++w) t += red[w * 64 + lane];
        const float rl = 1.0f / sqrtf(t / (float)K + eps);
#pragma unroll
        for (int i = 0; i < MT; ++i) rstd[i] = __shfl(rl, i * 16 + (lane & 15), 64);
    }

    PP_STAMP(3);
    f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < KS; ++d) {
        u32x4 Xn[MT], Wn[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) Xn[i] = PRO == 2 ? xnorm_frag(Xq[d % G][i], NWq[d % G], rstd[i]) : Xq[d % G][i];
#pragma unroll
        for (int j = 0; j < NT; ++j) Wn[j] = Wq[d % G][j];
        if (d + G < KS) {
            load_w(d + G);
            load_x(d + G);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[j][i] = mfma16(Wn[j], Xn[i], acc[j][i]);
    }

    // ---- combine the 4 K-partials through the group's LDS area (wave order), epilogue with write-through stores
    constexpr int PASSES = NT * MT > 6 ? 2 : 1, TP = NT * MT / PASSES;
    static_assert(NT * MT % PASSES == 0, "pp_gemm: tiles per combine pass");
    f32x4* lds4 = reinterpret_cast<f32x4*>(lds);
    constexpr int LN = GU8 ? 32 : 64;
    constexpr int ITEMS = TP * LN;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        if (pass > 0) pp_barrier(g);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if ((j * MT + i) / TP == pass) lds4[(gw * TP + (j * MT + i) % TP) * 64 + lane] = acc[j][i];
        if (pass == 0) PP_STAMP(4);
        pp_barrier(g);
        if (pass == 0) PP_STAMP(5);
        for (int it = tg; it < ITEMS; it += PP_GT) {
            const int l = it % LN;
            const int tl = it / LN, t = pass * TP + tl;
            const int i = t % MT, j = t / MT;
            const int ml = i * 16 + (l & 15);
            if (ml >= Mloc) continue;
            const int m = m_base + ml;
            f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f}, sum2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < PP_GW; ++w) {
                sum += lds4[(w * TP + tl) * 64 + l];
                if (GU8) sum2 += lds4[(w * TP + tl) * 64 + l + 32];
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
    }
    PP_STAMP(6);
    pp_gate_arrive(g, done_stages);
    PP_STAMP(7);
#undef PP_STAMP
}

struct PpArgs {
    const uint16_t *wo, *ln2, *wgu, *wdown, *ln1_next, *wqkv_next;      // *_next == NULL: last layer, no qkv stage
    const uint16_t* attn; uint16_t* resid; float* part; uint16_t* act; uint16_t* qkv;
    int B, nap; float eps;
    uint32_t* flags; int32_t* err;
    unsigned long long* stamps;
};

#define PP_GROUP_LDS (PP_GW * 6 * 64 * 16 + PP_GW * 64 * 4)              // combine slots of one pass (<= 6 tiles) + the rstd area
#define PP_LDS_BYTES (64 + 2 * PP_GROUP_LDS)

__global__ __launch_bounds__(CH_THREADS) void bb_pp_kernel(const PpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    PpSync* ps = reinterpret_cast<PpSync*>(lds);
    if (threadIdx.x < 4) reinterpret_cast<unsigned*>(ps)[threadIdx.x] = 0u;
    __syncthreads();                                   // the only s_barrier: from here on the groups run out of step
    PpGate g;
    g.frs = coh_rsrc(a.flags);
    g.err = a.err;
    g.base = __builtin_amdgcn_readfirstlane(coh_ld4(g.frs, blockIdx.x * 4));
    g.dead = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0;
    g.nap = a.nap;
    g.gid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    g.ps = ps;
    g.bgen = 0;
    float* gl = lds + 16 + g.gid * (PP_GROUP_LDS / 4);
    const int wg = blockIdx.x;
    constexpr int H = 2048, I = 6144, NQ = 4096;
    const uint32_t nstages = a.wqkv_next ? 4u : 3u;
    if (g.gid == 0) {
        pp_gemm<2, 1, 16, 0, OMNI_EPI_RESID, 8>(a.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, gl, g,
                                                0u, 1u, 0x2001, a.stamps);
        pp_gemm<2, 1, 48, 0, OMNI_EPI_RESID, 12>(a.wdown, nullptr, a.act, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, gl, g,
                                                 2u, 3u, 0x2003, a.stamps);
    } else {
        pp_gemm<4, 3, 16, 2, OMNI_EPI_SILU_MUL_GU8, 4>(a.wgu, a.ln2, a.resid, a.part, H / 16, a.act, 0, nullptr, a.B, I, a.eps, wg, 0, gl, g,
                                                       1u, 2u, 0x2002, a.stamps);
        if (nstages == 4u)
            pp_gemm<2, 2, 16, 2, OMNI_EPI_BF16, 6>(a.wqkv_next, a.ln1_next, a.resid, a.part, H / 16, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg & 127,
                                                   wg >> 7, gl, g, 3u, 4u, 0x2004, a.stamps);
    }
    if (ps->dead && (threadIdx.x & (PP_GT - 1)) == 0) atomicCAS(a.err, 0, 0x20ff);
}

// Measured (round 3, W3 step, same process): backbone 1.93 ms per step against the plain chain's 1.49 -- OFF.  The in-kernel
// timeline shows why (scripts/bb_timeline.py, BB_STAMPS=pp): the idle group's 16-48 weight loads take 2-3 us just to ISSUE and
// the other group's flag poll / activation loads return 2.4-4 us late: the CU's vector-memory path is one queue for all its
// waves, so bulk weight loads in flight delay every latency-critical load of the workgroup whichever wave issues them; and
// with a ring of G k-steps per wave the in-order return ties the weight prefetch distance to the activation ring's depth
// (gate_up's main loop: 14.7 us for 192 KB = 3.3 TB/s chip-wide).  Kept with its knob as the A/B arm of DESIGN section 6.
OMNI_KNOB g_bb_pp = 0;
#ifdef OMNI_DEBUG_HOOKS
static unsigned long long* g_pp_stamps = nullptr;
extern "C" void omni_debug_bb_pp(int on) { g_bb_pp = on != 0; }
extern "C" void omni_debug_pp_stamps(void* buf) { g_pp_stamps = (unsigned long long*)buf; }
#endif

bool k_bb_pp_enabled() { return g_bb_pp != 0; }

int k_bb_pp(const omni_layer_weights& w, const omni_layer_weights* next, const void* attn, void* resid, float* part, void* act, void* qkv,
            int B, float eps, uint32_t* flags, int32_t* err, void* stream) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)bb_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
        attr = true;
    }
    PpArgs a{};
    a.wo = (const uint16_t*)w.wo; a.ln2 = (const uint16_t*)w.ln2; a.wgu = (const uint16_t*)w.wgu; a.wdown = (const uint16_t*)w.wdown;
    a.ln1_next = next ? (const uint16_t*)next->ln1 : nullptr;
    a.wqkv_next = next ? (const uint16_t*)next->wqkv : nullptr;
    a.attn = (const uint16_t*)attn; a.resid = (uint16_t*)resid; a.part = part; a.act = (uint16_t*)act; a.qkv = (uint16_t*)qkv;
    a.B = B; a.nap = 1; a.eps = eps; a.flags = flags; a.err = err;
#ifdef OMNI_DEBUG_HOOKS
    a.stamps = g_pp_stamps;
#endif
    hipLaunchKernelGGL(bb_pp_kernel, dim3(OMNI_CHAIN_WGS), dim3(CH_THREADS), PP_LDS_BYTES, (hipStream_t)stream, a);
    OMNI_CHECK_LAUNCH("bb_pp");
    return OMNI_OK;
}
