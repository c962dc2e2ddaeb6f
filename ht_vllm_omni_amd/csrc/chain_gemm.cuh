// One skinny-GEMM stage of a persistent chain (cp_chain.hip: the code predictor; bb_chain.hip: the talker backbone's
// o_proj -> gate_up -> down_proj -> next qkv segment): weights into registers BEFORE the stage flags, activations behind
// them with sc1 loads, gemm_skinny_kernel's arithmetic (k-step ownership of the 8 waves, accumulation and LDS combine
// order, rounding points) -- see cp_chain.hip for the protocol.
#pragma once
#include "coherent.cuh"
#include "common.cuh"
#include "gemm_frag.cuh"

#define CH_WAVES 8
#define CH_THREADS (CH_WAVES * 64)

// ---- in-kernel timeline of a stage (libomni_talker_debug.so only; scripts/chain_timeline.py): wave 0 of every workgroup
// stamps the constant 100 MHz counter at fixed points.  Product builds compile the macro away.
#define CH_NSTAMP 8
#define CH_STAMP_PASS (40 * CH_NSTAMP * OMNI_CHAIN_WGS)      // stamp words of one predictor pass (cp_chain.hip: block `pass` of the buffer)
#ifdef OMNI_DEBUG_HOOKS
#define CH_STAMP(buf, sidx, k)                                                                                          \
    do {                                                                                                                \
        if ((buf) != nullptr && threadIdx.x == 0)                                                                       \
            (buf)[((size_t)(sidx) * CH_NSTAMP + (k)) * OMNI_CHAIN_WGS + blockIdx.x] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define CH_STAMP(buf, sidx, k) do { (void)(buf); (void)(sidx); } while (0)
#endif

// PRO 3 (round 4: an A/B arm; round 6: the product form of every norm-fused stage whose normalised rows are nobody's output -- qkv, gate_up --
// in the chains and, with the same arithmetic, in gemm_skinny_kernel): the RMSNorm's rstd OFF the critical path.  PRO 2 needs rstd before the first MFMA
// (x = w * bf16(r * rstd)): slab loads -> LDS reduction -> barrier sit between the flags and the arithmetic.  PRO 3 feeds the MFMAs
// bf16(w * r) -- no rstd -- and scales the fp32 sums by rstd[row] in the epilogue: the slab reduction shares the combine barrier.
// One rounding point moves (bf16(w * bf16(r * rstd)) -> rstd * sum(W * bf16(w * r))): NOT the reference's bits; gated on the accuracy
// tests, never on bit-identity.
// ---- one skinny GEMM stage.  K = NTW * 256 (every wave owns NTW k-steps: wave, wave + 8, ...), workgroup tile = NT 16-row
// n-tiles x MT 16-row m-tiles at (bx, by).  PRO 2 = RMSNorm folded into the x fragments (slabs part_in), EPI as gemm.hip.
// XG = k-steps of x (and norm weights) in flight per wave behind the flags: 0 = the wave's whole share at once (small K), else
// a ring refilled as the MFMAs retire steps (fully unrolled: every load unconditional, hipcc counts vmcnt).
// WSRC: where the weight fragments come from -- 0: fetched whole into registers ahead of the flags; 1: they ride in the x ring
// (wide slices under a tight register budget); 2: the workgroup's LDS FIFO that loader waves fill by LDS-DMA (bb_engine.hip):
// round d = the k-steps 8 d .. 8 d + 7 of every n-tile = 8 NT pieces in the order [tile][k-step], wave w reads piece (j, w).
//        3: registers the PREVIOUS stage filled (Wpre): a stage calls `prefetch()` -- PF::LOADS loads of the next stage's slice
//        per wave -- behind its epilogue stores and then drains only down to vmcnt(PF::LOADS): the stores are out, the weight
//        loads stay in flight through the barrier, the flag and the next hand-off.  (Issued earlier -- behind the last
//        activation load -- they sit in front of the stores in the wave's in-order queue and the flag waits for the whole
//        slice: measured +0.2 ms per step.)
struct ChainNoPrefetch { static constexpr int LOADS = 0; __device__ __forceinline__ void operator()() const {} };
// ---- round 6: the tensor-parallel all-reduce INSIDE a residual stage (bb_chain.hip on a tensor-parallel rank; allreduce.hip is the same
// exchange as a launch of its own).  The stage's workgroup owns one 16-column x (16 MT)-row tile of the row-parallel GEMM's output on EVERY
// rank, so the exchange is per tile and needs no stage of its own:
//   partial   the tile's K-combined sums, rounded to bf16, into this rank's peer-mapped buffer data[rank] (system-scope write-through stores)
//   arrive    stores drained, workgroup barrier, system-scope release; lane p stores  tflags[p][rank][tile] = epoch  (peer p's flag array)
//   wait      lane p polls tflags[rank][p][tile] >= epoch (system-scope loads, bounded by wall clock: a peer that never arrives sets every
//             rank's error word and the launch carries on -- a wrong step, never a hung GPU); system-scope acquire; barrier
//   reduce    sum over the ranks in RANK ORDER (fp32, one rounding to bf16: the same bits on every rank), then the residual epilogue as ever
// Tile flags carry allreduce.hip's epoch sequence (o_proj and down_proj calls alternate between two data buffers): a workgroup passes the
// wait of call e + 1 only after every peer's workgroup of the same tile has finished reading call e, so call e + 2 may overwrite its buffer.
#define CH_AR_MAX_WORLD 8
#define CH_AR_TILES 256
#define CH_AR_WAIT_TICKS 500000000ull        // 5 s of the 100 MHz counter (allreduce.hip's bound)
#define CH_AUX_SYS 17                        // sc0 | sc1: system-scope coherent access
struct ChainAr {
    int world, rank;
    const uint16_t* data[CH_AR_MAX_WORLD];       // the ranks' partial buffers of this stage, fragment-major bf16 [64][N]
    uint32_t* tflags[CH_AR_MAX_WORLD];           // rank p's tile flags [world][CH_AR_TILES]
    int32_t* error[CH_AR_MAX_WORLD];             // every rank's error word (error[rank] = this rank's)
};
template <int N, class F>
struct ChainPrefetch {           // N = loads per wave that F issues
    F& f;
    static constexpr int LOADS = N;
    __device__ __forceinline__ void operator()() const { f(); }
};
// ONEPASS: more than 6 tiles still combine in ONE pass (the caller's LDS holds CH_WAVES * NT * MT KB of slots): one barrier
// instead of three and twice the epilogue threads at work; same order of additions.
// PS: rows per sum(r^2) slab (64; 128 for the code predictor's two-position pass, whose stream holds 128 rows)
// WAUX: cache policy of the WEIGHT loads (0 default, OMNI_AUX_NT = non-temporal: MI355X_MICROARCH "nt-weights" -- for slices that exactly
// one workgroup reads once per launch; round 5 A/B, profiles/r05_ab_table.txt)
template <int MT, int NT, int NTW, int PRO, int EPI, int XG = 0, int WSRC = 0, class PF = ChainNoPrefetch, bool ONEPASS = false, int PS = 64, int WAUX = 0,
          bool AR = false>
__device__ __forceinline__ void chain_gemm(const uint16_t* __restrict__ W, const uint16_t* __restrict__ norm_w, const uint16_t* x,
                                           const float* part_in, int np_in, void* out, int ldo, float* part_out, int M, int N, float eps,
                                           int bx, int by, float* lds, ChainGate& g, bool wait, int code,
                                           unsigned long long* stamps, const u32x4 (*Wpre)[NT] = nullptr, PF prefetch = PF(),
                                           uint16_t* normed_out = nullptr /* PRO 2: the normalised rows, row-major [M][K] (plain stores:
                                           for the NEXT launch), written by the workgroup columns bx < 8 as gemm_skinny_kernel spreads them */,
                                           const uint8_t* mask = nullptr /* EPI_F32_BF16RND: logit n is `mask_fill` where mask[n] == 0 (the codec mask) */,
                                           float mask_fill = 0.f, const int32_t* nlive_ptr = nullptr /* normed_out: rows >= *nlive_ptr are not written */,
                                           const ChainAr* ar = nullptr, uint32_t ar_epoch = 0 /* AR: the exchange's peers and its call number */) {
    const int sidx = ((code & 255) >> 4) * 5 + (code & 15) - 1;
    constexpr int G = (XG == 0 || XG > NTW) ? NTW : XG;
    constexpr bool NORM = PRO == 2 || PRO == 3, DEFER = PRO == 3;
    constexpr bool NW_EARLY = NORM && G == NTW && NT * NTW < 24;      // norm weights ahead of the flags (registers permitting)
    constexpr int K = NTW * CH_WAVES * 32;
    constexpr int nsteps = K / 32;
    constexpr bool GU8 = EPI == OMNI_EPI_SILU_MUL_GU8;
    // OMNI_EPI_SILU_MUL (round 5: the MoE layer's shared expert): W = [gate rows | up rows], n-tiles [0, NT/2) are gate tiles, [NT/2, NT)
    // the matching up tiles N rows further down (gemm_skinny_kernel's pairing); out = N activation columns
    constexpr bool SILU2 = EPI == OMNI_EPI_SILU_MUL;
    constexpr bool WRING = WSRC == 1, WFIFO = WSRC == 2;
    static_assert(EPI == OMNI_EPI_BF16 || EPI == OMNI_EPI_RESID || EPI == OMNI_EPI_F32_BF16RND || GU8 || SILU2, "chain_gemm: epilogue");
    static_assert(!SILU2 || (NT % 2 == 0 && WSRC == 0 && NT * MT <= 6), "chain_gemm: plain SiLU-mul = gate / up tile pairs, whole slice ahead of the flags");
    // first 16-row tile of W behind n-tile j of this workgroup
    auto wtile = [&](int j) { return SILU2 ? (j < NT / 2 ? bx * (NT / 2) + j : (N >> 4) + bx * (NT / 2) + (j - NT / 2)) : bx * NT + j; };
    static_assert(EPI != OMNI_EPI_RESID || (NT == 1 && PRO == 0), "chain_gemm: residual epilogue = one n-tile, plain x");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave-uniform: address parts below stay in SGPRs
    const int q = lane >> 4;
    const int m_base = by * (MT * 16);
    const int Mloc = min(M - m_base, MT * 16);
    if (Mloc <= 0) {                      // no rows here (batch smaller than the grid's row range): publish the stage, run ahead
        chain_gate_skip(g);
        return;
    }
    // every operand load is a buffer load: lane part (lane * 16 bytes) in ONE VGPR, tile / k-step part in an SGPR offset -- no
    // 64-bit per-load address pairs (they cost the wide gate_up tile its last registers)
    const coh_rsrc_t xrs = coh_rsrc(x), ors = coh_rsrc(out), wrs = coh_rsrc(W), nrs = coh_rsrc(NORM ? norm_w : W);
    const uint32_t lane16 = lane * 16;
    CH_STAMP(stamps, sidx, 0);                                           // 0: stage entered

    // ---- before the flags: everything that does not depend on the previous stage.  (The polling wave's first poll returns
    // behind its own weight loads -- a wave's loads return in order -- but letting wave 0 fetch its share of the slice behind
    // the flags instead measured WORSE: the predictor 1.80 -> 2.11 ms; its weights then arrive later than the activations.)
    constexpr int WS = (WFIFO || WSRC == 3) ? 1 : (WRING ? G : NTW);      // weight slots held at a time
    u32x4 Wq[WS][NT], NWq[G];
#pragma unroll
    for (int d = 0; d < NTW; ++d) {
        const int ks = wave + d * CH_WAVES;
        if (WSRC == 0) {
#ifdef OMNI_DEBUG_HOOKS
            if ((g.skip == 1 && d >= (NTW + 1) / 2) ||     // ingest experiment: the second half of the slice is not fetched
                (g.skip == 4 && wave == 0) || g.skip == 5) {   // round 6: the POLLING wave fetches no weights (4) / no wave does (5): what the polls lose behind them
#pragma unroll
                for (int j = 0; j < NT; ++j) Wq[d][j] = (u32x4){0u, 0u, 0u, 0u};
            } else
#endif
#pragma unroll
            for (int j = 0; j < NT; ++j)
                Wq[d][j] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (uint32_t)((wtile(j) * nsteps + ks) * 1024), WAUX);
        }
        if (NW_EARLY) NWq[d % G] = __builtin_amdgcn_raw_buffer_load_b128(nrs, q * 16, ks * 64, 0);
    }
    u32x2 r_old = (u32x2){0u, 0u};
    if (EPI == OMNI_EPI_RESID && threadIdx.x < MT * 64) {
        // the old residual values of this thread's epilogue item: last written two or more stages ago (o_proj: by the
        // previous layer's down_proj; down_proj: by this layer's o_proj), final since the gate of the stage in between
        const int ml = (threadIdx.x >> 6) * 16 + (lane & 15);
        if (ml < Mloc) r_old = coh_ld8(ors, (uint32_t)frag_off(m_base + ml, bx * 16 + 4 * (lane >> 4), N) * 2);
    }

    CH_STAMP(stamps, sidx, 1);                                           // 1: weight loads issued
    if (wait) chain_gate_wait(g, code);
    CH_STAMP(stamps, sidx, 2);                                           // 2: flags seen, barrier passed

    // ---- behind the flags: the slabs (first: they return first) and the activation fragments, all in one round trip
    constexpr int XROWS = MT * 16;
    typedef SlabOrder<XROWS> SO;             // the canonical slab order (gemm_frag.cuh): the launch path's, whatever row tile it picked
    float pv[SO::PE];
    if (NORM) {
        const coh_rsrc_t prs = coh_rsrc(part_in);
        const int row = lane % XROWS, sub = lane / XROWS;
#pragma unroll
        for (int e = 0; e < SO::PE; ++e) {
            const int p = SO::index(wave, sub, e);
            pv[e] = coh_ldf(prs, (uint32_t)(min(p, np_in - 1) * PS + m_base + row) * 4);
            if (p >= np_in) pv[e] = 0.f;
        }
    }
    u32x4 Xq[G][MT];
    auto load_x = [&](int d) {           // k-step d of this wave's share into ring slot d % G
        if (WRING) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
                Wq[d % G][j] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (uint32_t)(((bx * NT + j) * nsteps + wave + d * CH_WAVES) * 1024), WAUX);
        }
        if (NORM && !NW_EARLY) NWq[d % G] = __builtin_amdgcn_raw_buffer_load_b128(nrs, q * 16, (wave + d * CH_WAVES) * 64, 0);
#ifdef OMNI_DEBUG_HOOKS
        if (g.skip == 2 && d >= (NTW + 1) / 2) {           // ingest experiment: half of the activation fragments are not fetched
#pragma unroll
            for (int i = 0; i < MT; ++i) Xq[d % G][i] = (u32x4){0u, 0u, 0u, 0u};
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < MT; ++i)
            Xq[d % G][i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane16, (uint32_t)((((m_base >> 4) + i) * nsteps + wave + d * CH_WAVES) * 1024),
                                                                 OMNI_AUX_SC1);
    };
#pragma unroll
    for (int d = 0; d < G; ++d) load_x(d);

    float rstd[MT];
    float* red = lds + CH_WAVES * ((NT * MT > 6 && !ONEPASS) ? NT * MT / 2 : NT * MT) * 4 * 64;      // behind the combine slots
    if (PRO == 2) {
        // fixed-order reduction of the slabs -> rstd of this workgroup's rows (gemm.hip xnorm_rstd, same order of additions)
        red[wave * 64 + lane] = SO::reduce(pv);
        chain_barrier(g);
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < CH_WAVES; ++w) t += red[w * 64 + lane];
        const float rl = 1.0f / sqrtf(t / (float)K + eps);
#pragma unroll
        for (int i = 0; i < MT; ++i) rstd[i] = __shfl(rl, i * 16 + (lane & 15), 64);
    }
    // (PRO 3: the slab loads went out ahead of the activation loads and returned ahead of them; their reduction waits behind the MFMA loop,
    //  where nothing is stalled on it -- written there, not left to the scheduler: the register pins of SlabOrder::reduce fix its place)

    CH_STAMP(stamps, sidx, 3);                                           // 3: rstd known (slabs arrived, reduced)
    f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < NTW; ++d) {
        u32x4 Xn[MT], Wn[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
            Xn[i] = PRO == 2 ? xnorm_frag(Xq[d % G][i], NWq[d % G], rstd[i]) : (DEFER ? xw_frag(Xq[d % G][i], NWq[d % G]) : Xq[d % G][i]);
        if (WFIFO) {
            const unsigned first = g.piece_base + (unsigned)d * (8 * NT);
            eng_wait_ready(g, first + (NT - 1) * 8 + wave + 1);               // this wave's last piece of the round (pieces land in order)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                Wn[j] = *reinterpret_cast<const u32x4*>(g.fifo + (size_t)((first + j * 8 + wave) % ENG_FIFO_PIECES) * 1024 + lane16);
        } else if (WSRC == 3) {
#pragma unroll
            for (int j = 0; j < NT; ++j) Wn[j] = Wpre[d][j];
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j) Wn[j] = Wq[WRING ? d % G : d][j];
        }
        if (PRO == 2 && normed_out != nullptr && bx < CH_WAVES && wave == bx) {
            // k-steps bx, bx + 8, ... of the normalised rows leave through workgroup column bx (gemm_skinny_kernel: column x takes k-steps
            // x + gridDim.x * wave): here wave w owns k-steps w + 8 d, so column bx's wave bx holds exactly those -- the same values
            const int ks = wave + d * CH_WAVES;
            const int nlive = nlive_ptr ? min(*nlive_ptr, M) - m_base : Mloc;      // h[t+1] of a padded bucket's rows stays untouched
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (i * 16 + (lane & 15) < min(Mloc, nlive))
                    *reinterpret_cast<u32x4*>(normed_out + (size_t)(m_base + i * 16 + (lane & 15)) * K + ks * 32 + 8 * q) = Xn[i];
        }
        if (d + G < NTW) load_x(d + G);                                   // refill the slot just consumed
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[j][i] = mfma16(Wn[j], Xn[i], acc[j][i]);
        if (WFIFO) eng_release(g, wave, g.piece_base + (unsigned)(d + 1) * (8 * NT));      // the MFMAs above hold the fragments: the pieces are free
    }

    if (DEFER) red[wave * 64 + lane] = SO::reduce(pv);      // the row statistics: read behind the combine barrier, by the epilogue
    // ---- combine the 8 K-partials through LDS (wave order 0..7), epilogue with write-through stores.  More than 6 tiles (the
    // backbone's gate_up: 12) go through the combine area in two passes of NT * MT / 2 tiles, so that it stays at 48 KB
    constexpr int PASSES = (NT * MT > 6 && !ONEPASS) ? 2 : 1, TP = NT * MT / PASSES;
    static_assert(NT * MT % PASSES == 0, "chain_gemm: tiles per combine pass");
    f32x4* lds4 = reinterpret_cast<f32x4*>(lds);
    constexpr int LN = GU8 ? 32 : 64;
    constexpr int ITEMS = (SILU2 ? TP / 2 : TP) * LN;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
    if (pass > 0) chain_barrier(g);                                      // the previous pass's readers are done with the slots
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
            if ((j * MT + i) / TP == pass) lds4[(wave * TP + (j * MT + i) % TP) * 64 + lane] = acc[j][i];
    if (pass == 0) CH_STAMP(stamps, sidx, 4);                            // 4: operands arrived, MFMAs done, partials in LDS
    chain_barrier(g);
    if (pass == 0) CH_STAMP(stamps, sidx, 5);                            // 5: combine barrier passed
    if constexpr (AR) {
        static_assert(EPI == OMNI_EPI_RESID && PASSES == 1 && ITEMS <= CH_THREADS, "chain_gemm: the all-reduce rides in the residual epilogue, one item per thread");
        const int l = threadIdx.x & 63, tl = threadIdx.x >> 6;
        const int ml = (tl % MT) * 16 + (l & 15);
        const bool has = (int)threadIdx.x < ITEMS && ml < Mloc;
        const int m = m_base + ml, n = bx * 16 + 4 * (l >> 4);
        const uint32_t off = (uint32_t)frag_off(has ? m : m_base, n, N) * 2;
        f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has) {
#pragma unroll
            for (int w = 0; w < CH_WAVES; ++w) sum += lds4[(w * TP + tl) * 64 + l];
        }
        const u32x2 mine = (u32x2){pack_bf2(sum[0], sum[1]), pack_bf2(sum[2], sum[3])};      // this rank's partial, bf16
        float acc[4] = {bf_lo(mine[0]), bf_hi(mine[0]), bf_lo(mine[1]), bf_hi(mine[1])};
        const int world = ar->world, rank = ar->rank;
        if (world > 1) {
            const int tile = by * (N >> 4) + bx;
            if (has) __builtin_amdgcn_raw_buffer_store_b64(mine, coh_rsrc(ar->data[rank]), off, 0, CH_AUX_SYS);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            chain_barrier(g);
            if (threadIdx.x < 64) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                  // system scope: the tile is out before its flag
                const int p = threadIdx.x;
                if (p < world && p != rank) {
                    __hip_atomic_store(ar->tflags[p] + rank * CH_AR_TILES + tile, ar_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    // a peer that failed to arrive once is not waited for again (allreduce.hip): one time-out per dead rank, not one per stage
                    if (!g.dead && __hip_atomic_load(ar->error[rank], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                        const uint32_t* f = ar->tflags[rank] + p * CH_AR_TILES + tile;
                        uint32_t spins = 0;
                        unsigned long long t0 = 0;
                        while ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - ar_epoch) < 0) {
                            __builtin_amdgcn_s_sleep(1);
                            if ((++spins & 1023u) == 0) {
                                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                                if (spins == 1024u) t0 = now;
                                else if (now - t0 > CH_AR_WAIT_TICKS) {
                                    const int ecode = 1 + p + 256 * (rank + 1) + 0x10000;      // 0x10000: raised by a chain stage
                                    for (int q = 0; q < world; ++q) __hip_atomic_store(ar->error[q], ecode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    break;
                                }
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                      // system scope: the peers' tiles below are fresh
            chain_barrier(g);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = 0.f;
            if (has) {
                for (int p = 0; p < world; ++p) {                              // rank order: identical bits on every rank
                    const u32x2 v = p == rank ? mine : __builtin_amdgcn_raw_buffer_load_b64(coh_rsrc(ar->data[p]), off, 0, CH_AUX_SYS);
                    acc[0] += bf_lo(v[0]); acc[1] += bf_hi(v[0]); acc[2] += bf_lo(v[1]); acc[3] += bf_hi(v[1]);
                }
            }
        }
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (has) {
#pragma unroll
            for (int e = 0; e < 4; ++e) rv[e] = bfround(acc[e]);               // the all-reduced delta, bf16 (one rank: the partial itself)
            rv[0] = bfround(bf_lo(r_old[0]) + rv[0]); rv[1] = bfround(bf_hi(r_old[0]) + rv[1]);
            rv[2] = bfround(bf_lo(r_old[1]) + rv[2]); rv[3] = bfround(bf_hi(r_old[1]) + rv[3]);
            coh_st8(ors, off, (u32x2){pack_bf2(rv[0], rv[1]), pack_bf2(rv[2], rv[3])});
        }
        float ss = sq4_sum(rv[0], rv[1], rv[2], rv[3]);
        ss = xor32_sum(xor16_sum(ss));                                         // (whole waves: rows past Mloc add into lanes nobody stores)
        if (has && l < 16) coh_st4(coh_rsrc(part_out), (uint32_t)(bx * PS + m) * 4, __float_as_uint(ss));
    } else
    for (int it = threadIdx.x; it < ITEMS; it += CH_THREADS) {
        const int l = it % LN;
        const int tl = it / LN, t = pass * TP + tl;                       // tile index j * MT + i
        const int i = t % MT, j = t / MT;
        const int ml = i * 16 + (l & 15);
        if (ml >= Mloc) continue;
        const int m = m_base + ml;
        f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f}, sum2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < CH_WAVES; ++w) {
            sum += lds4[(w * TP + tl) * 64 + l];
            if (GU8) sum2 += lds4[(w * TP + tl) * 64 + l + 32];
            if (SILU2) sum2 += lds4[(w * TP + (NT / 2 + j) * MT + i) * 64 + l];
        }
        if (DEFER) {
            // rstd of this item's row: the eight wave partials of the slab reduction, in wave order (the combine barrier covered them)
            float tsum = 0.f;
#pragma unroll
            for (int w = 0; w < CH_WAVES; ++w) tsum += red[w * 64 + ml];
            const float rl = 1.0f / sqrtf(tsum / (float)K + eps);
            sum *= rl;
            if (GU8 || SILU2) sum2 *= rl;
        }
#ifdef OMNI_DEBUG_HOOKS
        if (g.skip == 3 && sum[0] != 12345.678f) continue;       // timing experiment: no epilogue stores (results garbage)
#endif
        if (SILU2) {
            const int n = bx * 16 * (NT / 2) + j * 16 + 4 * (l >> 4);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = silu_mul_bf16(sum[e], sum2[e]);
            coh_st8(ors, (uint32_t)frag_off(m, n, N) * 2, (u32x2){pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])});
        } else if (GU8) {
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
            float ss = sq4_sum(rv[0], rv[1], rv[2], rv[3]);
            ss = xor32_sum(xor16_sum(ss));
            if (l < 16) coh_st4(coh_rsrc(part_out), (uint32_t)(bx * PS + m) * 4, __float_as_uint(ss));
        } else if (EPI == OMNI_EPI_F32_BF16RND) {
            // logits: fp32 cells holding bf16-rounded values (the reference's bf16 head output), row-major
            const int n = (bx * NT + j) * 16 + 4 * (l >> 4);
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = bfround(sum[e]);
                if (mask != nullptr && !mask[n + e]) y[e] = mask_fill;
            }
            coh_st16(ors, (uint32_t)((size_t)m * ldo + n) * 4,
                     (u32x4){__float_as_uint(y[0]), __float_as_uint(y[1]), __float_as_uint(y[2]), __float_as_uint(y[3])});
        } else {
            const int n = (bx * NT + j) * 16 + 4 * (l >> 4);
            coh_st8(ors, (uint32_t)((size_t)m * ldo + n) * 2, (u32x2){pack_bf2(sum[0], sum[1]), pack_bf2(sum[2], sum[3])});
        }
    }
    }   // combine pass
    CH_STAMP(stamps, sidx, 6);                                           // 6: epilogue stores issued
    prefetch();
    chain_gate_arrive<PF::LOADS>(g);
    CH_STAMP(stamps, sidx, 7);                                           // 7: stores drained, barrier, flag published
}

