// Cross-workgroup hand-offs INSIDE one launch (cp_chain.hip).  gfx950 has one L2 per XCD and a vector L1 per CU that other
// CUs' stores never refresh, so bytes that another workgroup wrote during the same launch are moved "coherent by access"
// (MI355X_MICROARCH, Workgroup dispatch / inter-workgroup visibility, first row of the sc1 table):
//   producer  every store of the handed-off bytes carries sc1 (write-through, the line is dropped from the XCD's L2); every
//             storing wave runs `s_waitcnt vmcnt(0)`; a workgroup barrier; then ONE lane publishes the workgroup's flag (sc1 store)
//   consumer  one wave polls the flags with sc1 loads; the other waves pass a workgroup barrier behind it; every load of the
//             handed-off bytes is an sc1 buffer load to registers (bypasses the L1; 4-, 8- or 16-byte accesses only)
// No release / acquire fence anywhere: a fence writes back or invalidates whole caches (38-43 us per grid barrier measured in
// round 2), an sc1 access costs what a plain one costs at 16 bytes.
//
// Buffer addressing: the resource descriptor is built from a wave-uniform base (SGPRs), lanes supply 32-bit byte offsets
// (`buffer_load_dwordx4 v, v_off, s[rsrc], 0 offen sc1`); hipcc counts these loads in vmcnt like any other.
#pragma once
#include "common.cuh"
#include "gemm_frag.cuh"

#define OMNI_AUX_SC1 16        // cache-policy bit of the raw buffer builtins: sc1 on gfx940+
#define OMNI_AUX_NT 2          // ... and nt (non-temporal: streamed bytes that nobody re-reads)

// The flag words are kept in OMNI_FLAG_REPLICAS copies, OMNI_FLAG_STRIDE words apart (lines of different memory channels): a
// workgroup publishes to every copy with ONE store instruction (one lane per copy) and polls only the copy of its XCD
// (blockIdx.x % 8).  With one copy, the 128 pollers of a flag domain read the same four 128-byte lines through the fabric at
// once and the producers' flag stores queue behind them at one memory channel (MI355X_MICROARCH: a counter kept in R replicas).
#define OMNI_FLAG_REPLICAS 8
#define OMNI_FLAG_STRIDE 2048          // words between two copies (8 KB)
#define OMNI_FLAG_WORDS (OMNI_FLAG_REPLICAS * OMNI_FLAG_STRIDE)

typedef __amdgpu_buffer_rsrc_t coh_rsrc_t;

__device__ __forceinline__ coh_rsrc_t coh_rsrc(const void* base) {
    // stride 0 (raw), num_records = max, flags: DATA_FORMAT = 32 (bits 12-18 of dword 3: 0x00020000)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 coh_ld16(coh_rsrc_t rs, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, OMNI_AUX_SC1);
}
__device__ __forceinline__ u32x2 coh_ld8(coh_rsrc_t rs, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, OMNI_AUX_SC1);
}
__device__ __forceinline__ uint32_t coh_ld4(coh_rsrc_t rs, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, OMNI_AUX_SC1);
}
__device__ __forceinline__ float coh_ldf(coh_rsrc_t rs, uint32_t byte_off) { return __uint_as_float(coh_ld4(rs, byte_off)); }
__device__ __forceinline__ void coh_st16(coh_rsrc_t rs, uint32_t byte_off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, OMNI_AUX_SC1);
}
__device__ __forceinline__ void coh_st8(coh_rsrc_t rs, uint32_t byte_off, u32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, byte_off, 0, OMNI_AUX_SC1);
}
__device__ __forceinline__ void coh_st4(coh_rsrc_t rs, uint32_t byte_off, uint32_t v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, rs, byte_off, 0, OMNI_AUX_SC1);
}

// publish `value` as workgroup `wg`'s flag in every copy (called by the publishing wave: lanes 0 .. REPLICAS - 1 store)
__device__ __forceinline__ void chain_flag_publish(coh_rsrc_t frs, int wg, uint32_t value) {
    const int l = threadIdx.x & 63;
    if (l < OMNI_FLAG_REPLICAS) coh_st4(frs, (uint32_t)(l * OMNI_FLAG_STRIDE + wg) * 4, value);
}
// byte offset of the copy this workgroup polls
__device__ __forceinline__ uint32_t chain_flag_copy() { return (uint32_t)(blockIdx.x & (OMNI_FLAG_REPLICAS - 1)) * (OMNI_FLAG_STRIDE * 4); }

// ---- grid-wide stage flags: flag[w] = number of stages workgroup w has completed since the engine was created (wraps).
// A stage's dependent part starts once every flag has reached the epoch of the stage before it.  Spins are bounded: a
// workgroup that never sees its peers (a grid that is not co-resident, a lost launch) sets *err and the launch runs to its
// end without waiting -- a wrong step that the host sees in the error word, never a hung GPU.
#define OMNI_CHAIN_WGS 256
#define OMNI_CHAIN_SPIN_BOUND (1u << 21)   // (round-3 arms of the debug library)
// the product chains wait by WALL CLOCK: 2 s of the constant 100 MHz counter, read once per 1024 polls -- a poll is
// a memory round trip of 0.5-2 us, so the first reading comes after about a millisecond of waiting and costs a healthy stage nothing
#define OMNI_CHAIN_WAIT_TICKS 200000000ull
__device__ __forceinline__ bool chain_wait_expired(unsigned spins, unsigned long long& t0) {
    if ((spins & 1023u) != 0) return false;
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (spins == 1024u) { t0 = now; return false; }
    return now - t0 > OMNI_CHAIN_WAIT_TICKS;
}

// ---- loader / consumer engine (bb_engine.hip): waves 0-7 of a workgroup compute, waves 8-11 stream weights into an LDS FIFO
// by LDS-DMA.  All intra-workgroup synchronisation is through these LDS words (s_barrier would stop the loader waves too):
// monotonic counters, relaxed polls, every wait bounded.
#define ENG_FIFO_PIECES 96           // 1 KB pieces (one MFMA operand fragment of one wave each) in the FIFO: 96 KB
#define ENG_SPIN_BOUND (1u << 24)
struct EngSync {
    unsigned loaded[4];              // per loader wave: its pieces that have landed in LDS
    unsigned consumed[8];            // per compute wave: every piece below this global index has been read by it
    unsigned bar;                    // compute-wave barrier: arrivals
    unsigned dead;                   // a bounded wait ran out (results invalid; the launch runs to its end)
};

struct ChainGate {
    coh_rsrc_t frs;        // flag words [OMNI_CHAIN_WGS]
    int wg;                // the workgroup this gate speaks for: blockIdx.x, or a VIRTUAL one (bb_chain.hip's half grid: 128 workgroups play 256)
    uint32_t epoch;        // stages this workgroup has completed
    int32_t* err;
    bool dead;             // a spin ran out (here or in an earlier launch): stop waiting
    int dom;               // log2 of the flag domain: 6 = the row group's 64 workgroups, 7 = a pair of groups, 8 = all 256
    int nap;               // s_sleep units (64 clocks) between two polls: 0, 1, 2, 4, 8
    int ahead;             // stages this workgroup has passed without rows since it last waited (chain_gate_skip)
    int skip_units;        // sleep per skipped stage before the first poll, in units of 32 x 64 clocks (3 = ~2.9 us; A/B knob)
    int skip;              // debug library only (timing experiments, results garbage): 1 = fetch half of every weight slice, 2 = half of the activations
    unsigned long long* stamps;   // debug library only: timeline stamps of the attention stage (pa_body.cuh, CHAIN), or NULL
    // engine mode (NULL / unused in the plain chains): barriers among the 8 compute waves only, weights from the LDS FIFO
    EngSync* es;
    unsigned bgen;         // compute-wave barriers this wave has passed
    const uint8_t* fifo;   // LDS FIFO base
    unsigned piece_base;   // global index of the current stage's first piece
    unsigned ready;        // pieces known to have landed (cache of 4 * min(loaded[]))
};

// workgroup barrier of the chain code: s_barrier, or -- engine mode -- arrivals of the 8 compute waves on an LDS counter
__device__ __forceinline__ void chain_barrier(ChainGate& g) {
    if (g.es == nullptr) {
        __syncthreads();
        return;
    }
    // LDS operations of a wave execute in order and the words below are relaxed atomics: what is needed is that the COMPILER
    // keeps the order (a workgroup-scope fence would also drain the wave's outstanding global loads: vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    g.bgen += 1;
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&g.es->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned target = g.bgen * 8u;
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(&g.es->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > ENG_SPIN_BOUND) {
            __hip_atomic_store(&g.es->dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
    }
    asm volatile("" ::: "memory");
}

// engine mode: wait until every FIFO piece below `upto` has landed; publish that this wave has read everything below `upto`
__device__ __forceinline__ void eng_wait_ready(ChainGate& g, unsigned upto) {
    unsigned spins = 0;
    while ((int)(g.ready - upto) < 0) {
        unsigned m = __hip_atomic_load(&g.es->loaded[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int q = 1; q < 4; ++q) m = min(m, __hip_atomic_load(&g.es->loaded[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        g.ready = 4u * m;             // loader wave q owns pieces q, q + 4, ...: everything below 4 * min is in LDS
        if ((int)(g.ready - upto) < 0) __builtin_amdgcn_s_sleep(1);
        if (++spins > ENG_SPIN_BOUND) {
            __hip_atomic_store(&g.es->dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
    }
    asm volatile("" ::: "memory");
}
// the FIFO reads of this wave were issued before this store and LDS executes a wave's operations in order: a relaxed store
// behind a compiler barrier is enough (a release store would wait for the wave's global loads in flight too)
__device__ __forceinline__ void eng_release(ChainGate& g, int wave, unsigned upto) {
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(&g.es->consumed[wave], upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void chain_gate_init(ChainGate& g, uint32_t* flags, int32_t* err, int wg = blockIdx.x) {
    g.frs = coh_rsrc(flags);
    g.err = err;
    g.wg = wg;
    g.epoch = __builtin_amdgcn_readfirstlane(coh_ld4(g.frs, wg * 4));
    g.dead = __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0;
    g.dom = 8;
    g.nap = 1;
    g.skip_units = 3;
    g.skip = 0;
    g.stamps = nullptr;
    g.ahead = 0;
    g.es = nullptr; g.bgen = 0; g.fifo = nullptr; g.piece_base = 0; g.ready = 0;
}

// wait until every workgroup of this workgroup's flag DOMAIN has completed the stage this workgroup completed last (g.epoch).
// Domain = 2^dom consecutive workgroups: the 64 workgroups w with w / 64 == blockIdx.x / 64 own the same 16 batch rows in
// every 16-row-tile stage, and no stage reads another row group's rows -- so a chain whose stages all use 16-row tiles needs
// dom = 6 only (four independent chains of 64 workgroups); 32-row tiles tie two groups together (dom = 7).  Wave 0 polls,
// the other waves wait at the barrier.
// A workgroup without rows in a stage (a partly filled batch) does not wait there at all: it publishes the stage at once and
// runs AHEAD of the others (chain_gate_skip) -- nobody ever waits for it, and it does not poll.  Polling idle workgroups hammer the
// few memory lines the flags live in and the busy workgroups' flag stores queue behind them: at 48 of 64 rows the hand-off gap
// grew from 1.2-1.9 to 3-6 us and the chain lost to the launch path.  When it reaches a stage where it HAS rows it is `ahead`
// stages early: it sleeps ~2.5 us per skipped stage (a stage takes 3-5.5 us) before its first poll.
__device__ __forceinline__ void chain_gate_wait(ChainGate& g, int code) {
    if (threadIdx.x < 64 && !g.dead) {
        for (int i = 0; i < g.ahead * g.skip_units; ++i) __builtin_amdgcn_s_sleep(32);
        // (1 << dom) / 4 lanes x 4 flags each (the other lanes read copies)
        const uint32_t off = chain_flag_copy() + ((g.wg >> g.dom) << (g.dom + 2)) + (threadIdx.x & ((1u << (g.dom - 2)) - 1)) * 16;
        unsigned spins = 0;
        unsigned long long t0 = 0;
        if (g.nap >= 16) {
            // two polls in flight, half a round trip apart: the flags are sampled twice as often as one load's latency allows
            u32x4 fa = coh_ld16(g.frs, off);
            for (;;) {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_sleep(4);
                const u32x4 fb = coh_ld16(g.frs, off);
                const bool behind = (int)(fa[0] - g.epoch) < 0 || (int)(fa[1] - g.epoch) < 0 || (int)(fa[2] - g.epoch) < 0 ||
                                    (int)(fa[3] - g.epoch) < 0;
                if (__builtin_amdgcn_ballot_w64(behind) == 0) break;
                fa = fb;
                if (chain_wait_expired(++spins, t0)) {
                    if (threadIdx.x == 0) atomicCAS(g.err, 0, code);
                    g.dead = true;
                    break;
                }
            }
        } else
        for (;;) {
            asm volatile("" ::: "memory");                // the buffer load is a plain read to the compiler: keep it inside the loop
            const u32x4 f = coh_ld16(g.frs, off);
            const bool behind = (int)(f[0] - g.epoch) < 0 || (int)(f[1] - g.epoch) < 0 || (int)(f[2] - g.epoch) < 0 ||
                                (int)(f[3] - g.epoch) < 0;
            if (__builtin_amdgcn_ballot_w64(behind) == 0) break;
            // back-to-back polls from 256 waves slow every other memory access of the chip down (measured: the whole
            // predictor 1.90 -> 2.55 ms without any pause)
            if (g.nap == 1) __builtin_amdgcn_s_sleep(1);
            else if (g.nap == 2) __builtin_amdgcn_s_sleep(2);
            else if (g.nap == 4) __builtin_amdgcn_s_sleep(4);
            else if (g.nap >= 8) __builtin_amdgcn_s_sleep(8);
            if (chain_wait_expired(++spins, t0)) {
                if (threadIdx.x == 0) atomicCAS(g.err, 0, code);
                g.dead = true;
                break;
            }
        }
    }
    g.ahead = 0;
    chain_barrier(g);
}

// this workgroup's stores of the stage are out (every wave drains its own), then one lane publishes.  KEEP = loads the wave
// issued BEHIND its stores (the next stage's weight prefetch) that may stay in flight: the queue is in order, so vmcnt(KEEP)
// covers exactly the stores
template <int KEEP = 0>
__device__ __forceinline__ void chain_gate_arrive(ChainGate& g) {
    static_assert(KEEP >= 0 && KEEP <= 48, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(KEEP) : "memory");
    if (KEEP > 0 && g.es == nullptr) {
        // bare s_barrier: __syncthreads() carries workgroup-scope fences, and the release fence would wait for the KEEP loads
        // this wave wants to keep in flight.  Nothing of this stage crosses the barrier through memory any more: every wave has
        // just waited for its own stores, and the waves exchange nothing through LDS here.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
        chain_barrier(g);
    }
    g.epoch += 1;
    if (threadIdx.x < 64) chain_flag_publish(g.frs, g.wg, g.epoch);
}

// a stage in which this workgroup has no rows: no wait, no work -- publish it and run ahead (see chain_gate_wait)
__device__ __forceinline__ void chain_gate_skip(ChainGate& g) {
    g.ahead = g.ahead < 12 ? g.ahead + 1 : 12;
    g.epoch += 1;
    if (threadIdx.x < 64) chain_flag_publish(g.frs, g.wg, g.epoch);
}
