// The backbone segment o_proj -> gate_up -> down_proj -> next qkv (see bb_chain.hip) as a LOADER / CONSUMER engine.
//
// What bounds the launch path and the plain chain alike is per-CU ingest: a layer makes every CU pull ~390 KB of weights from
// HBM (~24 GB/s per CU) and ~900 KB of activations from L2 (~70 GB/s per CU), and a wave's loads return IN ORDER -- a compute
// wave that has a slow weight load outstanding cannot consume the fast activation load behind it, nor the flag poll behind
// that.  So the two streams are given to different waves:
//   waves 8-11 (loaders)  walk the segment's weight slice as ONE sequence of 1 KB pieces (a piece = the MFMA operand fragment
//                         of one n-tile at one k-step) in exactly the order the compute waves consume it, and move it HBM -> LDS
//                         by LDS-DMA (no registers) into a 96 KB FIFO, ~12 pieces (48 KB per CU) in flight.  They never wait for
//                         a stage flag: weights do not depend on anything, so the stream runs on through the stage hand-offs,
//                         up to 96 KB ahead.
//   waves 0-7 (compute)   chain_gemm with WSRC = 2: weights by ds_read_b128 from the FIFO, activations by sc1 loads behind the
//                         stage flags (their load queue holds nothing slow), same k-step ownership, accumulation order,
//                         combine order and rounding points as gemm_skinny_kernel: bit-identical to the launch path.
// Synchronisation inside the workgroup is LDS words only (coherent.cuh EngSync): `loaded[q]` (pieces of loader q that have
// landed: published behind the covering vmcnt), `consumed[w]` (piece index below which compute wave w has read everything),
// and an arrival counter as the compute waves' barrier (s_barrier would stop the loaders).  Every wait is bounded.
// Reference: the decoder layer vLLM's Qwen3Model runs under qwen3_tts_talker.py:341,414-422.
#include "chain_gemm.cuh"
#include "common.cuh"
#include "kernels.h"

#define ENG_LOADERS 4
#define ENG_THREADS (CH_THREADS + ENG_LOADERS * 64)
#define ENG_COMBINE_BYTES (CH_WAVES * 6 * 64 * 16 + CH_WAVES * 64 * 4)            // 6 tiles per combine pass + rstd area
#define ENG_LDS_BYTES (ENG_FIFO_PIECES * 1024 + ENG_COMBINE_BYTES + 256)
#define ENG_GROUP 3                    // pieces a loader issues between two vmcnt checks
#define ENG_INFLIGHT 18                // pieces a loader keeps in flight (x 4 loaders x 1 KB = 72 KB per CU: ~24 GB/s at ~3 us of loaded HBM latency)

typedef __attribute__((address_space(3))) void eng_lds_void_t;

struct EngArgs {
    const uint16_t *wo, *ln2, *wgu, *wdown, *ln1_next, *wqkv_next;      // *_next == NULL: last layer, no qkv stage
    const uint16_t* attn;
    uint16_t* resid; float* part;
    uint16_t* act;
    uint16_t* qkv;
    int B, nap; float eps;
    uint32_t* flags; int32_t* err;
    unsigned long long* stamps;
};

// LDS words as the LOADER touches them: inline asm, invisible to hipcc's wait insertion -- for an LDS access it can see behind an
// LDS-DMA it emits `s_waitcnt vmcnt(0)` (it cannot prove that the access misses the DMA's destination), which would drain the
// loader's whole queue at every poll and every publish (measured: 8 GB/s per CU instead of ~24).
typedef __attribute__((address_space(3))) const void eng_lds_cvoid_t;
__device__ __forceinline__ uint32_t eng_lds_off(const void* p) { return (uint32_t)(uintptr_t)(eng_lds_cvoid_t*)p; }
__device__ __forceinline__ void eng_lds_store(uint32_t off, uint32_t v) { asm volatile("ds_write_b32 %0, %1" : : "v"(off), "v"(v) : "memory"); }
__device__ __forceinline__ u32x4 eng_lds_load16(uint32_t off) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
    return v;
}

// ---- loader wave q: pieces q, q + 4, q + 8, ... of the segment's stream
__device__ __forceinline__ void eng_loader(const EngArgs& a, EngSync* es, uint8_t* fifo, int q) {
    const int lane = threadIdx.x & 63;
    const int wg = blockIdx.x;
    // the stream: per stage `rounds` rounds of 8 NT pieces, piece p of a round = (tile p / 8, k-step 8 r + p % 8)
    unsigned n0 = 0;                   // global index of the current round's first piece
    unsigned mine = 0;                 // pieces this wave has issued
    unsigned freed = 0;                // pieces known to be consumed by every compute wave
    bool dead = false;
    const uint32_t consumed_off = eng_lds_off(&es->consumed[0]), loaded_off = eng_lds_off(&es->loaded[q]), dead_off = eng_lds_off(&es->dead);
    unsigned long long waited = 0;     // debug stamps: polls spent waiting for FIFO space
    unsigned long long* stamps = a.stamps;
    (void)waited;
    auto stage = [&](const uint16_t* W, int tile0, int NT, int rounds, int nsteps) {      // scalars only: no array, no scratch (a scratch
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);   // access would count in vmcnt)
        const int per_round = 8 * NT;
        for (int r = 0; r < rounds; ++r, n0 += per_round) {
            for (int p = q; p < per_round; p += ENG_LOADERS) {
                const unsigned n = n0 + p;
                // the FIFO slot of piece n is free once piece n - 96 has been read by every compute wave
                if (n >= ENG_FIFO_PIECES && (int)(freed - (n - ENG_FIFO_PIECES + 1)) < 0) {
                    // about to block: whatever this wave has issued must become visible first (pieces are published only
                    // once newer ones are behind them in the queue -- a blocked loader issues none, and the compute waves
                    // would wait for landed pieces they were never told about)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) eng_lds_store(loaded_off, mine);
                    unsigned spins = 0;
                    while (!dead && (int)(freed - (n - ENG_FIFO_PIECES + 1)) < 0) {
                        __builtin_amdgcn_s_sleep(8);                   // nothing to do: leave the issue slots to the compute waves
                        ++waited;
                        const u32x4 c0 = eng_lds_load16(consumed_off), c1 = eng_lds_load16(consumed_off + 16);
                        freed = min(min(min(c0[0], c0[1]), min(c0[2], c0[3])), min(min(c1[0], c1[1]), min(c1[2], c1[3])));
                        if (++spins > ENG_SPIN_BOUND) {
                            eng_lds_store(dead_off, 2u);
                            dead = true;
                        }
                    }
                    asm volatile("" ::: "memory");
                }
                const unsigned src = (unsigned)((tile0 + (p >> 3)) * nsteps + 8 * r + (p & 7)) * 1024u + lane * 16u;
                uint8_t* dst = fifo + (size_t)(n % ENG_FIFO_PIECES) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (eng_lds_void_t*)dst, 16, src, 0, 0, 0);
                ++mine;
                if (mine % ENG_GROUP == 0) {
                    // all but the newest ENG_INFLIGHT - ENG_GROUP pieces of this wave have landed
                    static_assert(ENG_INFLIGHT - ENG_GROUP == 15, "the immediate below");
                    asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                    if (mine > ENG_INFLIGHT - ENG_GROUP && lane == 0) eng_lds_store(loaded_off, mine - (ENG_INFLIGHT - ENG_GROUP));   // the wait above IS the ordering
                }
            }
        }
    };
#ifdef OMNI_DEBUG_HOOKS
#define ENG_LSTAMP(k) do { if (stamps && q == 0 && lane == 0) stamps[((size_t)32 * CH_NSTAMP + (k)) * OMNI_CHAIN_WGS + blockIdx.x] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ENG_LSTAMP(k) do { (void)stamps; } while (0)
#endif
    ENG_LSTAMP(0);
    stage(a.wo, wg & 127, 1, 8, 64);
    ENG_LSTAMP(1);
    stage(a.wgu, wg * 3, 3, 8, 64);
    ENG_LSTAMP(2);
    stage(a.wdown, wg & 127, 1, 24, 192);
    ENG_LSTAMP(3);
    if (a.wqkv_next) stage(a.wqkv_next, (wg & 127) * 2, 2, 8, 64);
    ENG_LSTAMP(4);
#ifdef OMNI_DEBUG_HOOKS
    if (stamps && q == 0 && lane == 0) stamps[((size_t)32 * CH_NSTAMP + 5) * OMNI_CHAIN_WGS + blockIdx.x] = waited;
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) eng_lds_store(loaded_off, mine);
}

__global__ __launch_bounds__(ENG_THREADS) void bb_engine_kernel(const EngArgs a) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    uint8_t* fifo = smem;
    float* lds = reinterpret_cast<float*>(smem + ENG_FIFO_PIECES * 1024);
    EngSync* es = reinterpret_cast<EngSync*>(smem + ENG_FIFO_PIECES * 1024 + ENG_COMBINE_BYTES);
    if (threadIdx.x < 16) reinterpret_cast<unsigned*>(es)[threadIdx.x] = 0;
    __syncthreads();                   // the only s_barrier of the launch: the sync words are zero before anyone uses them
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wave-uniform (the loader's control flow is scalar)
    if (wave >= CH_WAVES) {
        eng_loader(a, es, fifo, wave - CH_WAVES);
        return;
    }
    ChainGate g;
    chain_gate_init(g, a.flags, a.err);
    g.dom = 8;
    g.nap = a.nap;
    g.es = es; g.fifo = fifo;
    const int wg = blockIdx.x;
    constexpr int H = 2048, I = 6144, NQ = 4096;
    // stage codes (error word): 0x2001 .. 0x2004; piece_base = the stream position of the stage's first piece
    g.piece_base = 0;
    chain_gemm<2, 1, 8, 0, OMNI_EPI_RESID, 0, 2>(a.wo, nullptr, a.attn, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, lds, g,
                                                 false, 0x2001, a.stamps);
    g.piece_base += 8 * 8 * 1;
    chain_gemm<4, 3, 8, 3, OMNI_EPI_SILU_MUL_GU8, 4, 2>(a.wgu, a.ln2, a.resid, a.part, H / 16, a.act, 0, nullptr, a.B, I, a.eps, wg, 0, lds, g,
                                                        true, 0x2002, a.stamps);
    g.piece_base += 8 * 8 * 3;
    chain_gemm<2, 1, 24, 0, OMNI_EPI_RESID, 8, 2>(a.wdown, nullptr, a.act, nullptr, 0, a.resid, 0, a.part, a.B, H, a.eps, wg & 127, wg >> 7, lds, g,
                                                  true, 0x2003, a.stamps);
    g.piece_base += 24 * 8 * 1;
    if (a.wqkv_next)
        chain_gemm<2, 2, 8, 3, OMNI_EPI_BF16, 0, 2>(a.wqkv_next, a.ln1_next, a.resid, a.part, H / 16, a.qkv, NQ, nullptr, a.B, NQ, a.eps, wg & 127,
                                                    wg >> 7, lds, g, true, 0x2004, a.stamps);
    // a bounded LDS wait that ran out goes into the global error word the host reads
    if (threadIdx.x == 0 && __hip_atomic_load(&es->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) atomicCAS(a.err, 0, 0x2fff);
}

OMNI_KNOB g_bb_engine = 0, g_eng_nap = 1;      // off: correct (bit-identical) but 2.03 ms of backbone per step against the plain chain's 1.56
#ifdef OMNI_DEBUG_HOOKS
static unsigned long long* g_eng_stamps = nullptr;
extern "C" void omni_debug_bb_engine(int on) { g_bb_engine = on; }
extern "C" void omni_debug_eng_stamps(void* buf) { g_eng_stamps = (unsigned long long*)buf; }
#endif

bool k_bb_engine_enabled() { return g_bb_engine != 0; }

int k_bb_engine(const omni_layer_weights& w, const omni_layer_weights* next, const void* attn, void* resid, float* part, void* act, void* qkv,
                int B, float eps, uint32_t* flags, int32_t* err, void* stream) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)bb_engine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ENG_LDS_BYTES);
        attr = true;
    }
    EngArgs a{};
    a.wo = (const uint16_t*)w.wo; a.ln2 = (const uint16_t*)w.ln2; a.wgu = (const uint16_t*)w.wgu; a.wdown = (const uint16_t*)w.wdown;
    a.ln1_next = next ? (const uint16_t*)next->ln1 : nullptr;
    a.wqkv_next = next ? (const uint16_t*)next->wqkv : nullptr;
    a.attn = (const uint16_t*)attn; a.resid = (uint16_t*)resid; a.part = part; a.act = (uint16_t*)act; a.qkv = (uint16_t*)qkv;
    a.B = B; a.nap = g_eng_nap; a.eps = eps; a.flags = flags; a.err = err;
#ifdef OMNI_DEBUG_HOOKS
    a.stamps = g_eng_stamps;
#endif
    hipLaunchKernelGGL(bb_engine_kernel, dim3(OMNI_CHAIN_WGS), dim3(ENG_THREADS), ENG_LDS_BYTES, (hipStream_t)stream, a);
    OMNI_CHECK_LAUNCH("bb_engine");
    return OMNI_OK;
}
