// Row sampler body shared by sample_kernel (sampler.hip: one 256-thread workgroup per row) and the persistent code-predictor
// chain (cp_chain.hip: 512-thread workgroups -- waves 4-7 ride along PASSIVE: they hold no elements and store nothing, but
// reach every workgroup barrier, so the 256 live threads compute bit for bit what the stand-alone kernel computes).
// Repetition penalty -> temperature -> top-k (ties kept) [-> top-p] -> Gumbel-max with the oracle's counter-based RNG;
// greedy = first argmax.  Oracle: talker_oracle.sample_row.
#pragma once
#include "common.cuh"

// debug library: in-kernel timeline of the pick (cp_chain.hip's stamp buffer, stage slot 27; scripts/step_timeline.py)
#ifdef OMNI_DEBUG_HOOKS
#define SMP_STAMP(buf, k)                                                                                   \
    do {                                                                                                    \
        if ((buf) != nullptr && threadIdx.x == 0) (buf)[((size_t)27 * 8 + (k)) * 256 + blockIdx.x] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define SMP_STAMP(buf, k) do { (void)(buf); } while (0)
#endif

#define SMP_THREADS 256            // LIVE threads of a row
#define SMP_WAVES (SMP_THREADS / 64)
#define SMP_MAXW 8                 // waves that may reach the barriers (live + passive)
#define SMP_MAXV 8192
#define SMP_PCAP 1024              // top-p candidate capacity (top_k + ties at the threshold)

__device__ __forceinline__ uint32_t ord_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_val(uint32_t k) {   // inverse of ord_key
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// LDS working set of one row, carved by the caller (NPT = elements per live thread: the row copy is the radix fallback's)
struct SmpLds {
    unsigned long long* ckey;      // [SMP_PCAP + 8]
    float* row;                    // [NPT * 256]
    uint32_t* hist;                // [4][256]
    float* cexp;                   // [SMP_PCAP + 8]
    float* sval; int* sidx;        // [SMP_MAXW]
    uint32_t* sel_prefix; uint32_t* sel_k;   // [4]
    uint32_t* wtot;                // [4][SMP_MAXW]
    int* coff;                     // [SMP_MAXW]
    float* wq; float* wmx;         // [SMP_MAXW]
    float* kth_s;                  // [1]
};
#define SMP_LDS_BYTES(NPT) ((SMP_PCAP + 8) * 12 + (NPT) * SMP_THREADS * 4 + 4 * 256 * 4 + 2 * SMP_MAXW * 4 + 32 + 4 * SMP_MAXW * 4 + 3 * SMP_MAXW * 4 + 16)
template <int NPT>
__device__ __forceinline__ SmpLds smp_carve(void* base) {
    char* p = reinterpret_cast<char*>(base);
    SmpLds L;
    L.ckey = reinterpret_cast<unsigned long long*>(p); p += (SMP_PCAP + 8) * 8;
    L.row = reinterpret_cast<float*>(p); p += NPT * SMP_THREADS * 4;
    L.hist = reinterpret_cast<uint32_t*>(p); p += 4 * 256 * 4;
    L.cexp = reinterpret_cast<float*>(p); p += (SMP_PCAP + 8) * 4;
    L.sval = reinterpret_cast<float*>(p); p += SMP_MAXW * 4;
    L.sidx = reinterpret_cast<int*>(p); p += SMP_MAXW * 4;
    L.sel_prefix = reinterpret_cast<uint32_t*>(p); p += 16;
    L.sel_k = reinterpret_cast<uint32_t*>(p); p += 16;
    L.wtot = reinterpret_cast<uint32_t*>(p); p += 4 * SMP_MAXW * 4;
    L.coff = reinterpret_cast<int*>(p); p += SMP_MAXW * 4;
    L.wq = reinterpret_cast<float*>(p); p += SMP_MAXW * 4;
    L.wmx = reinterpret_cast<float*>(p); p += SMP_MAXW * 4;
    L.kth_s = reinterpret_cast<float*>(p);
    return L;
}

// block-wide argmax over the LIVE waves with smallest-index tie-break; result broadcast to all threads.  (value, index) pairs
// merge through DPP row steps and permlane swaps (any pairing that merges disjoint lane groups: the merge is associative and
// commutative), not ds_bpermute
__device__ __forceinline__ int block_argmax(float v, int idx, float* sval, int* sidx) {
    wave_argmax(v, idx);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sval[wave] = v; sidx[wave] = idx; }
    __syncthreads();
    float bv = sval[0];
    int bi = sidx[0];
#pragma unroll
    for (int w = 1; w < SMP_WAVES; ++w)
        if (sval[w] > bv || (sval[w] == bv && sidx[w] < bi)) { bv = sval[w]; bi = sidx[w]; }
    __syncthreads();
    return bi;
}

// The pick of one row.  xr[e] = element threadIdx.x + 256 e of the row AFTER penalty / temperature (-inf beyond V and in every
// passive thread); live = threadIdx.x < 256; step = the RNG counter of this draw.  EVERY thread of the workgroup calls this
// (workgroup barriers inside, under workgroup-uniform conditions only).
template <int NPT>
__device__ __forceinline__ int smp_pick(float (&xr)[NPT], int V, int greedy, int top_k, float top_p, uint32_t seed, uint32_t step,
                                        const SmpLds& S, bool live, unsigned long long* stamps = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    SMP_STAMP(stamps, 0);
    float* row = S.row;
    uint32_t (*hist)[256] = reinterpret_cast<uint32_t (*)[256]>(S.hist);
    float* sval = S.sval; int* sidx = S.sidx; uint32_t* sel_prefix = S.sel_prefix; uint32_t* sel_k = S.sel_k;
    uint32_t (*wtot)[SMP_MAXW] = reinterpret_cast<uint32_t (*)[SMP_MAXW]>(S.wtot);
    unsigned long long* ckey = S.ckey; float* cexp = S.cexp; int* coff = S.coff; float* wq = S.wq; float* wmx = S.wmx;
    int pick;
    if (greedy) {
        float bv = -INFINITY;
        int bi = 0x7FFFFFFF;
#pragma unroll
        for (int e = 0; e < NPT; ++e) {
            const int i = threadIdx.x + e * SMP_THREADS;
            if (xr[e] > bv) { bv = xr[e]; bi = i; }             // ascending i: the first maximum wins
        }
        if (bi == 0x7FFFFFFF) bi = threadIdx.x < V ? threadIdx.x : 0;   // all -inf / NaN row
        pick = block_argmax(bv, bi, sval, sidx);
    } else {
        // ---- top-k threshold kth (k-th largest value; ties kept) and, for top-p, the nucleus cut.
        // Fast path (top_k <= 256): a lower bound L of kth from the threads' local maxima filters the row down to a few
        // dozen candidates, which are then ranked exactly by counting -- no histogram atomics (the first radix digit of
        // fp32 logits hits 2-3 bins: ~2 k serialised LDS atomics) and 4 barriers instead of the radix select's 12.
        //   A  lm = max of this thread's elements; every wave takes the q-th largest of its 64 lm (q = ceil(k / 4), rank
        //      by counting over readlane broadcasts); L = min over the 4 waves: at least 4 q >= k elements are >= L.
        //   B  candidates {x >= L, x > -inf} compacted in thread-major order (shuffle scan) as 64-bit sort keys.
        //   C  rank of each candidate by (value desc, index asc) = number of candidates before it; rank k-1 holds kth;
        //      for top-p the same loop adds up the softmax mass that sorts before the candidate.
        //   D  Gumbel scores of the kept candidates only (one per thread), block argmax.
        // Fallback (top_k > 256, or > SMP_PCAP candidates, e.g. a constant row): 4-pass radix select over an LDS copy.
        const bool want_k = top_k > 0 && top_k < V;
        const bool want_p = want_k && top_p > 0.f && top_p < 1.f;
        // ckey: (ordered value key << 32) | ~index: larger = sorts first
        float kth = -INFINITY, L = -INFINITY, mx = -INFINITY;
        bool fast = want_k && top_k <= SMP_THREADS;
        bool have_list = false;
        int n = 0;
        auto radix_kth = [&]() -> float {
            if (live) {
#pragma unroll
                for (int e = 0; e < NPT; ++e) row[threadIdx.x + e * SMP_THREADS] = xr[e];
#pragma unroll
                for (int p = 0; p < 4; ++p) hist[p][threadIdx.x] = 0;
            }
            __syncthreads();
            // radix select of the top_k-th largest ordered key, 8 bits per pass
            uint32_t prefix = 0, mask = 0, krem = (uint32_t)top_k;
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                for (int i = live ? (int)threadIdx.x : V; i < V; i += SMP_THREADS) {
                    const uint32_t k = ord_key(row[i]);
                    if ((k & mask) == prefix) atomicAdd(&hist[pass][(k >> shift) & 0xFF], 1u);
                }
                __syncthreads();
                {
                    // thread t owns bin t: inclusive suffix sum S[t] = sum_{b >= t} hist[b]
                    const uint32_t cnt = live ? hist[pass][threadIdx.x] : 0u;
                    uint32_t sfx = cnt;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const uint32_t up = __shfl_down(sfx, o, 64);
                        if (lane + o < 64) sfx += up;
                    }
                    if (lane == 0) wtot[pass][wave] = sfx;
                    __syncthreads();
                    for (int w = wave + 1; w < SMP_THREADS / 64; ++w) sfx += wtot[pass][w];
                    if (live && sfx >= krem && sfx - cnt < krem) {  // exactly one bin satisfies this
                        sel_prefix[pass] = prefix | ((uint32_t)threadIdx.x << shift);
                        sel_k[pass] = krem - (sfx - cnt);
                    }
                }
                __syncthreads();
                prefix = sel_prefix[pass];
                krem = sel_k[pass];
                mask |= 0xFFu << shift;
            }
            return key_val(prefix);
        };
        auto compact = [&](float thr) -> int {
            // candidates {x >= thr, x > -inf} -> ckey (/ cexp) in thread-major order; returns their number
            int cnt = 0;
#pragma unroll
            for (int e = 0; e < NPT; ++e) cnt += (xr[e] >= thr && xr[e] > -INFINITY) ? 1 : 0;
            int inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(inc, d, 64);
                if (lane >= d) inc += up;
            }
            if (lane == 63) coff[wave] = inc;
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < SMP_THREADS / 64; ++w) {
                if (w < wave) base += coff[w];
                total += coff[w];
            }
            int o = base + inc - cnt;
            if (total <= SMP_PCAP || !fast) {
#pragma unroll
                for (int e = 0; e < NPT; ++e)
                    if (xr[e] >= thr && xr[e] > -INFINITY) {
                        if (o < SMP_PCAP) {
                            ckey[o] = ((unsigned long long)ord_key(xr[e]) << 32) | (uint32_t)(~(uint32_t)(threadIdx.x + e * SMP_THREADS));
                            if (want_p) cexp[o] = expf(xr[e] - mx);
                        }
                        ++o;
                    }
            }
            // pad to a multiple of 8 with keys that sort after everything (the rank loop runs 8 entries per trip)
            const int nn = min(total, SMP_PCAP);
            if (threadIdx.x < 8 && nn + (int)threadIdx.x < ((nn + 7) & ~7)) { ckey[nn + threadIdx.x] = 0ull; cexp[nn + threadIdx.x] = 0.f; }
            __syncthreads();
            return total;
        };
        constexpr int SLOTS = SMP_PCAP / SMP_THREADS;        // candidates per thread (fully unrolled: registers)
        float before[SLOTS];
        float z = 1.f;
        if (want_k) {
            if (fast || want_p) {
                // A: local maxima, the row maximum, the per-wave quota bound
                float lm = -INFINITY;
#pragma unroll
                for (int e = 0; e < NPT; ++e) lm = fmaxf(lm, xr[e]);
                const int q = (top_k + SMP_THREADS / 64 - 1) / (SMP_THREADS / 64);
                int rk = 0;
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lm), j));
                    rk += (o > lm || (o == lm && j < lane)) ? 1 : 0;
                }
                if (rk == q - 1) wq[wave] = lm;
                const float wm = wave_max(lm);
                if (lane == 0) wmx[wave] = wm;
                if (threadIdx.x == 0) *S.kth_s = -INFINITY;
                __syncthreads();
                mx = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
                L = fminf(fminf(wq[0], wq[1]), fminf(wq[2], wq[3]));
                SMP_STAMP(stamps, 1);
            }
            if (fast) {
                n = compact(L);
                if (n > SMP_PCAP) fast = false;                      // uniform: every thread sees the same total
                SMP_STAMP(stamps, 2);
            }
            if (!fast) {
                kth = radix_kth();
                if (want_p) n = min(compact(kth), SMP_PCAP);
            }
            if (fast || want_p) {
                have_list = true;
                // C: rank (and, for top-p, the softmax mass that sorts before) of every candidate
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    const int c = threadIdx.x + sl * SMP_THREADS;
                    before[sl] = 0.f;
                    if (live && c < n) {
                        const unsigned long long me = ckey[c];
                        int rank = 0;
                        float bf = 0.f;
                        for (int j0 = 0; j0 < n; j0 += 8) {          // 8 independent broadcast reads per trip
                            unsigned long long kj[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) kj[u] = ckey[j0 + u];
                            if (want_p) {
                                float ej[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) ej[u] = cexp[j0 + u];
#pragma unroll
                                for (int u = 0; u < 8; ++u) bf += (kj[u] > me) ? ej[u] : 0.f;
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u) rank += (kj[u] > me) ? 1 : 0;
                        }
                        before[sl] = bf;
                        if (fast && rank == top_k - 1) *S.kth_s = key_val((uint32_t)(me >> 32));
                    }
                }
                if (fast) {
                    __syncthreads();
                    kth = *S.kth_s;                                   // -inf when fewer than top_k finite values exist
                    SMP_STAMP(stamps, 3);
                }
                if (want_p) {
                    // partition sum over the kept candidates {x >= kth}; candidate c stays iff the mass before it is < top_p
                    z = 0.f;
                    for (int c = live ? (int)threadIdx.x : n; c < n; c += SMP_THREADS) z += (key_val((uint32_t)(ckey[c] >> 32)) >= kth) ? cexp[c] : 0.f;
                    z = wave_sum(z);
                    if (lane == 0) sval[wave] = z;
                    __syncthreads();
                    z = (sval[0] + sval[1]) + (sval[2] + sval[3]);
                    __syncthreads();                                  // sval is reused by the argmax below
                }
            }
        }
        float bv = -INFINITY;
        int bi = 0x7FFFFFFF;
        if (have_list) {
            // D: the kept candidates are the only elements with a finite score
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int c = threadIdx.x + sl * SMP_THREADS;
                if (live && c < n) {
                    const unsigned long long me = ckey[c];
                    const float x = key_val((uint32_t)(me >> 32));
                    const int i = (int)(~(uint32_t)me);
                    if (x >= kth && !(want_p && before[sl] / z >= top_p)) {
                        const float u = hash_uniform(seed, step, (uint32_t)i);
                        const float sc = x - logf(-logf(u));
                        if (sc > bv || (sc == bv && i < bi)) { bv = sc; bi = i; }
                    }
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < NPT; ++e) {
                const int i = threadIdx.x + e * SMP_THREADS;
                const float x = xr[e];
                if (x >= kth && x > -INFINITY) {
                    const float u = hash_uniform(seed, step, (uint32_t)i);
                    const float sc = x - logf(-logf(u));
                    if (sc > bv) { bv = sc; bi = i; }                // ascending i: the first maximum wins
                }
            }
        }
        if (bi == 0x7FFFFFFF) bi = 0;
        SMP_STAMP(stamps, 4);
        pick = block_argmax(bv, bi, sval, sidx);
        SMP_STAMP(stamps, 5);
#ifdef OMNI_DEBUG_HOOKS
        if (stamps != nullptr && threadIdx.x == 0) stamps[((size_t)27 * 8 + 6) * 256 + blockIdx.x] = (unsigned long long)n;      // candidates ranked
#endif
    }
    return pick;
}

// ---- Round 6: the pick of one row by ONE wave, no workgroup barrier (the code-predictor chain's sampler stage: 8.4 us of its 11.4 were the
// 4-wave pick above -- 6 barriers, a 64-step readlane rank loop, two LDS compaction passes; profiles/r06_step_timeline.txt).  Same function
// of the row as smp_pick without top-p (top-k with ties kept -> Gumbel-max with the counter RNG, smallest index on equal scores; greedy =
// first argmax), so both paths pick the same id bit for bit (tests/test_gpu_chain.py: chain == launch path).
//   lane l holds EPL elements: x[e] = element (e >> 2) * 256 + 4 l + (e & 3)  (16-byte loads), -inf past V
//   late_temp: x is BEFORE the temperature and bf16-exact (the chain's head GEMM rounds its logits to bf16): dividing by a moderate T > 0
//     keeps such values strictly ordered, so the k-th largest is found on the raw values and only the kept candidates are divided (2 instead
//     of EPL IEEE divisions per lane); otherwise x is final.
//   1  two group maxima per lane; L = largest 16-bit key prefix with >= k of the 128 group maxima at or above it (16 ballot-count
//      bisection steps): at least k elements are >= L
//   2  candidates {x >= L, x > -inf}: wave prefix scan, 64-bit sort keys (value key << 32 | ~index) into LDS, read back two per lane
//   3  exact k-th largest key among them by the same bisection (16 steps when every key is bf16-exact, else 32); ties kept
//   4  Gumbel scores of the kept candidates, wave argmax
// Rows that do not fit (no top-k; more than 128 candidates: massive ties) take the same steps over all EPL registers of every lane (slow,
// exact, still without a barrier).
#define SMP_WAVE_CAND 128
__device__ __forceinline__ int smp_count_ge(uint32_t a, uint32_t b, uint32_t t) {
    return __builtin_popcountll(__builtin_amdgcn_ballot_w64(a >= t)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(b >= t));
}
template <int EPL>
__device__ __forceinline__ int smp_pick_wave(const float (&x)[EPL], int V, int greedy, int top_k, bool late_temp, float temperature,
                                             uint32_t seed, uint32_t step, unsigned long long* cand /* LDS [SMP_WAVE_CAND + 64] */) {
    static_assert(EPL % 8 == 0, "smp_pick_wave: elements per lane");
    int lane = threadIdx.x & 63;
    // opaque to the optimiser: inside a persistent kernel's pass loop hipcc otherwise hoists the EPL per-element index values out of the
    // loop and keeps them in scratch for the whole launch (128 spilled VGPRs in cp_chain_kernel)
    asm volatile("" : "+v"(lane));
    auto index_of = [&](int e) { return (e >> 2) * 256 + 4 * lane + (e & 3); };
    float bv = -INFINITY;
    int bi = 0x7FFFFFFF;
    if (greedy) {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if (x[e] > bv) { bv = x[e]; bi = index_of(e); }        // a lane's indices ascend with e: its first maximum wins
        if (bi == 0x7FFFFFFF) bi = 4 * lane < V ? 4 * lane : 0x7FFFFFFE;
        wave_argmax(bv, bi);
        return bi >= V ? 0 : bi;
    }
    const bool want_k = top_k > 0 && top_k < V;
    int n = 0, cnt = 0, inc = 0;
    float L = -INFINITY;
    bool bounded = false;
    if (want_k) {
        // 1: lower bound of the k-th largest value
        float g0 = -INFINITY, g1 = -INFINITY;
#pragma unroll
        for (int e = 0; e < EPL / 2; ++e) { g0 = fmaxf(g0, x[e]); g1 = fmaxf(g1, x[EPL / 2 + e]); }
        const uint32_t k0 = ord_key(g0) >> 16, k1 = ord_key(g1) >> 16;
        uint32_t t = 0;
#pragma unroll
        for (int bit = 15; bit >= 0; --bit) {
            const uint32_t c = t | (1u << bit);
            if (smp_count_ge(k0, k1, c) >= top_k) t = c;
        }
        // a FINITE bound (key prefix above -inf's 0x007F) or none: fewer than k finite group maxima take the slow form below
        bounded = t >= 0x0080u;
        L = key_val(t << 16);
    }
    if (bounded) {
        // 2: compaction (x >= L implies x > -inf)
#pragma unroll
        for (int e = 0; e < EPL; ++e) cnt += x[e] >= L ? 1 : 0;
        // inclusive prefix over the 64 lanes: 4 DPP row shifts, then the totals of the rows below (lanes 15 / 31 / 47)
        int v = cnt;
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
        const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
        const int row = lane >> 4;
        inc = v + (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
        n = __builtin_amdgcn_readlane(inc, 63);
    }
    if (!bounded || n > SMP_WAVE_CAND) {
        // no top-k, or massive ties: every register of every lane competes.  The exact k-th largest key by bisection over all of them
        // (32 steps x EPL ballots); fewer than k finite values: kth stays 0 = everything finite
        float kt = -INFINITY;
        if (want_k) {
            uint32_t kth = 0;
#pragma unroll 1
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t c = kth | (1u << bit);
                int ge = 0;
#pragma unroll
                for (int e = 0; e < EPL; ++e)
                    ge += __builtin_popcountll(__builtin_amdgcn_ballot_w64(x[e] > -INFINITY && ord_key(x[e]) >= c));
                if (ge >= top_k) kth = c;
            }
            const float kth_f = kth ? key_val(kth) : -INFINITY;
            kt = late_temp ? kth_f / temperature : kth_f;
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {          // (fully unrolled: a dynamic register index would put x[] into scratch)
            const float xt = late_temp ? x[e] / temperature : x[e];
            if (xt >= kt && x[e] > -INFINITY) {
                const int i = index_of(e);
                const float sc = xt - logf(-logf(hash_uniform(seed, step, (uint32_t)i)));
                if (sc > bv || (sc == bv && i < bi)) { bv = sc; bi = i; }
            }
        }
    } else {
        // branch-free: every element is written as (value bits, ~index), the ones below L to this lane's dump slot behind the list (an
        // exec-masked store per element cost 17 instructions each, 544 of the pick's ~1500); byte offsets, sort keys built after the read-back
        char* const cb = reinterpret_cast<char*>(cand);
        const int dump = (SMP_WAVE_CAND + lane) * 8;
        int o8 = (inc - cnt) * 8;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const bool c = x[e] >= L;
            *reinterpret_cast<uint2*>(cb + (c ? o8 : dump)) = make_uint2(__float_as_uint(x[e]), ~(uint32_t)index_of(e));
            o8 += c ? 8 : 0;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint2 w0 = *reinterpret_cast<const uint2*>(cb + lane * 8), w1 = *reinterpret_cast<const uint2*>(cb + (lane + 64) * 8);
        // (slots past n hold stale bytes: key 0 sorts after everything)
        const uint32_t q0 = lane < n ? ord_key(__uint_as_float(w0.x)) : 0u, q1 = lane + 64 < n ? ord_key(__uint_as_float(w1.x)) : 0u;
        const unsigned long long c0 = ((unsigned long long)q0 << 32) | w0.y, c1 = ((unsigned long long)q1 << 32) | w1.y;
        // 3: the exact k-th largest key (ties kept); fewer than k candidates = fewer than k finite values: keep them all
        uint32_t kth = 0;
        if (n >= top_k) {
            // bf16-exact values (their float bits end in 16 zeros; the key of a negative one then ends in 16 ones): 16 steps on the key
            // prefixes; anything else: all 32 bits
            const bool wide = __builtin_amdgcn_ballot_w64((((lane < n ? w0.x : 0u) | (lane + 64 < n ? w1.x : 0u)) & 0xFFFFu) != 0) != 0;
            if (wide) {
#pragma unroll 1
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t c = kth | (1u << bit);
                    if (smp_count_ge(q0, q1, c) >= top_k) kth = c;
                }
            } else {
                const uint32_t h0 = q0 >> 16, h1 = q1 >> 16;
                uint32_t t16 = 0;
#pragma unroll
                for (int bit = 15; bit >= 0; --bit) {
                    const uint32_t c = t16 | (1u << bit);
                    if (smp_count_ge(h0, h1, c) >= top_k) t16 = c;
                }
                kth = (t16 << 16) | ((t16 & 0x8000u) ? 0u : 0xFFFFu);
            }
        }
        // 4: Gumbel scores of the kept candidates (value >= kth by FLOAT compare of the final values, as smp_pick keeps them)
        const float kth_f = n >= top_k ? key_val(kth) : -INFINITY;
        const float kt = late_temp ? kth_f / temperature : kth_f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const unsigned long long me = s2 ? c1 : c0;
            if ((uint32_t)(me >> 32) != 0u) {
                const float xv = key_val((uint32_t)(me >> 32));
                const int i = (int)(~(uint32_t)me);
                const float xt = late_temp ? xv / temperature : xv;
                if (xt >= kt) {
                    const float sc = xt - logf(-logf(hash_uniform(seed, step, (uint32_t)i)));
                    if (sc > bv || (sc == bv && i < bi)) { bv = sc; bi = i; }
                }
            }
        }
    }
    wave_argmax(bv, bi);
    return bi == 0x7FFFFFFF ? 0 : bi;
}
