// Row sampler body shared by sample_kernel (sampler.hip: one 256-thread workgroup per row) and the persistent code-predictor
// chain (cp_chain.hip: 512-thread workgroups -- waves 4-7 ride along PASSIVE: they hold no elements and store nothing, but
// reach every workgroup barrier, so the 256 live threads compute bit for bit what the stand-alone kernel computes).
// Repetition penalty -> temperature -> top-k (ties kept) [-> top-p] -> Gumbel-max with the oracle's counter-based RNG;
// greedy = first argmax.  Oracle: talker_oracle.sample_row.
#pragma once
#include "common.cuh"

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
                                        const SmpLds& S, bool live) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
            }
            if (fast) {
                n = compact(L);
                if (n > SMP_PCAP) fast = false;                      // uniform: every thread sees the same total
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
        pick = block_argmax(bv, bi, sval, sidx);
    }
    return pick;
}
