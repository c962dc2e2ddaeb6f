// Row sampler: repetition penalty -> temperature -> top-k (radix select, ties kept) -> Gumbel-max
// with the oracle's counter-based RNG; greedy = first argmax.  One 256-thread workgroup per row,
// the row (V <= 8192 floats) staged once in LDS.  Oracle: talker_oracle.sample_row.
#include "common.cuh"
#include "kernels.h"

#define SMP_THREADS 256
#define SMP_MAXV 8192

__device__ __forceinline__ uint32_t ord_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float key_val(uint32_t k) {   // inverse of ord_key
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// block-wide argmax with smallest-index tie-break; result broadcast to all threads.  (value, index) pairs merge through
// DPP row steps and permlane swaps (any pairing that merges disjoint lane groups: the merge is associative and
// commutative), not ds_bpermute
__device__ __forceinline__ int block_argmax(float v, int idx, float* sval, int* sidx) {
    wave_argmax(v, idx);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sval[wave] = v; sidx[wave] = idx; }
    __syncthreads();
    float bv = sval[0];
    int bi = sidx[0];
#pragma unroll
    for (int w = 1; w < SMP_THREADS / 64; ++w)
        if (sval[w] > bv || (sval[w] == bv && sidx[w] < bi)) { bv = sval[w]; bi = sidx[w]; }
    __syncthreads();
    return bi;
}

#define SMP_PCAP 1024      // top-p candidate capacity (top_k + ties at the threshold)

// NPT = elements per thread (V <= NPT * 256): the row lives in registers, element e of thread t is index t + 256 e
template <int NPT>
__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(const float* __restrict__ logits, int ld, int V, int greedy,
                                                             float temperature, int top_k, float top_p, float rep_penalty,
                                                             uint8_t* __restrict__ seen, uint32_t seed,
                                                             int32_t* __restrict__ steps, int step_mul, int step_add,
                                                             int inc_steps, int32_t* __restrict__ out_ids, int out_stride,
                                                             const uint16_t* __restrict__ gtab, uint16_t* __restrict__ gout, int gdim,
                                                             float* __restrict__ gpart, int32_t* __restrict__ inc0,
                                                             int32_t* __restrict__ inc1, const omni_row_sampling rs,
                                                             const int32_t* __restrict__ num_live) {
    __shared__ float row[NPT * SMP_THREADS];       // LDS copy of the row: radix fallback only
    __shared__ uint32_t hist[4][256];              // one histogram per radix pass, cleared once
    __shared__ float sval[SMP_THREADS / 64];
    __shared__ int sidx[SMP_THREADS / 64];
    __shared__ uint32_t sel_prefix[4], sel_k[4];   // per pass: no barrier needed before the next pass overwrites
    __shared__ uint32_t wtot[4][SMP_THREADS / 64];
    const int b = blockIdx.x;
    // rows past the live count of a padded graph bucket leave no trace (block-uniform exit, before any barrier)
    if (num_live && b >= *num_live) return;
    // per-request sampling parameters (V/worker/gpu_model_runner.py:315-319): a NULL array = the launch-wide scalar
    if (rs.greedy) greedy = rs.greedy[b];
    if (rs.temperature) temperature = rs.temperature[b];
    if (rs.top_k) top_k = rs.top_k[b];
    if (rs.top_p) top_p = rs.top_p[b];
    if (rs.rep_penalty) rep_penalty = rs.rep_penalty[b];
    if (rs.seed) seed = rs.seed[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* src = logits + (size_t)b * ld;
    uint8_t* sn = seen ? seen + (size_t)b * V : nullptr;
    float xr[NPT];
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
        const int i = threadIdx.x + e * SMP_THREADS;
        float x = -INFINITY;                        // padding past V: never a candidate, never an argmax
        if (i < V) {
            x = src[i];
            if (sn && rep_penalty != 1.0f && sn[i]) x = x > 0.f ? x / rep_penalty : x * rep_penalty;
            if (!greedy) x = x / temperature;
        }
        xr[e] = x;
    }
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
        __shared__ unsigned long long ckey[SMP_PCAP + 8];   // (ordered value key << 32) | ~index: larger = sorts first
        __shared__ float cexp[SMP_PCAP + 8];
        __shared__ int coff[SMP_THREADS / 64];
        __shared__ float wq[SMP_THREADS / 64], wmx[SMP_THREADS / 64], kth_s;
        float kth = -INFINITY, L = -INFINITY, mx = -INFINITY;
        bool fast = want_k && top_k <= SMP_THREADS;
        bool have_list = false;
        int n = 0;
        auto radix_kth = [&]() -> float {
#pragma unroll
            for (int e = 0; e < NPT; ++e) row[threadIdx.x + e * SMP_THREADS] = xr[e];
#pragma unroll
            for (int p = 0; p < 4; ++p) hist[p][threadIdx.x] = 0;
            __syncthreads();
            // radix select of the top_k-th largest ordered key, 8 bits per pass
            uint32_t prefix = 0, mask = 0, krem = (uint32_t)top_k;
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                for (int i = threadIdx.x; i < V; i += SMP_THREADS) {
                    const uint32_t k = ord_key(row[i]);
                    if ((k & mask) == prefix) atomicAdd(&hist[pass][(k >> shift) & 0xFF], 1u);
                }
                __syncthreads();
                {
                    // thread t owns bin t: inclusive suffix sum S[t] = sum_{b >= t} hist[b]
                    const uint32_t cnt = hist[pass][threadIdx.x];
                    uint32_t sfx = cnt;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const uint32_t up = __shfl_down(sfx, o, 64);
                        if (lane + o < 64) sfx += up;
                    }
                    if (lane == 0) wtot[pass][wave] = sfx;
                    __syncthreads();
                    for (int w = wave + 1; w < SMP_THREADS / 64; ++w) sfx += wtot[pass][w];
                    if (sfx >= krem && sfx - cnt < krem) {          // exactly one bin satisfies this
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
                if (threadIdx.x == 0) kth_s = -INFINITY;
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
                    if (c < n) {
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
                        if (fast && rank == top_k - 1) kth_s = key_val((uint32_t)(me >> 32));
                    }
                }
                if (fast) {
                    __syncthreads();
                    kth = kth_s;                                      // -inf when fewer than top_k finite values exist
                }
                if (want_p) {
                    // partition sum over the kept candidates {x >= kth}; candidate c stays iff the mass before it is < top_p
                    z = 0.f;
                    for (int c = threadIdx.x; c < n; c += SMP_THREADS) z += (key_val((uint32_t)(ckey[c] >> 32)) >= kth) ? cexp[c] : 0.f;
                    z = wave_sum(z);
                    if (lane == 0) sval[wave] = z;
                    __syncthreads();
                    z = (sval[0] + sval[1]) + (sval[2] + sval[3]);
                    __syncthreads();                                  // sval is reused by the argmax below
                }
            }
        }
        const uint32_t step = (uint32_t)(steps ? steps[b] * step_mul + step_add : step_add);
        float bv = -INFINITY;
        int bi = 0x7FFFFFFF;
        if (have_list) {
            // D: the kept candidates are the only elements with a finite score
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int c = threadIdx.x + sl * SMP_THREADS;
                if (c < n) {
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
    if (gtab) {
        float ss = 0.f;
        for (int v = threadIdx.x; v < gdim / 8; v += SMP_THREADS) {
            const uint4 a = *reinterpret_cast<const uint4*>(gtab + (size_t)pick * gdim + v * 8);
            if (gpart) {
                const uint32_t* w = reinterpret_cast<const uint32_t*>(&a);
#pragma unroll
                for (int j = 0; j < 4; ++j) ss += bf_lo(w[j]) * bf_lo(w[j]) + bf_hi(w[j]) * bf_hi(w[j]);
                *reinterpret_cast<uint4*>(gout + frag_off(b, v * 8, gdim)) = a;
            } else {
                *reinterpret_cast<uint4*>(gout + (size_t)b * gdim + v * 8) = a;
            }
        }
        if (gpart) {        // sval is free again after block_argmax's trailing barrier
            ss = wave_sum(ss);
            if ((threadIdx.x & 63) == 0) sval[threadIdx.x >> 6] = ss;
            __syncthreads();
            if (threadIdx.x == 0) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < SMP_THREADS / 64; ++w) t += sval[w];
                gpart[b] = t;
            }
        }
    }
    if (threadIdx.x == 0) {
        out_ids[(size_t)b * out_stride] = pick;
        if (sn && pick >= 0 && pick < V) sn[pick] = 1;
        if (steps && inc_steps) steps[b] += 1;
        if (inc0) inc0[b] += 1;                      // the step's position / sequence-length advance rides along
        if (inc1) inc1[b] += 1;
    }
}

int k_sample_gather(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k, float top_p,
                    float rep_penalty, uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul, int step_add,
                    int inc_steps, int32_t* out_ids, int out_stride, const void* gather_table, void* gather_out,
                    int gather_dim, float* gather_part, void* stream, int32_t* inc0, int32_t* inc1, const omni_row_sampling* rows,
                    const int32_t* num_live) {
    OMNI_CHECK_ARG(logits && out_ids, "omni_sample: null pointer");
    const omni_row_sampling rs = rows ? *rows : omni_row_sampling{};
    OMNI_CHECK_ARG(V > 0 && V <= SMP_MAXV && ld >= V, "omni_sample: V=%d ld=%d (V <= %d)", V, ld, SMP_MAXV);
    // scalars are validated when they are the ones in use; per-row arrays are the caller's contract (device memory)
    OMNI_CHECK_ARG(greedy || rs.greedy || rs.temperature || temperature > 0.f, "omni_sample: temperature must be > 0 when sampling");
    OMNI_CHECK_ARG(rs.rep_penalty || rep_penalty > 0.f, "omni_sample: rep_penalty must be > 0");
    OMNI_CHECK_ARG(greedy || rs.greedy || rs.top_p || rs.top_k || !(top_p > 0.f && top_p < 1.f) || (top_k > 0 && top_k <= SMP_PCAP),
                   "omni_sample: top_p needs 0 < top_k <= %d (got top_k=%d)", SMP_PCAP, top_k);
    OMNI_CHECK_ARG(!gather_table || (gather_out && gather_dim % 8 == 0), "omni_sample: bad gather arguments");
    OMNI_CHECK_ARG(!gather_part || (gather_table && gather_dim % 32 == 0), "omni_sample: bad fragment-major gather arguments");
    if (B <= 0) return OMNI_OK;
#define SMP_LAUNCH(NPT_)                                                                                                  \
    hipLaunchKernelGGL(sample_kernel<NPT_>, dim3(B), dim3(SMP_THREADS), 0, (hipStream_t)stream, logits, ld, V, greedy,     \
                       temperature, top_k, top_p, rep_penalty, seen, seed, steps, step_mul, step_add, inc_steps, out_ids, \
                       out_stride, (const uint16_t*)gather_table, (uint16_t*)gather_out, gather_dim, gather_part, inc0, inc1, rs, \
                       num_live)
    if (V <= 8 * SMP_THREADS) SMP_LAUNCH(8);
    else if (V <= 12 * SMP_THREADS) SMP_LAUNCH(12);
    else SMP_LAUNCH(32);
#undef SMP_LAUNCH
    OMNI_CHECK_LAUNCH("omni_sample");
    return OMNI_OK;
}

int k_sample(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k, float top_p, float rep_penalty,
             uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul, int step_add, int inc_steps, int32_t* out_ids,
             int out_stride, void* stream, int32_t* inc0, int32_t* inc1, const omni_row_sampling* rows, const int32_t* num_live) {
    return k_sample_gather(logits, ld, B, V, greedy, temperature, top_k, top_p, rep_penalty, seen, seed, steps, step_mul, step_add,
                           inc_steps, out_ids, out_stride, nullptr, nullptr, 0, nullptr, stream, inc0, inc1, rows, num_live);
}

extern "C" int omni_sample(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k, float top_p,
                           float rep_penalty, uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul, int step_add,
                           int inc_steps, int32_t* out_ids, void* stream) {
    return k_sample(logits, ld, B, V, greedy, temperature, top_k, top_p, rep_penalty, seen, seed, steps, step_mul, step_add,
                    inc_steps, out_ids, 1, stream);
}

extern "C" int omni_sample_rows(const float* logits, int ld, int B, int V, const omni_row_sampling* rows, uint8_t* seen,
                                int32_t* steps, int step_mul, int step_add, int inc_steps, int32_t* out_ids, void* stream) {
    OMNI_CHECK_ARG(rows && rows->greedy && rows->temperature && rows->top_k && rows->top_p && rows->rep_penalty && rows->seed,
                   "omni_sample_rows: every per-row array is required");
    return k_sample(logits, ld, B, V, 1, 1.0f, 0, 1.0f, 1.0f, seen, 0, steps, step_mul, step_add, inc_steps, out_ids, 1, stream,
                    nullptr, nullptr, rows, nullptr);
}
