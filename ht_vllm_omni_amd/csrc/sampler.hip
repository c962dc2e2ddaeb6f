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

// block-wide argmax with smallest-index tie-break; result broadcast to all threads
__device__ __forceinline__ int block_argmax(float v, int idx, float* sval, int* sidx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
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

__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(const float* __restrict__ logits, int ld, int V, int greedy,
                                                             float temperature, int top_k, float top_p, float rep_penalty,
                                                             uint8_t* __restrict__ seen, uint32_t seed,
                                                             int32_t* __restrict__ steps, int step_mul, int step_add,
                                                             int inc_steps, int32_t* __restrict__ out_ids, int out_stride,
                                                             const uint16_t* __restrict__ gtab, uint16_t* __restrict__ gout, int gdim,
                                                             float* __restrict__ gpart) {
    __shared__ float row[SMP_MAXV];
    __shared__ uint32_t hist[4][256];              // one histogram per radix pass, cleared once
    __shared__ float sval[SMP_THREADS / 64];
    __shared__ int sidx[SMP_THREADS / 64];
    __shared__ uint32_t sel_prefix[4], sel_k[4];   // per pass: no barrier needed before the next pass overwrites
    __shared__ uint32_t wtot[4][SMP_THREADS / 64];
    const int b = blockIdx.x;
    const float* src = logits + (size_t)b * ld;
    uint8_t* sn = seen ? seen + (size_t)b * V : nullptr;
    for (int i = threadIdx.x; i < V; i += SMP_THREADS) {
        float x = src[i];
        if (sn && rep_penalty != 1.0f && sn[i]) x = x > 0.f ? x / rep_penalty : x * rep_penalty;
        if (!greedy) x = x / temperature;
        row[i] = x;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) hist[p][threadIdx.x] = 0;
    __syncthreads();
    int pick;
    if (greedy) {
        float bv = -INFINITY;
        int bi = 0x7FFFFFFF;
        for (int i = threadIdx.x; i < V; i += SMP_THREADS) {
            const float x = row[i];
            if (x > bv || (x == bv && i < bi)) { bv = x; bi = i; }
        }
        if (bi == 0x7FFFFFFF) bi = threadIdx.x < V ? threadIdx.x : 0;   // all -inf / NaN row
        pick = block_argmax(bv, bi, sval, sidx);
    } else {
        float kth = -INFINITY;
        if (top_k > 0 && top_k < V) {
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
                        if ((int)(threadIdx.x & 63) + o < 64) sfx += up;
                    }
                    if ((threadIdx.x & 63) == 0) wtot[pass][threadIdx.x >> 6] = sfx;
                    __syncthreads();
                    for (int w = (threadIdx.x >> 6) + 1; w < SMP_THREADS / 64; ++w) sfx += wtot[pass][w];
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
            const uint32_t kb = (prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix;
            kth = __uint_as_float(kb);
        }
        if (top_p > 0.f && top_p < 1.f && top_k > 0 && top_k < V) {
            // ---- nucleus cut among the top-k candidates {x >= kth}: candidate i stays iff the softmax mass of the
            // candidates that sort before it (value desc, index asc) is < top_p.  Candidates are compacted in a fixed
            // order (thread-major) so the sums are run-to-run deterministic; n <= top_k + ties, O(n^2 / 256) per thread.
            __shared__ float cval[SMP_PCAP], cexp[SMP_PCAP];
            __shared__ int cidx[SMP_PCAP];
            __shared__ int coff[SMP_THREADS / 64];
            int cnt = 0;
            for (int i = threadIdx.x; i < V; i += SMP_THREADS) cnt += (row[i] >= kth && row[i] > -INFINITY) ? 1 : 0;
            // exclusive prefix of the per-thread counts: shuffle scan inside each wave + the 4 wave totals through LDS
            int inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(inc, d, 64);
                if ((int)(threadIdx.x & 63) >= d) inc += up;
            }
            if ((threadIdx.x & 63) == 63) coff[threadIdx.x >> 6] = inc;
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < SMP_THREADS / 64; ++w) {
                if (w < (int)(threadIdx.x >> 6)) base += coff[w];
                total += coff[w];
            }
            int o = base + inc - cnt;
            const int n = min(total, SMP_PCAP);
            for (int i = threadIdx.x; i < V; i += SMP_THREADS)
                if (row[i] >= kth && row[i] > -INFINITY) {
                    if (o < SMP_PCAP) { cval[o] = row[i]; cidx[o] = i; }
                    ++o;
                }
            __syncthreads();
            // max and partition sum over the candidates by block reductions (a serial loop over n LDS reads per thread
            // cost ~2 us each)
            float mx = -INFINITY;
            for (int c = threadIdx.x; c < n; c += SMP_THREADS) mx = fmaxf(mx, cval[c]);
            mx = wave_max(mx);
            if ((threadIdx.x & 63) == 0) sval[threadIdx.x >> 6] = mx;
            __syncthreads();
            mx = fmaxf(fmaxf(sval[0], sval[1]), fmaxf(sval[2], sval[3]));
            __syncthreads();
            float z = 0.f;
            for (int c = threadIdx.x; c < n; c += SMP_THREADS) {
                const float e = expf(cval[c] - mx);
                cexp[c] = e;
                z += e;
            }
            z = wave_sum(z);
            if ((threadIdx.x & 63) == 0) sval[threadIdx.x >> 6] = z;
            __syncthreads();
            z = (sval[0] + sval[1]) + (sval[2] + sval[3]);
            // the kept set is a prefix of the sorted order: find the smallest kept value (and, among equal values, the
            // largest kept index) = the cut; every thread tests its candidates
            for (int c = threadIdx.x; c < n; c += SMP_THREADS) {
                const float v = cval[c];
                const int vi = cidx[c];
                float before = 0.f;
                for (int j = 0; j < n; ++j)
                    if (cval[j] > v || (cval[j] == v && cidx[j] < vi)) before += cexp[j];
                if (before / z >= top_p) row[vi] = -INFINITY;        // removed from the row the Gumbel stage reads
            }
            __syncthreads();
        }
        const uint32_t step = (uint32_t)(steps ? steps[b] * step_mul + step_add : step_add);
        float bv = -INFINITY;
        int bi = 0x7FFFFFFF;
        for (int i = threadIdx.x; i < V; i += SMP_THREADS) {
            const float x = row[i];
            float sc = -INFINITY;
            if (x >= kth && x > -INFINITY) {
                const float u = hash_uniform(seed, step, (uint32_t)i);
                sc = x - logf(-logf(u));
            }
            if (sc > bv || (sc == bv && i < bi)) { bv = sc; bi = i; }
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
    }
}

int k_sample_gather(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k, float top_p,
                    float rep_penalty, uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul, int step_add,
                    int inc_steps, int32_t* out_ids, int out_stride, const void* gather_table, void* gather_out,
                    int gather_dim, float* gather_part, void* stream) {
    OMNI_CHECK_ARG(logits && out_ids, "omni_sample: null pointer");
    OMNI_CHECK_ARG(V > 0 && V <= SMP_MAXV && ld >= V, "omni_sample: V=%d ld=%d (V <= %d)", V, ld, SMP_MAXV);
    OMNI_CHECK_ARG(greedy || temperature > 0.f, "omni_sample: temperature must be > 0 when sampling");
    OMNI_CHECK_ARG(rep_penalty > 0.f, "omni_sample: rep_penalty must be > 0");
    OMNI_CHECK_ARG(greedy || !(top_p > 0.f && top_p < 1.f) || (top_k > 0 && top_k <= SMP_PCAP),
                   "omni_sample: top_p needs 0 < top_k <= %d (got top_k=%d)", SMP_PCAP, top_k);
    OMNI_CHECK_ARG(!gather_table || (gather_out && gather_dim % 8 == 0), "omni_sample: bad gather arguments");
    OMNI_CHECK_ARG(!gather_part || (gather_table && gather_dim % 32 == 0), "omni_sample: bad fragment-major gather arguments");
    if (B <= 0) return OMNI_OK;
    hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(SMP_THREADS), 0, (hipStream_t)stream, logits, ld, V, greedy,
                       temperature, top_k, top_p, rep_penalty, seen, seed, steps, step_mul, step_add, inc_steps, out_ids,
                       out_stride, (const uint16_t*)gather_table, (uint16_t*)gather_out, gather_dim, gather_part);
    OMNI_CHECK_LAUNCH("omni_sample");
    return OMNI_OK;
}

int k_sample(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k, float top_p, float rep_penalty,
             uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul, int step_add, int inc_steps, int32_t* out_ids,
             int out_stride, void* stream) {
    return k_sample_gather(logits, ld, B, V, greedy, temperature, top_k, top_p, rep_penalty, seen, seed, steps, step_mul, step_add,
                           inc_steps, out_ids, out_stride, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int omni_sample(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k, float top_p,
                           float rep_penalty, uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul, int step_add,
                           int inc_steps, int32_t* out_ids, void* stream) {
    return k_sample(logits, ld, B, V, greedy, temperature, top_k, top_p, rep_penalty, seen, seed, steps, step_mul, step_add,
                    inc_steps, out_ids, 1, stream);
}
