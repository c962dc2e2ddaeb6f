// Row sampler: repetition penalty -> temperature -> top-k (radix select, ties kept) -> Gumbel-max
// with the oracle's counter-based RNG; greedy = first argmax.  One 256-thread workgroup per row,
// the row (V <= 8192 floats) staged once in LDS.  Oracle: talker_oracle.sample_row.
#include "common.cuh"
#include "kernels.h"

#include "sampler_body.cuh"

// ---- Round 6: a row without top-p that fits a wave's registers (V <= 3072, 16-byte aligned) is picked by ONE wave (smp_pick_wave,
// sampler_body.cuh: no workgroup barrier; 16.8 -> ~6 us for the talker's 3072-entry row) -- wave 0 of the row's workgroup, the others leave.
// Same outputs as the 4-wave path below (ids, seen marks, counters, gathered row + slab): a row's pick is the same function of its logits.
// Rows whose logits are bf16-exact and unpenalised take the temperature late (only the kept candidates are divided).
template <int EPL>
__device__ __forceinline__ void smp_row_wave(int b, int lane, const float* __restrict__ logits, int ld, int V, int greedy, float temperature, int top_k,
                                             float rep_penalty, uint8_t* __restrict__ seen, uint32_t seed, int32_t* __restrict__ steps, int step_mul,
                                             int step_add, int inc_steps, int32_t* __restrict__ out_ids, int out_stride,
                                             const uint16_t* __restrict__ gtab, uint16_t* __restrict__ gout, int gdim, float* __restrict__ gpart,
                                             int32_t* __restrict__ inc0, int32_t* __restrict__ inc1, unsigned long long* cand) {
    const float* src = logits + (size_t)b * ld;
    uint8_t* sn = seen ? seen + (size_t)b * V : nullptr;
    const bool pen = sn != nullptr && rep_penalty != 1.0f;
    float xr[EPL];
    uint32_t lowbits = 0;
#pragma unroll
    for (int j = 0; j < EPL / 4; ++j) {
        const int i0 = j * 256 + 4 * lane;
        f32x4 v4 = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        uint32_t s4 = 0;
        if (i0 + 3 < V) {
            v4 = *reinterpret_cast<const f32x4*>(src + i0);
            if (pen) s4 = *reinterpret_cast<const uint32_t*>(sn + i0);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (i0 + c < V) {
                    v4[c] = src[i0 + c];
                    if (pen) s4 |= (uint32_t)sn[i0 + c] << (8 * c);
                }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float x = v4[c];
            if (pen && ((s4 >> (8 * c)) & 0xFFu)) x = x > 0.f ? x / rep_penalty : x * rep_penalty;
            lowbits |= i0 + c < V ? __float_as_uint(x) & 0xFFFFu : 0u;
            xr[j * 4 + c] = x;
        }
    }
    // late temperature: bf16-exact values under a moderate T stay strictly ordered through the division (smp_pick_wave)
    const bool late = !greedy && __builtin_amdgcn_ballot_w64(lowbits != 0) == 0 && temperature >= 1.0f / 64 && temperature <= 64.0f;
    if (!greedy && !late) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int e = 0; e < EPL; ++e) xr[e] = xr[e] / temperature;
    }
    const uint32_t step = (uint32_t)(steps ? steps[b] * step_mul + step_add : step_add);
    const int pick = smp_pick_wave<EPL>(xr, V, greedy, top_k, late, temperature, seed, step, cand);
    if (gtab) {
        // sample_kernel's order of additions: thread t of its 256 sums pieces t, t + 256, ...; one wave_sum per 64 threads; the four in order
        float ssw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w)
            for (int v = lane + 64 * w; v < gdim / 8; v += SMP_THREADS) {
                const uint4 a = *reinterpret_cast<const uint4*>(gtab + (size_t)pick * gdim + v * 8);
                if (gpart) {
                    const uint32_t* wd = reinterpret_cast<const uint32_t*>(&a);
#pragma unroll
                    for (int j = 0; j < 4; ++j) ssw[w] += bf_lo(wd[j]) * bf_lo(wd[j]) + bf_hi(wd[j]) * bf_hi(wd[j]);
                    *reinterpret_cast<uint4*>(gout + frag_off(b, v * 8, gdim)) = a;
                } else {
                    *reinterpret_cast<uint4*>(gout + (size_t)b * gdim + v * 8) = a;
                }
            }
        if (gpart) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += wave_sum(ssw[w]);
            if (lane == 0) gpart[b] = t;
        }
    }
    if (lane == 0) {
        out_ids[(size_t)b * out_stride] = pick;
        if (sn && pick >= 0 && pick < V) sn[pick] = 1;
        if (steps && inc_steps) steps[b] += 1;
        if (inc0) inc0[b] += 1;
        if (inc1) inc1[b] += 1;
    }
}

OMNI_KNOB g_sample_wave = 1;
#ifdef OMNI_DEBUG_HOOKS
extern "C" void omni_debug_sample_wave(int on) { g_sample_wave = on; }      // 0: every row on the 4-wave sample_kernel (round 5)
#endif

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
                                                             const int32_t* __restrict__ num_live, const omni_step_status stt, int wave_ok) {
    __shared__ __attribute__((aligned(16))) char smem[SMP_LDS_BYTES(NPT)];
    const SmpLds S = smp_carve<NPT>(smem);
    float* sval = S.sval;
    const int b = blockIdx.x;
    // the decode step's status words ride out with its last launch (every launch that could set them finished before this one)
    if (stt.dst && b == 0 && threadIdx.x == 0) {
        stt.dst[0] = stt.src0 ? __hip_atomic_load(stt.src0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        stt.dst[1] = stt.src1 ? __hip_atomic_load(stt.src1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        stt.dst[2] = stt.ran;
        stt.dst[3] = 0;
    }
    // rows past the live count of a padded graph bucket leave no trace (block-uniform exit, before any barrier)
    if (num_live && b >= *num_live) return;
    // per-request sampling parameters (V/worker/gpu_model_runner.py:315-319): a NULL array = the launch-wide scalar
    if (rs.greedy) greedy = rs.greedy[b];
    if (rs.temperature) temperature = rs.temperature[b];
    if (rs.top_k) top_k = rs.top_k[b];
    if (rs.top_p) top_p = rs.top_p[b];
    if (rs.rep_penalty) rep_penalty = rs.rep_penalty[b];
    if (rs.seed) seed = rs.seed[b];
    if constexpr (NPT <= 12) {
        // workgroup-uniform (row parameters).  A row under a repetition penalty stays on the 4-wave path: its values are no longer bf16-exact,
        // so all of them are divided twice (penalty, temperature) -- 96 IEEE divisions in one wave cost more than the barriers they save
        // (measured: the talker's 3072-entry row 16.8 -> 18.1 us)
        const bool pen_row = seen != nullptr && rep_penalty != 1.0f && !greedy;
        if (wave_ok && !pen_row && (greedy || !(top_p > 0.f && top_p < 1.f))) {
            if (threadIdx.x < 64)
                smp_row_wave<NPT * 4>(b, threadIdx.x, logits, ld, V, greedy, temperature, top_k, rep_penalty, seen, seed, steps, step_mul, step_add,
                                      inc_steps, out_ids, out_stride, gtab, gout, gdim, gpart, inc0, inc1, S.ckey);
            return;
        }
    }
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
    const uint32_t step = (uint32_t)(steps ? steps[b] * step_mul + step_add : step_add);
    const int pick = smp_pick<NPT>(xr, V, greedy, top_k, top_p, seed, step, S, true);
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
                    const int32_t* num_live, const omni_step_status* status) {
    OMNI_CHECK_ARG(logits && out_ids, "omni_sample: null pointer");
    const omni_row_sampling rs = rows ? *rows : omni_row_sampling{};
    const omni_step_status stt = status ? *status : omni_step_status{};
    OMNI_CHECK_ARG(V > 0 && V <= SMP_MAXV && ld >= V, "omni_sample: V=%d ld=%d (V <= %d)", V, ld, SMP_MAXV);
    // scalars are validated when they are the ones in use; per-row arrays are the caller's contract (device memory)
    OMNI_CHECK_ARG(greedy || rs.greedy || rs.temperature || temperature > 0.f, "omni_sample: temperature must be > 0 when sampling");
    OMNI_CHECK_ARG(rs.rep_penalty || rep_penalty > 0.f, "omni_sample: rep_penalty must be > 0");
    OMNI_CHECK_ARG(greedy || rs.greedy || rs.top_p || rs.top_k || !(top_p > 0.f && top_p < 1.f) || (top_k > 0 && top_k <= SMP_PCAP),
                   "omni_sample: top_p needs 0 < top_k <= %d (got top_k=%d)", SMP_PCAP, top_k);
    OMNI_CHECK_ARG(!gather_table || (gather_out && gather_dim % 8 == 0), "omni_sample: bad gather arguments");
    OMNI_CHECK_ARG(!gather_part || (gather_table && gather_dim % 32 == 0), "omni_sample: bad fragment-major gather arguments");
    if (B <= 0) return OMNI_OK;
    // rows without top-p may be picked by one wave (sample_kernel's per-row dispatch): 16-byte row loads need aligned rows
    const int wave_ok = g_sample_wave && V <= 48 * 64 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 &&
                        (seen == nullptr || (V % 4 == 0 && (reinterpret_cast<uintptr_t>(seen) & 3) == 0));
#define SMP_LAUNCH(NPT_)                                                                                                  \
    hipLaunchKernelGGL(sample_kernel<NPT_>, dim3(B), dim3(SMP_THREADS), 0, (hipStream_t)stream, logits, ld, V, greedy,     \
                       temperature, top_k, top_p, rep_penalty, seen, seed, steps, step_mul, step_add, inc_steps, out_ids, \
                       out_stride, (const uint16_t*)gather_table, (uint16_t*)gather_out, gather_dim, gather_part, inc0, inc1, rs, \
                       num_live, stt, wave_ok)
    if (V <= 8 * SMP_THREADS) SMP_LAUNCH(8);
    else if (V <= 12 * SMP_THREADS) SMP_LAUNCH(12);
    else SMP_LAUNCH(32);
#undef SMP_LAUNCH
    OMNI_CHECK_LAUNCH("omni_sample");
    return OMNI_OK;
}

int k_sample(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k, float top_p, float rep_penalty,
             uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul, int step_add, int inc_steps, int32_t* out_ids,
             int out_stride, void* stream, int32_t* inc0, int32_t* inc1, const omni_row_sampling* rows, const int32_t* num_live,
             const omni_step_status* status) {
    return k_sample_gather(logits, ld, B, V, greedy, temperature, top_k, top_p, rep_penalty, seen, seed, steps, step_mul, step_add,
                           inc_steps, out_ids, out_stride, nullptr, nullptr, 0, nullptr, stream, inc0, inc1, rows, num_live, status);
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
