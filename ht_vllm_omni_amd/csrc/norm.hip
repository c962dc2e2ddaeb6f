// Fused residual-add + RMSNorm and small row-wise helpers (HBM/L2-bound elementwise work).
// One wave per row, 16-byte loads, fp32 statistics, wave64 butterfly reduce.
// Numerics follow HF Qwen3RMSNorm (oracle: talker_oracle.rms_norm):
//   r = bf16(residual + delta);  out = w * bf16(r * rsqrt(mean(r^2) + eps))   (product rounds to bf16)
#include "common.cuh"
#include "kernels.h"

#define NORM_ROWS_PER_BLOCK 4

template <int MAXV>   // MAXV = max 8-element vectors per lane (hidden <= 64*8*MAXV)
__global__ __launch_bounds__(64 * NORM_ROWS_PER_BLOCK) void rmsnorm_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ delta, const uint16_t* residual, uint16_t* residual_out,
    const uint16_t* __restrict__ w, uint16_t* __restrict__ out, uint16_t* __restrict__ out_frag, int rows, int hidden,
    float eps, const int32_t* __restrict__ out_live) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * NORM_ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = hidden >> 3;
    const uint16_t* src = (residual ? residual : x) + (size_t)row * hidden;
    float v[MAXV][8];
    float ss = 0.f;
    uint4 wv[MAXV];                       // norm weights: issued up front, consumed after the reduction
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        wv[i] = (vi < nvec) ? *reinterpret_cast<const uint4*>(w + vi * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            uint4 a = *reinterpret_cast<const uint4*>(src + vi * 8);
            const uint32_t* aw = reinterpret_cast<const uint32_t*>(&a);
            if (delta) {
                uint4 d = *reinterpret_cast<const uint4*>(delta + (size_t)row * hidden + vi * 8);
                const uint32_t* dw = reinterpret_cast<const uint32_t*>(&d);
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = bfround(bf_lo(aw[j]) + bf_lo(dw[j]));
                    const float hi = bfround(bf_hi(aw[j]) + bf_hi(dw[j]));
                    v[i][2 * j] = lo;
                    v[i][2 * j + 1] = hi;
                    o[j] = pack_bf2(lo, hi);
                }
                *reinterpret_cast<uint4*>(residual_out + (size_t)row * hidden + vi * 8) = make_uint4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[i][2 * j] = bf_lo(aw[j]);
                    v[i][2 * j + 1] = bf_hi(aw[j]);
                }
                if (residual_out && residual_out != residual)
                    *reinterpret_cast<uint4*>(residual_out + (size_t)row * hidden + vi * 8) = a;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
        }
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)hidden + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            const uint32_t* wp = reinterpret_cast<const uint32_t*>(&wv[i]);
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = bf_lo(wp[j]) * bfround(v[i][2 * j] * rstd);
                const float hi = bf_hi(wp[j]) * bfround(v[i][2 * j + 1] * rstd);
                o[j] = pack_bf2(lo, hi);
            }
            const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
            if (out && (!out_live || row < *out_live)) *reinterpret_cast<uint4*>(out + (size_t)row * hidden + vi * 8) = ov;
            if (out_frag) *reinterpret_cast<uint4*>(out_frag + frag_off(row, vi * 8, hidden)) = ov;
        }
    }
}

int k_rmsnorm(const void* x, const void* delta, const void* residual, void* residual_out, const void* w, void* out,
              void* out_frag, int rows, int hidden, float eps, void* stream, const int32_t* out_live) {
    OMNI_CHECK_ARG(w && (out || out_frag) && (x || residual), "omni_rmsnorm: null pointer");
    OMNI_CHECK_ARG(!out_frag || hidden % 32 == 0, "omni_rmsnorm: fragment-major output needs hidden %% 32 == 0");
    OMNI_CHECK_ARG(!(delta && !(residual && residual_out)), "omni_rmsnorm: delta needs residual in/out");
    OMNI_CHECK_ARG(rows >= 0 && hidden > 0 && hidden % 8 == 0 && hidden <= 64 * 8 * 8,
                   "omni_rmsnorm: hidden=%d unsupported (multiple of 8, <= 4096)", hidden);
    if (rows == 0) return OMNI_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((rows + NORM_ROWS_PER_BLOCK - 1) / NORM_ROWS_PER_BLOCK), block(64 * NORM_ROWS_PER_BLOCK);
    const int nvec = hidden / 8;
    if (nvec <= 64 * 2)
        hipLaunchKernelGGL(rmsnorm_kernel<2>, grid, block, 0, st, (const uint16_t*)x, (const uint16_t*)delta,
                           (const uint16_t*)residual, (uint16_t*)residual_out, (const uint16_t*)w, (uint16_t*)out, (uint16_t*)out_frag, rows, hidden, eps, out_live);
    else if (nvec <= 64 * 4)
        hipLaunchKernelGGL(rmsnorm_kernel<4>, grid, block, 0, st, (const uint16_t*)x, (const uint16_t*)delta,
                           (const uint16_t*)residual, (uint16_t*)residual_out, (const uint16_t*)w, (uint16_t*)out, (uint16_t*)out_frag, rows, hidden, eps, out_live);
    else
        hipLaunchKernelGGL(rmsnorm_kernel<8>, grid, block, 0, st, (const uint16_t*)x, (const uint16_t*)delta,
                           (const uint16_t*)residual, (uint16_t*)residual_out, (const uint16_t*)w, (uint16_t*)out, (uint16_t*)out_frag, rows, hidden, eps, out_live);
    OMNI_CHECK_LAUNCH("omni_rmsnorm");
    return OMNI_OK;
}

extern "C" int omni_rmsnorm(const void* x, const void* delta, void* residual, const void* w, void* out, int rows,
                            int hidden, float eps, void* stream) {
    return k_rmsnorm(x, delta, residual, residual, w, out, nullptr, rows, hidden, eps, stream);
}

// ---- embedding gather: out[t] = table[ids[t]] (ids outside [0,vocab) -> zeros)
__global__ void embed_kernel(const int32_t* __restrict__ ids, int ids_stride, const uint16_t* __restrict__ table,
                             uint16_t* __restrict__ out, int T, int hidden, int vocab) {
    const int t = blockIdx.x;
    const int id = ids[(size_t)t * ids_stride];
    const bool ok = id >= 0 && id < vocab;
    for (int v = threadIdx.x; v < hidden / 8; v += blockDim.x) {
        uint4 a = ok ? *reinterpret_cast<const uint4*>(table + (size_t)id * hidden + v * 8) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(out + (size_t)t * hidden + v * 8) = a;
    }
}

// rows -> fragment-major residual stream + slab 0 of the sum-of-squares partials (ids == NULL: row t of `table`)
__global__ __launch_bounds__(128) void gather_frag_kernel(const int32_t* __restrict__ ids, int ids_stride,
                                                          const uint16_t* __restrict__ table, uint16_t* __restrict__ r_out,
                                                          float* __restrict__ part_out, int hidden, int vocab,
                                                          int zero_slabs, int pstride) {
    const int t = blockIdx.x;
    const int id = ids ? ids[(size_t)t * ids_stride] : t;
    const bool ok = !ids || (id >= 0 && id < vocab);
    float ss = 0.f;
    for (int v = threadIdx.x; v < hidden / 8; v += 128) {
        const uint4 a = ok ? *reinterpret_cast<const uint4*>(table + (size_t)id * hidden + v * 8) : make_uint4(0, 0, 0, 0);
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&a);
#pragma unroll
        for (int j = 0; j < 4; ++j) ss += bf_lo(w[j]) * bf_lo(w[j]) + bf_hi(w[j]) * bf_hi(w[j]);
        *reinterpret_cast<uint4*>(r_out + frag_off(t, v * 8, hidden)) = a;
    }
    __shared__ float red[2];
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) part_out[t] = red[0] + red[1];
    for (int p = 1 + threadIdx.x; p < zero_slabs; p += 128) part_out[(size_t)p * pstride + t] = 0.f;
}

int k_gather_frag(const int32_t* ids, int ids_stride, const void* table, void* r_out, float* part_out, int T, int hidden,
                  int vocab, void* stream, int zero_slabs, int pstride) {
    OMNI_CHECK_ARG(table && r_out && part_out, "gather_frag: null pointer");
    OMNI_CHECK_ARG(hidden % 32 == 0, "gather_frag: hidden=%d not a multiple of 32", hidden);
    if (T <= 0) return OMNI_OK;
    hipLaunchKernelGGL(gather_frag_kernel, dim3(T), dim3(128), 0, (hipStream_t)stream, ids, ids_stride, (const uint16_t*)table,
                       (uint16_t*)r_out, part_out, hidden, vocab, zero_slabs, pstride);
    OMNI_CHECK_LAUNCH("gather_frag");
    return OMNI_OK;
}

int k_embed(const int32_t* ids, int ids_stride, const void* table, void* out, int T, int hidden, int vocab,
            void* stream) {
    OMNI_CHECK_ARG(ids && table && out, "omni_embed: null pointer");
    OMNI_CHECK_ARG(hidden % 8 == 0, "omni_embed: hidden=%d not a multiple of 8", hidden);
    if (T <= 0) return OMNI_OK;
    hipLaunchKernelGGL(embed_kernel, dim3(T), dim3(128), 0, (hipStream_t)stream, ids, ids_stride, (const uint16_t*)table,
                       (uint16_t*)out, T, hidden, vocab);
    OMNI_CHECK_LAUNCH("omni_embed");
    return OMNI_OK;
}

extern "C" int omni_embed(const int32_t* ids, const void* table, void* out, int T, int hidden, int vocab,
                          void* stream) {
    return k_embed(ids, 1, table, out, T, hidden, vocab, stream);
}

// act[t, i] = bf16(bf16(silu(gate[t, i])) * up[t, i]) for gate_up = [gate | up] rows of 2 * inter (the prefill path's
// hipBLASLt gate_up GEMM has no fused activation); same roundings as the skinny GEMM's SiLU epilogue
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ gu, uint16_t* __restrict__ out, size_t nvec, int inter8) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    const size_t t = v / inter8;
    const int c = (int)(v - t * inter8);
    const uint4 g = *reinterpret_cast<const uint4*>(gu + (t * 2 * inter8 + c) * 8);
    const uint4 u = *reinterpret_cast<const uint4*>(gu + (t * 2 * inter8 + inter8 + c) * 8);
    const uint32_t* gw = reinterpret_cast<const uint32_t*>(&g);
    const uint32_t* uw = reinterpret_cast<const uint32_t*>(&u);
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float g0 = bf_lo(gw[j]), g1 = bf_hi(gw[j]);
        const float s0 = bfround(g0 / (1.0f + expf(-g0))), s1 = bfround(g1 / (1.0f + expf(-g1)));
        o[j] = pack_bf2(s0 * bf_lo(uw[j]), s1 * bf_hi(uw[j]));
    }
    *reinterpret_cast<uint4*>(out + v * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

extern "C" int omni_silu_mul(const void* gate_up, void* out, int T, int inter, void* stream) {
    OMNI_CHECK_ARG(gate_up && out, "omni_silu_mul: null pointer");
    OMNI_CHECK_ARG(inter > 0 && inter % 8 == 0, "omni_silu_mul: inter=%d not a multiple of 8", inter);
    if (T <= 0) return OMNI_OK;
    const size_t nvec = (size_t)T * (inter / 8);
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)gate_up, (uint16_t*)out, nvec, inter / 8);
    OMNI_CHECK_LAUNCH("omni_silu_mul");
    return OMNI_OK;
}

// y = bf16(silu(x)) elementwise (fp32 math, one rounding): the activation between linear_fc1 and linear_fc2 of the Omni talker's
// thinker -> talker projections (HF Qwen3OmniMoeTalkerResizeMLP; reference use sites qwen3_omni.py:671,987-997)
__global__ __launch_bounds__(256) void silu_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, size_t n) {
    const size_t v = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (v >= n) return;
    if (v + 8 <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(x + v);
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&a);
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g0 = bf_lo(w[j]), g1 = bf_hi(w[j]);
            o[j] = pack_bf2(g0 / (1.0f + expf(-g0)), g1 / (1.0f + expf(-g1)));
        }
        *reinterpret_cast<uint4*>(out + v) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
        for (size_t i = v; i < n; ++i) { const float g = bf2f(x[i]); out[i] = f2bf(g / (1.0f + expf(-g))); }
    }
}

extern "C" int omni_silu(const void* x, void* out, long long n, void* stream) {
    OMNI_CHECK_ARG(x && out, "omni_silu: null pointer");
    OMNI_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "omni_silu: buffers must be 16-byte aligned");
    if (n <= 0) return OMNI_OK;
    const size_t nv = ((size_t)n + 7) / 8;
    hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (uint16_t*)out, (size_t)n);
    OMNI_CHECK_LAUNCH("omni_silu");
    return OMNI_OK;
}

// SnakeBeta activation of the Code2Wav decoder: out = x + inv_beta[c] * sin^2(x * exp_alpha[c]) over
// [B, C, T] rows of length T, one (b, c) row per blockIdx.y; exp_alpha = exp(alpha), inv_beta = 1 / (exp(beta) + 1e-9) are
// precomputed per channel as the reference's precompute_exp_cache does.  Replaces the reference's Triton kernel
// (tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:632-655, same fused form).  fp32 math; T-contiguous 16-B accesses.
template <bool IS_BF16>
__global__ __launch_bounds__(256) void snake_beta_kernel(const void* __restrict__ x, const float* __restrict__ exp_alpha,
                                                         const float* __restrict__ inv_beta, void* __restrict__ out, int C, int T) {
    const size_t rowi = blockIdx.x;          // grid.x = B * C rows (up to 2^31 - 1), grid.y = chunks of T
    const int c = (int)(rowi % C);
    const float ea = exp_alpha[c], ib = inv_beta[c];
    constexpr int VEC = IS_BF16 ? 8 : 4;
    const int t0 = (blockIdx.y * 256 + threadIdx.x) * VEC;
    if (t0 >= T) return;
    if constexpr (VEC == 4) {        // fp32
        const float* xr = reinterpret_cast<const float*>(x) + rowi * T;
        float* orow = reinterpret_cast<float*>(out) + rowi * T;
        if (t0 + 4 <= T && ((reinterpret_cast<uintptr_t>(xr + t0) | reinterpret_cast<uintptr_t>(orow + t0)) & 15) == 0) {
            float4 v = *reinterpret_cast<const float4*>(xr + t0);
            float s;
            s = sinf(v.x * ea); v.x += ib * s * s;
            s = sinf(v.y * ea); v.y += ib * s * s;
            s = sinf(v.z * ea); v.z += ib * s * s;
            s = sinf(v.w * ea); v.w += ib * s * s;
            *reinterpret_cast<float4*>(orow + t0) = v;
        } else {
            for (int t = t0; t < min(t0 + 4, T); ++t) { const float v = xr[t], s = sinf(v * ea); orow[t] = v + ib * s * s; }
        }
    } else {                         // bf16 storage, fp32 math, one rounding
        const uint16_t* xr = reinterpret_cast<const uint16_t*>(x) + rowi * T;
        uint16_t* orow = reinterpret_cast<uint16_t*>(out) + rowi * T;
        for (int t = t0; t < min(t0 + 8, T); ++t) { const float v = bf2f(xr[t]), s = sinf(v * ea); orow[t] = f2bf(v + ib * s * s); }
    }
}

extern "C" int omni_snake_beta(const void* x, const float* exp_alpha, const float* inv_beta, void* out, int B, int C, int T,
                               int is_bf16, void* stream) {
    OMNI_CHECK_ARG(x && exp_alpha && inv_beta && out, "omni_snake_beta: null pointer");
    OMNI_CHECK_ARG(B >= 0 && C > 0 && T >= 0, "omni_snake_beta: bad shape");
    if (B == 0 || T == 0) return OMNI_OK;
    const int vec = is_bf16 ? 8 : 4;
    const long long chunks = ((long long)T + 256 * vec - 1) / (256 * vec);
    OMNI_CHECK_ARG((long long)B * C <= 2147483647LL && chunks <= 65535, "omni_snake_beta: B * C = %lld rows, %lld chunks of T: too large",
                   (long long)B * C, chunks);
    dim3 grid((unsigned)(B * C), (unsigned)chunks);
    if (is_bf16) hipLaunchKernelGGL(snake_beta_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, exp_alpha, inv_beta, out, C, T);
    else hipLaunchKernelGGL(snake_beta_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, exp_alpha, inv_beta, out, C, T);
    OMNI_CHECK_LAUNCH("omni_snake_beta");
    return OMNI_OK;
}
