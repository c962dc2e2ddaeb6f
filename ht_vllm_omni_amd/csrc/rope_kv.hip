// Per-head q/k RMSNorm + neox RoPE + KV-cache write with bf16 / fp8-e4m3fn / int8 quantisation,
// and the device-side slot mapping.  16 lanes per (token, head slot); head_dim = 128 and a lane owns
// elements e and e+64 -- exactly the rotate_half pairs, no cross-lane traffic for RoPE.
// Numerics = oracle (talker_oracle.rms_norm / apply_rope / fp8_quant / int8_quant).
#include "common.cuh"

// 16 lanes per (token, head slot): lane j owns elements [4j, 4j+4) and [64+4j, 64+4j+4) -- the rotate_half pairs stay
// in one lane and every access is 8 B (the prefill calls this for thousands of tokens per layer)
// (a group is one 16-lane DPP row and leaves the kernel as a whole, so the row-local DPP reductions see no disabled lane)
__device__ __forceinline__ float sum16(float v) { return row16_sum(v); }
__device__ __forceinline__ float max16(float v) { return row16_max(v); }
__device__ __forceinline__ void unpack4(uint2 w, float* f) { f[0] = bf_lo(w.x); f[1] = bf_hi(w.x); f[2] = bf_lo(w.y); f[3] = bf_hi(w.y); }

// MROPE: `positions` is [3, T] (temporal / height / width ids, vLLM MRotaryEmbedding) and rotary pair p of a head takes the id of
// axis mrope_axis[p] (the section layout, chunked or interleaved, as a 64-entry table): cos / sin are gathered per pair from the
// row of THAT id -- with three identical rows this is the plain kernel bit for bit.
template <int KV, bool MROPE = false>
__global__ __launch_bounds__(256) void qknorm_rope_kvwrite_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ qnorm_w, const uint16_t* __restrict__ knorm_w,
    const int32_t* __restrict__ positions, const uint8_t* __restrict__ mrope_axis, int T,
    const uint16_t* __restrict__ cos_sin, const int64_t* __restrict__ slot_mapping,
    uint16_t* __restrict__ q_out, void* __restrict__ k_cache, void* __restrict__ v_cache,
    float* __restrict__ k_scales, float* __restrict__ v_scales, int q_heads, int kv_heads, float eps, float inv_k_scale,
    float inv_v_scale, float k_scale, float v_scale) {
    const int j = threadIdx.x & 15;
    const int slot_h = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int nslots = q_heads + 2 * kv_heads;
    if (slot_h >= nslots) return;                     // whole 16-lane groups leave together (shuffles stay in-group)
    const int t = blockIdx.y;
    const uint16_t* src = qkv + ((size_t)t * nslots + slot_h) * 128;
    float x0[4], x1[4];
    unpack4(*reinterpret_cast<const uint2*>(src + 4 * j), x0);
    unpack4(*reinterpret_cast<const uint2*>(src + 64 + 4 * j), x1);
    const bool is_q = slot_h < q_heads;
    const bool is_v = slot_h >= q_heads + kv_heads;
    if (!is_v) {
        const uint16_t* nw = is_q ? qnorm_w : knorm_w;
        float w0[4], w1[4], c[4], sn[4];
        unpack4(*reinterpret_cast<const uint2*>(nw + 4 * j), w0);
        unpack4(*reinterpret_cast<const uint2*>(nw + 64 + 4 * j), w1);
        if (MROPE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = 4 * j + e;
                const uint16_t* row = cos_sin + (size_t)positions[(size_t)mrope_axis[p] * T + t] * 128;
                c[e] = bf2f(row[p]);
                sn[e] = bf2f(row[64 + p]);
            }
        } else {
            const int pos = positions[t];
            unpack4(*reinterpret_cast<const uint2*>(cos_sin + (size_t)pos * 128 + 4 * j), c);
            unpack4(*reinterpret_cast<const uint2*>(cos_sin + (size_t)pos * 128 + 64 + 4 * j), sn);
        }
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += x0[e] * x0[e] + x1[e] * x1[e];
        ss = sum16(ss);
        const float rstd = 1.0f / sqrtf(ss * (1.0f / 128.0f) + eps);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float n0 = bfround(w0[e] * bfround(x0[e] * rstd));
            const float n1 = bfround(w1[e] * bfround(x1[e] * rstd));
            x0[e] = bfround(bfround(n0 * c[e]) + bfround(-n1 * sn[e]));
            x1[e] = bfround(bfround(n1 * c[e]) + bfround(n0 * sn[e]));
        }
    }
    if (is_q) {
        uint16_t* dst = q_out + ((size_t)t * q_heads + slot_h) * 128;
        *reinterpret_cast<uint2*>(dst + 4 * j) = make_uint2(pack_bf2(x0[0], x0[1]), pack_bf2(x0[2], x0[3]));
        *reinterpret_cast<uint2*>(dst + 64 + 4 * j) = make_uint2(pack_bf2(x1[0], x1[1]), pack_bf2(x1[2], x1[3]));
        return;
    }
    const int64_t slot = slot_mapping[t];
    if (slot < 0) return;   // padded token
    const int kvh = is_v ? slot_h - q_heads - kv_heads : slot_h - q_heads;
    const size_t row = (size_t)slot * kv_heads + kvh;
    void* cache = is_v ? v_cache : k_cache;
    if (KV == OMNI_KV_BF16) {
        uint16_t* dst = reinterpret_cast<uint16_t*>(cache) + row * 128;
        *reinterpret_cast<uint2*>(dst + 4 * j) = make_uint2(pack_bf2(x0[0], x0[1]), pack_bf2(x0[2], x0[3]));
        *reinterpret_cast<uint2*>(dst + 64 + 4 * j) = make_uint2(pack_bf2(x1[0], x1[1]), pack_bf2(x1[2], x1[3]));
    } else if (KV == OMNI_KV_FP16) {
        // the model's bf16 K / V cast to half (RNE; exact inside the half range)
        uint16_t* dst = reinterpret_cast<uint16_t*>(cache) + row * 128;
        uint32_t h0[2], h1[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            h0[e] = (uint32_t)f2h(bfround(x0[2 * e])) | ((uint32_t)f2h(bfround(x0[2 * e + 1])) << 16);
            h1[e] = (uint32_t)f2h(bfround(x1[2 * e])) | ((uint32_t)f2h(bfround(x1[2 * e + 1])) << 16);
        }
        *reinterpret_cast<uint2*>(dst + 4 * j) = make_uint2(h0[0], h0[1]);
        *reinterpret_cast<uint2*>(dst + 64 + 4 * j) = make_uint2(h1[0], h1[1]);
    } else if (KV == OMNI_KV_FP8) {
        const float inv = is_v ? inv_v_scale : inv_k_scale;
        const float sc = is_v ? v_scale : k_scale;
        if (inv != 1.0f) {            // x / scale: correctly rounded divide
#pragma unroll
            for (int e = 0; e < 4; ++e) { x0[e] = x0[e] / sc; x1[e] = x1[e] / sc; }
        }
        uint8_t* dst = reinterpret_cast<uint8_t*>(cache) + row * 128;
        *reinterpret_cast<uint32_t*>(dst + 4 * j) = pack_fp8x4(x0[0], x0[1], x0[2], x0[3]);
        *reinterpret_cast<uint32_t*>(dst + 64 + 4 * j) = pack_fp8x4(x1[0], x1[1], x1[2], x1[3]);
    } else {
        float am = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) am = fmaxf(am, fmaxf(fabsf(x0[e]), fabsf(x1[e])));
        const float amax = fmaxf(max16(am), 1e-8f);
        const float sc = amax / 127.0f;
        uint32_t p0 = 0, p1 = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int y0 = (int)fminf(fmaxf(rintf(x0[e] / sc), -127.f), 127.f);
            const int y1 = (int)fminf(fmaxf(rintf(x1[e] / sc), -127.f), 127.f);
            p0 |= ((uint32_t)(y0 & 0xFF)) << (8 * e);
            p1 |= ((uint32_t)(y1 & 0xFF)) << (8 * e);
        }
        int8_t* dst = reinterpret_cast<int8_t*>(cache) + row * 128;
        *reinterpret_cast<uint32_t*>(dst + 4 * j) = p0;
        *reinterpret_cast<uint32_t*>(dst + 64 + 4 * j) = p1;
        if (j == 0) (is_v ? v_scales : k_scales)[row] = sc;
    }
}

static int qknorm_rope_launch(const void* qkv, const void* qnorm_w, const void* knorm_w, const int32_t* positions,
                              const uint8_t* mrope_axis, const void* cos_sin, const int64_t* slot_mapping, void* q_out, void* k_cache,
                              void* v_cache, float* k_scales, float* v_scales, int T, int q_heads, int kv_heads, int head_dim, float eps,
                              int kv_dtype, float k_scale, float v_scale, void* stream, const char* who) {
    OMNI_CHECK_ARG(qkv && qnorm_w && knorm_w && positions && cos_sin && slot_mapping && q_out && k_cache && v_cache, "%s: null pointer", who);
    OMNI_CHECK_ARG(head_dim == 128, "%s: head_dim=%d (only 128)", who, head_dim);
    OMNI_CHECK_ARG(kv_dtype != OMNI_KV_INT8 || (k_scales && v_scales), "%s: int8 needs scale arrays", who);
    OMNI_CHECK_ARG(k_scale > 0.f && v_scale > 0.f, "%s: scales must be > 0", who);
    if (T <= 0) return OMNI_OK;
    const int nslots = q_heads + 2 * kv_heads;
    dim3 grid((nslots + 15) / 16, T), block(256);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(KVT, MR)                                                                                                  \
    hipLaunchKernelGGL((qknorm_rope_kvwrite_kernel<KVT, MR>), grid, block, 0, st, (const uint16_t*)qkv,                  \
                       (const uint16_t*)qnorm_w, (const uint16_t*)knorm_w, positions, mrope_axis, T, (const uint16_t*)cos_sin, \
                       slot_mapping, (uint16_t*)q_out, k_cache, v_cache, k_scales, v_scales, q_heads, kv_heads,          \
                       eps, 1.0f / k_scale, 1.0f / v_scale, k_scale, v_scale)
#define LAUNCH2(KVT) do { if (mrope_axis) LAUNCH(KVT, true); else LAUNCH(KVT, false); } while (0)
    switch (kv_dtype) {
        case OMNI_KV_BF16: LAUNCH2(OMNI_KV_BF16); break;
        case OMNI_KV_FP8: LAUNCH2(OMNI_KV_FP8); break;
        case OMNI_KV_INT8: LAUNCH2(OMNI_KV_INT8); break;
        case OMNI_KV_FP16: LAUNCH2(OMNI_KV_FP16); break;
        default: omni_set_error("%s: kv_dtype=%d", who, kv_dtype); return OMNI_EINVAL;
    }
#undef LAUNCH2
#undef LAUNCH
    OMNI_CHECK_LAUNCH(who);
    return OMNI_OK;
}

extern "C" int omni_qknorm_rope_kvwrite(const void* qkv, const void* qnorm_w, const void* knorm_w,
                                        const int32_t* positions, const void* cos_sin, const int64_t* slot_mapping,
                                        void* q_out, void* k_cache, void* v_cache, float* k_scales, float* v_scales,
                                        int T, int q_heads, int kv_heads, int head_dim, float eps, int kv_dtype,
                                        float k_scale, float v_scale, void* stream) {
    return qknorm_rope_launch(qkv, qnorm_w, knorm_w, positions, nullptr, cos_sin, slot_mapping, q_out, k_cache, v_cache, k_scales, v_scales,
                              T, q_heads, kv_heads, head_dim, eps, kv_dtype, k_scale, v_scale, stream, "omni_qknorm_rope_kvwrite");
}

extern "C" int omni_qknorm_mrope_kvwrite(const void* qkv, const void* qnorm_w, const void* knorm_w, const int32_t* positions3,
                                         const uint8_t* mrope_axis, const void* cos_sin, const int64_t* slot_mapping, void* q_out,
                                         void* k_cache, void* v_cache, float* k_scales, float* v_scales, int T, int q_heads,
                                         int kv_heads, int head_dim, float eps, int kv_dtype, float k_scale, float v_scale,
                                         void* stream) {
    OMNI_CHECK_ARG(mrope_axis != nullptr, "omni_qknorm_mrope_kvwrite: null axis table");
    return qknorm_rope_launch(qkv, qnorm_w, knorm_w, positions3, mrope_axis, cos_sin, slot_mapping, q_out, k_cache, v_cache, k_scales,
                              v_scales, T, q_heads, kv_heads, head_dim, eps, kv_dtype, k_scale, v_scale, stream, "omni_qknorm_mrope_kvwrite");
}

__global__ void slot_mapping_kernel(const int32_t* __restrict__ bt, int bt_stride, const int32_t* __restrict__ positions,
                                    int64_t* __restrict__ slots, int B, int B_padded, int bs) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B_padded) return;
    if (r >= B) { slots[r] = -1; return; }
    const int p = positions[r];
    slots[r] = (int64_t)bt[(size_t)r * bt_stride + p / bs] * bs + p % bs;
}

extern "C" int omni_slot_mapping(const int32_t* block_table, int bt_stride, const int32_t* positions,
                                 int64_t* slot_mapping, int B, int B_padded, int block_size, void* stream) {
    OMNI_CHECK_ARG(block_table && positions && slot_mapping, "omni_slot_mapping: null pointer");
    OMNI_CHECK_ARG(block_size > 0 && B >= 0 && B_padded >= B, "omni_slot_mapping: bad sizes");
    if (B_padded == 0) return OMNI_OK;
    hipLaunchKernelGGL(slot_mapping_kernel, dim3((B_padded + 63) / 64), dim3(64), 0, (hipStream_t)stream, block_table,
                       bt_stride, positions, slot_mapping, B, B_padded, block_size);
    OMNI_CHECK_LAUNCH("omni_slot_mapping");
    return OMNI_OK;
}
