// Per-head q/k RMSNorm + neox RoPE + KV-cache write with bf16 / fp8-e4m3fn / int8 quantisation,
// and the device-side slot mapping.  One wave per (token, head slot); head_dim = 128 so lane l
// owns elements l and l+64 -- exactly the rotate_half pair, no cross-lane traffic for RoPE.
// Numerics = oracle (talker_oracle.rms_norm / apply_rope / fp8_quant / int8_quant).
#include "common.cuh"

template <int KV>
__global__ __launch_bounds__(256) void qknorm_rope_kvwrite_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ qnorm_w, const uint16_t* __restrict__ knorm_w,
    const int32_t* __restrict__ positions, const uint16_t* __restrict__ cos_sin, const int64_t* __restrict__ slot_mapping,
    uint16_t* __restrict__ q_out, void* __restrict__ k_cache, void* __restrict__ v_cache,
    float* __restrict__ k_scales, float* __restrict__ v_scales, int q_heads, int kv_heads, float eps, float inv_k_scale,
    float inv_v_scale, float k_scale, float v_scale) {
    const int lane = threadIdx.x & 63;
    const int slot_h = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nslots = q_heads + 2 * kv_heads;
    if (slot_h >= nslots) return;
    const int t = blockIdx.y;
    const uint16_t* src = qkv + ((size_t)t * nslots + slot_h) * 128;
    float x0 = bf2f(src[lane]), x1 = bf2f(src[lane + 64]);
    const bool is_q = slot_h < q_heads;
    const bool is_v = slot_h >= q_heads + kv_heads;
    if (!is_v) {
        const uint16_t* nw = is_q ? qnorm_w : knorm_w;
        const float ss = wave_sum(x0 * x0 + x1 * x1);
        const float rstd = 1.0f / sqrtf(ss * (1.0f / 128.0f) + eps);
        const float n0 = bfround(bf2f(nw[lane]) * bfround(x0 * rstd));
        const float n1 = bfround(bf2f(nw[lane + 64]) * bfround(x1 * rstd));
        const int pos = positions[t];
        const float c = bf2f(cos_sin[(size_t)pos * 128 + lane]);
        const float s = bf2f(cos_sin[(size_t)pos * 128 + 64 + lane]);
        x0 = bfround(bfround(n0 * c) + bfround(-n1 * s));
        x1 = bfround(bfround(n1 * c) + bfround(n0 * s));
    }
    if (is_q) {
        uint16_t* dst = q_out + ((size_t)t * q_heads + slot_h) * 128;
        dst[lane] = f2bf(x0);
        dst[lane + 64] = f2bf(x1);
        return;
    }
    const int64_t slot = slot_mapping[t];
    if (slot < 0) return;   // padded token
    const int kvh = is_v ? slot_h - q_heads - kv_heads : slot_h - q_heads;
    const size_t row = (size_t)slot * kv_heads + kvh;
    void* cache = is_v ? v_cache : k_cache;
    if (KV == OMNI_KV_BF16) {
        uint16_t* dst = reinterpret_cast<uint16_t*>(cache) + row * 128;
        dst[lane] = f2bf(x0);
        dst[lane + 64] = f2bf(x1);
    } else if (KV == OMNI_KV_FP8) {
        const float inv = is_v ? inv_v_scale : inv_k_scale;
        const float sc = is_v ? v_scale : k_scale;
        // x / scale (correctly rounded divide unless scale == 1)
        const float y0 = (inv == 1.0f) ? x0 : x0 / sc;
        const float y1 = (inv == 1.0f) ? x1 : x1 / sc;
        uint8_t* dst = reinterpret_cast<uint8_t*>(cache) + row * 128;
        dst[lane] = (uint8_t)(pack_fp8x4(y0, 0.f, 0.f, 0.f) & 0xFF);
        dst[lane + 64] = (uint8_t)(pack_fp8x4(y1, 0.f, 0.f, 0.f) & 0xFF);
    } else {
        const float amax = fmaxf(wave_max(fmaxf(fabsf(x0), fabsf(x1))), 1e-8f);
        const float sc = amax / 127.0f;
        const float y0 = fminf(fmaxf(rintf(x0 / sc), -127.f), 127.f);
        const float y1 = fminf(fmaxf(rintf(x1 / sc), -127.f), 127.f);
        int8_t* dst = reinterpret_cast<int8_t*>(cache) + row * 128;
        dst[lane] = (int8_t)y0;
        dst[lane + 64] = (int8_t)y1;
        if (lane == 0) (is_v ? v_scales : k_scales)[row] = sc;
    }
}

extern "C" int omni_qknorm_rope_kvwrite(const void* qkv, const void* qnorm_w, const void* knorm_w,
                                        const int32_t* positions, const void* cos_sin, const int64_t* slot_mapping,
                                        void* q_out, void* k_cache, void* v_cache, float* k_scales, float* v_scales,
                                        int T, int q_heads, int kv_heads, int head_dim, float eps, int kv_dtype,
                                        float k_scale, float v_scale, void* stream) {
    OMNI_CHECK_ARG(qkv && qnorm_w && knorm_w && positions && cos_sin && slot_mapping && q_out && k_cache && v_cache,
                   "omni_qknorm_rope_kvwrite: null pointer");
    OMNI_CHECK_ARG(head_dim == 128, "omni_qknorm_rope_kvwrite: head_dim=%d (only 128)", head_dim);
    OMNI_CHECK_ARG(kv_dtype != OMNI_KV_INT8 || (k_scales && v_scales), "omni_qknorm_rope_kvwrite: int8 needs scale arrays");
    OMNI_CHECK_ARG(k_scale > 0.f && v_scale > 0.f, "omni_qknorm_rope_kvwrite: scales must be > 0");
    if (T <= 0) return OMNI_OK;
    const int nslots = q_heads + 2 * kv_heads;
    dim3 grid((nslots + 3) / 4, T), block(256);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(KVT)                                                                                              \
    hipLaunchKernelGGL(qknorm_rope_kvwrite_kernel<KVT>, grid, block, 0, st, (const uint16_t*)qkv,               \
                       (const uint16_t*)qnorm_w, (const uint16_t*)knorm_w, positions, (const uint16_t*)cos_sin, \
                       slot_mapping, (uint16_t*)q_out, k_cache, v_cache, k_scales, v_scales, q_heads, kv_heads,  \
                       eps, 1.0f / k_scale, 1.0f / v_scale, k_scale, v_scale)
    switch (kv_dtype) {
        case OMNI_KV_BF16: LAUNCH(OMNI_KV_BF16); break;
        case OMNI_KV_FP8: LAUNCH(OMNI_KV_FP8); break;
        case OMNI_KV_INT8: LAUNCH(OMNI_KV_INT8); break;
        default: omni_set_error("omni_qknorm_rope_kvwrite: kv_dtype=%d", kv_dtype); return OMNI_EINVAL;
    }
#undef LAUNCH
    OMNI_CHECK_LAUNCH("omni_qknorm_rope_kvwrite");
    return OMNI_OK;
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
