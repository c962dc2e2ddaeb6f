// Code2Wav (12 Hz speech-tokenizer decoder) kernels that are not GEMM-shaped.  Every convolution / linear layer of the decoder
// runs on omni_gemm_tile (gemm_prefill.hip) over TIME-major activations [T, C]; this file holds the rest:
//   codec_rvq_embed    residual-VQ code lookup: sum over the quantizers of one row of a folded table (codebook / usage, times the
//                      1x1 output projection, folded at load) -> bf16 [T, C]
//   codec_rmsnorm      RMSNorm of the fp32 residual stream -> bf16 GEMM operand
//   codec_rope         rotate-half RoPE on the q / k heads of a fused qkv row, in place
//   codec_window_attn  causal sliding-window attention, one wave per (frame, head): <= window keys
//   codec_dwconv_ln    ConvNeXt front: depthwise causal conv (7 taps) + LayerNorm over channels, fp32 stream -> bf16
//   codec_out_conv     last layer: causal conv to ONE channel + clamp to [-1, 1], bf16 -> fp32 waveform
// All HBM-bound element / row work: coalesced 16-byte accesses along the channel dimension, reductions by DPP / permlane
// (common.cuh), no LDS except where a row is re-read.
// Reference: tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py (Qwen3TTSTokenizerV2Decoder.forward :1009-1027 and the modules
// it calls); oracle: oracle/code2wav_oracle.py.
#include "common.cuh"
#include "kernels.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- residual VQ lookup
// out[t, :] = bf16( sum_q table[q][clamp(codes[q, t])][:] ), table fp32 [Q][bins][C]
__global__ __launch_bounds__(256) void codec_rvq_embed_kernel(const int64_t* __restrict__ codes, int ld_codes, const float* __restrict__ table,
                                                              uint16_t* __restrict__ out, int T, int Q, int bins, int C) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < Q; ++q) {
            long long id = codes[(size_t)q * ld_codes + t];
            id = id < 0 ? 0 : (id >= bins ? bins - 1 : id);            // memory safety only: the caller validates the range
            acc += *reinterpret_cast<const f32x4*>(table + ((size_t)q * bins + id) * C + c);
        }
        uint2 pk;
        pk.x = pack_bf2(acc[0], acc[1]);
        pk.y = pack_bf2(acc[2], acc[3]);
        *reinterpret_cast<uint2*>(out + (size_t)t * C + c) = pk;
    }
}

extern "C" int omni_codec_rvq_embed(const int64_t* codes, int ld_codes, const float* table, void* out, int T, int Q, int bins, int C,
                                    void* stream) {
    OMNI_CHECK_ARG(codes && table && out && T > 0 && Q > 0 && bins > 0 && C > 0 && C % 4 == 0, "omni_codec_rvq_embed: T=%d Q=%d bins=%d C=%d", T, Q, bins, C);
    hipLaunchKernelGGL(codec_rvq_embed_kernel, dim3(T), dim3(C >= 1024 ? 256 : 64), 0, (hipStream_t)stream, codes, ld_codes, table,
                       (uint16_t*)out, T, Q, bins, C);
    OMNI_CHECK_LAUNCH("omni_codec_rvq_embed");
    return OMNI_OK;
}

// ---------------------------------------------------------------- RMSNorm of the fp32 stream: one wave per row
__global__ __launch_bounds__(256) void codec_rmsnorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, float eps,
                                                            uint16_t* __restrict__ out, int T, int H) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const float* xr = x + (size_t)t * ldx;
    float ss = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(w + c);
        uint2 pk;
        pk.x = pack_bf2(g[0] * (v[0] * rstd), g[1] * (v[1] * rstd));
        pk.y = pack_bf2(g[2] * (v[2] * rstd), g[3] * (v[3] * rstd));
        *reinterpret_cast<uint2*>(out + (size_t)t * H + c) = pk;
    }
}

extern "C" int omni_codec_rmsnorm(const float* x, int ldx, const float* w, float eps, void* out, int T, int H, void* stream) {
    OMNI_CHECK_ARG(x && w && out && T > 0 && H > 0 && H % 4 == 0 && ldx >= H && ldx % 4 == 0, "omni_codec_rmsnorm: T=%d H=%d ldx=%d", T, H, ldx);
    hipLaunchKernelGGL(codec_rmsnorm_kernel, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, w, eps, (uint16_t*)out, T, H);
    OMNI_CHECK_LAUNCH("omni_codec_rmsnorm");
    return OMNI_OK;
}

// ---------------------------------------------------------------- RoPE (rotate-half) on the q and k heads of qkv rows, in place
// position = row index; angle(d) = t * theta^(-2 d / hd), d < hd / 2; (x1, x2) = (x[d], x[d + hd/2]) -> (x1 cos - x2 sin, x2 cos + x1 sin)
__global__ __launch_bounds__(256) void codec_rope_kernel(uint16_t* __restrict__ qkv, int ld, int T, int heads /* q + k heads */, int hd,
                                                         float log2_theta) {
    const int half = hd >> 1;
    const int per_row = heads * half;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)T * per_row) return;
    const int t = (int)(idx / per_row), r = (int)(idx - (long long)t * per_row);
    const int h = r / half, d = r - h * half;
    const float inv = exp2f(-log2_theta * (float)(2 * d) / (float)hd);
    float sn, cs;
    sincosf((float)t * inv, &sn, &cs);
    uint16_t* p = qkv + (size_t)t * ld + h * hd + d;
    const float x1 = bf2f(p[0]), x2 = bf2f(p[half]);
    p[0] = f2bf(x1 * cs - x2 * sn);
    p[half] = f2bf(x2 * cs + x1 * sn);
}

extern "C" int omni_codec_rope(void* qkv, int ld, int T, int q_heads, int kv_heads, int head_dim, float theta, void* stream) {
    OMNI_CHECK_ARG(qkv && T > 0 && q_heads > 0 && kv_heads > 0 && head_dim % 2 == 0 && theta > 0.f, "omni_codec_rope: bad shape");
    const long long n = (long long)T * (q_heads + kv_heads) * (head_dim / 2);
    hipLaunchKernelGGL(codec_rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)qkv, ld, T,
                       q_heads + kv_heads, head_dim, log2f(theta));
    OMNI_CHECK_LAUNCH("omni_codec_rope");
    return OMNI_OK;
}

// ---------------------------------------------------------------- causal sliding-window attention, one wave per (frame, head)
// qkv row = [q heads | k heads | v heads] x hd, already rotated.  Keys j in [max(0, t - window + 1), t], 64 per chunk: lane = key
// for the scores (q broadcast by v_readlane), online softmax across chunks, lane = output dims for P.V (p broadcast by
// v_readlane).  VPL = hd / 64 values per lane.
template <int VPL>
__global__ __launch_bounds__(64) void codec_window_attn_kernel(const uint16_t* __restrict__ qkv, int ld, uint16_t* __restrict__ out, int ldo,
                                                               int nh, int nkv, int window, float scale) {
    constexpr int HD = 64 * VPL;
    const int t = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int kvh = h / (nh / nkv);
    const uint16_t* qp = qkv + (size_t)t * ld + h * HD;
    const size_t koff = (size_t)(nh + kvh) * HD, voff = (size_t)(nh + nkv + kvh) * HD;
    int qv[VPL];                                          // q[lane + 64 v] * scale as raw bits (v_readlane works on b32)
#pragma unroll
    for (int v = 0; v < VPL; ++v) qv[v] = __float_as_int(bf2f(qp[lane + 64 * v]) * scale);
    const int j0 = max(0, t - window + 1);
    float m = -INFINITY, l = 0.f, acc[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = 0.f;
    for (int jb = j0; jb <= t; jb += 64) {
        const int cnt = min(64, t - jb + 1);
        const int j = min(jb + lane, t);                  // lanes past the window recompute key t and are masked below
        const uint16_t* kp = qkv + (size_t)j * ld + koff;
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int d8 = 0; d8 < 8; ++d8) {
                const u32x4 kk = *reinterpret_cast<const u32x4*>(kp + 64 * v + d8 * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float q0 = __int_as_float(__builtin_amdgcn_readlane(qv[v], d8 * 8 + 2 * e));
                    const float q1 = __int_as_float(__builtin_amdgcn_readlane(qv[v], d8 * 8 + 2 * e + 1));
                    dot += q0 * bf_lo(kk[e]) + q1 * bf_hi(kk[e]);
                }
            }
        const float s = lane < cnt ? dot : -INFINITY;
        const float m_new = fmaxf(m, wave_max(s));
        const float corr = __expf(m - m_new);             // first chunk: exp(-inf) = 0
        const float p = lane < cnt ? __expf(s - m_new) : 0.f;
        l = l * corr + wave_sum(p);
        m = m_new;
        const int pbits = __float_as_int(p);
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[v] *= corr;
        for (int jj = 0; jj < cnt; ++jj) {
            const float pj = __int_as_float(__builtin_amdgcn_readlane(pbits, jj));
            const uint16_t* vp = qkv + (size_t)(jb + jj) * ld + voff + lane;
#pragma unroll
            for (int v = 0; v < VPL; ++v) acc[v] += pj * bf2f(vp[64 * v]);
        }
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int v = 0; v < VPL; ++v) out[(size_t)t * ldo + h * HD + lane + 64 * v] = f2bf(acc[v] * inv);
}

extern "C" int omni_codec_window_attn(const void* qkv, int ld, void* out, int ldo, int T, int q_heads, int kv_heads, int head_dim,
                                      int window, float scale, void* stream) {
    OMNI_CHECK_ARG(qkv && out && T > 0 && q_heads > 0 && kv_heads > 0 && q_heads % kv_heads == 0 && window > 0, "omni_codec_window_attn: bad shape");
    OMNI_CHECK_ARG(head_dim == 64 || head_dim == 128, "omni_codec_window_attn: head_dim=%d (64 | 128)", head_dim);
    OMNI_CHECK_ARG(ld % 8 == 0 && ld >= (q_heads + 2 * kv_heads) * head_dim, "omni_codec_window_attn: ld=%d", ld);
    const dim3 grid(T, q_heads);
    if (head_dim == 64)
        hipLaunchKernelGGL(codec_window_attn_kernel<1>, grid, dim3(64), 0, (hipStream_t)stream, (const uint16_t*)qkv, ld, (uint16_t*)out, ldo,
                           q_heads, kv_heads, window, scale);
    else
        hipLaunchKernelGGL(codec_window_attn_kernel<2>, grid, dim3(64), 0, (hipStream_t)stream, (const uint16_t*)qkv, ld, (uint16_t*)out, ldo,
                           q_heads, kv_heads, window, scale);
    OMNI_CHECK_LAUNCH("omni_codec_window_attn");
    return OMNI_OK;
}

// ---------------------------------------------------------------- ConvNeXt front: depthwise causal conv + LayerNorm over channels
// y[t, c] = b[c] + sum_j w[c][j] x[t - (taps - 1) + j][c] (rows < 0 are zero);  out = bf16(LN(y) * g + beta).  One workgroup per frame.
__global__ __launch_bounds__(256) void codec_dwconv_ln_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                              const float* __restrict__ b, const float* __restrict__ ln_w,
                                                              const float* __restrict__ ln_b, float eps, uint16_t* __restrict__ out,
                                                              int T, int C, int taps) {
    __shared__ float red[2][4];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float y[4];                                           // C <= 1024: channel tid + 256 i
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        y[i] = 0.f;
        if (c < C) {
            float a = b[c];
            for (int j = 0; j < taps; ++j) {
                const int r = t - (taps - 1) + j;
                if (r >= 0) a += w[c * taps + j] * x[(size_t)r * ldx + c];
            }
            y[i] = a;
            sum += a;
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) red[0][wv] = sum;
    __syncthreads();
    const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (tid + 256 * i < C) var += (y[i] - mean) * (y[i] - mean);
    var = wave_sum(var);
    if (lane == 0) red[1][wv] = var;
    __syncthreads();
    const float rstd = rsqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        if (c < C) out[(size_t)t * C + c] = f2bf((y[i] - mean) * rstd * ln_w[c] + ln_b[c]);
    }
}

extern "C" int omni_codec_dwconv_ln(const float* x, int ldx, const float* w, const float* b, const float* ln_w, const float* ln_b, float eps,
                                    void* out, int T, int C, int taps, void* stream) {
    OMNI_CHECK_ARG(x && w && b && ln_w && ln_b && out && T > 0 && C > 0 && C <= 1024 && taps > 0 && ldx >= C, "omni_codec_dwconv_ln: T=%d C=%d (<= 1024)", T, C);
    hipLaunchKernelGGL(codec_dwconv_ln_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, x, ldx, w, b, ln_w, ln_b, eps, (uint16_t*)out, T, C, taps);
    OMNI_CHECK_LAUNCH("omni_codec_dwconv_ln");
    return OMNI_OK;
}

// ---------------------------------------------------------------- last layer: causal conv to one channel + clamp
// wav[t] = clamp(b + sum_{j, c} w[j][c] x[t - (taps - 1) + j][c], -1, 1); x bf16 [T, C] (already through the last snake), w fp32
// [taps][C] in LDS.  A wave produces 64 consecutive samples: lane = sample; rows are shared by neighbouring lanes through L1.
__global__ __launch_bounds__(256) void codec_out_conv_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w, float bias,
                                                             float* __restrict__ wav, int T, int C, int taps) {
    extern __shared__ float wl[];                         // [taps * C]
    for (int i = threadIdx.x; i < taps * C; i += blockDim.x) wl[i] = w[i];
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float acc = bias;
    for (int j = 0; j < taps; ++j) {
        const int r = t - (taps - 1) + j;
        if (r < 0) continue;
        const uint16_t* xr = x + (size_t)r * C;
        const float* wr = wl + j * C;
        for (int c = 0; c < C; c += 8) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += bf_lo(v[e]) * wr[c + 2 * e] + bf_hi(v[e]) * wr[c + 2 * e + 1];
        }
    }
    wav[t] = fminf(fmaxf(acc, -1.0f), 1.0f);
}

extern "C" int omni_codec_out_conv(const void* x, const float* w, float bias, float* wav, int T, int C, int taps, void* stream) {
    OMNI_CHECK_ARG(x && w && wav && T > 0 && C > 0 && C % 8 == 0 && taps > 0 && taps * C * 4 <= 64 * 1024, "omni_codec_out_conv: T=%d C=%d taps=%d", T, C, taps);
    hipLaunchKernelGGL(codec_out_conv_kernel, dim3((T + 255) / 256), dim3(256), taps * C * sizeof(float), (hipStream_t)stream,
                       (const uint16_t*)x, w, bias, wav, T, C, taps);
    OMNI_CHECK_LAUNCH("omni_codec_out_conv");
    return OMNI_OK;
}
