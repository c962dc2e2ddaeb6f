/* C-ABI of the Code2Wav (12 Hz speech-tokenizer decoder) stage kernels -- SURVEY 8f rank 3, the stage right after the talker.
 * Same library (libomni_talker.so), same conventions as omni_talker.h: plain pointers and sizes, device memory, every launch on
 * the HIP stream passed last, OMNI_OK / OMNI_E* return codes with omni_last_error().
 *
 * The decoder these replace is torch modules in the reference:
 *   /root/reference/vllm_omni/model_executor/models/qwen3_tts/tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py
 *     Qwen3TTSTokenizerV2Decoder.forward :1009-1027 (called per chunk by chunked_decode :1029-1043 and by
 *     cuda_graph_decoder_wrapper.py:95-121 under graph replay; driven by qwen3_tts_code2wav.py:21-334).
 * Activations are TIME-major [T, C] here (the reference is channel-major [B, C, T]): every convolution and linear layer is
 * omni_gemm_tile (omni_talker.h) over row windows; these entry points are the non-GEMM remainder.
 * Host-side mirror: ht_vllm_omni_amd/code2wav.py.  Oracle: oracle/code2wav_oracle.py. */
#ifndef OMNI_CODEC_H
#define OMNI_CODEC_H
#include "omni_talker.h"

#ifdef __cplusplus
extern "C" {
#endif

/* SplitResidualVectorQuantizer.decode (…:768-909) as ONE gather-sum: table fp32 [Q][bins][C] holds, per quantizer, the rows
 * (embedding_sum / clamp(cluster_usage, 1e-5)) . output_proj^T of ITS group (rvq_first for q = 0, rvq_rest otherwise), folded once
 * at load; out bf16 [T, C] = bf16(sum_q table[q][codes[q, t]]).  codes int64 [Q, T], row stride ld_codes (the layout the
 * reference hands the decoder: [1, Q, T] long).  Out-of-range codes are clamped for memory safety; range is the caller's check. */
int omni_codec_rvq_embed(const int64_t* codes, int ld_codes, const float* table, void* out, int T, int Q, int bins, int C,
                         void* stream);

/* Qwen3TTSTokenizerV2DecoderRMSNorm (…:397-414) on the fp32 residual stream: out bf16 [T, H] = w * (x * rsqrt(mean(x^2) + eps)). */
int omni_codec_rmsnorm(const float* x, int ldx, const float* w, float eps, void* out, int T, int H, void* stream);

/* apply_rotary_pos_emb (…:90-121) with the default rope init (…:51-66), positions 0..T-1, in place on the q and k heads of fused
 * qkv rows: bf16 [T, ld], row = [q_heads | kv_heads (k) | kv_heads (v)] x head_dim. */
int omni_codec_rope(void* qkv, int ld, int T, int q_heads, int kv_heads, int head_dim, float theta, void* stream);

/* Causal sliding-window self-attention (…:305-377 with the mask of create_sliding_window_causal_mask: key j visible to query i
 * iff 0 <= i - j < window), softmax in fp32: out bf16 [T, ldo] = heads x head_dim.  head_dim 64 | 128. */
int omni_codec_window_attn(const void* qkv, int ld, void* out, int ldo, int T, int q_heads, int kv_heads, int head_dim, int window,
                           float scale, void* stream);

/* ConvNeXt front (…:226-247): depthwise causal conv (w fp32 [C][taps], b [C]; left padding taps - 1) on the fp32 stream x [T, ldx]
 * followed by LayerNorm over the C channels (ln_w, ln_b, eps): out bf16 [T, C].  C <= 1024. */
int omni_codec_dwconv_ln(const float* x, int ldx, const float* w, const float* b, const float* ln_w, const float* ln_b, float eps,
                         void* out, int T, int C, int taps, void* stream);

/* Last layer (…:983-987, :1027): causal conv of the snake-activated bf16 [T, C] signal to ONE channel (w fp32 [taps][C]) + bias,
 * clamped to [-1, 1]: wav fp32 [T]. */
int omni_codec_out_conv(const void* x, const float* w, float bias, float* wav, int T, int C, int taps, void* stream);

/* One residual unit of a decoder block (…DecoderDecoderResidualUnit, …:726-742) in ONE launch, for the high-rate blocks:
 *     h <- h + conv1x1(snake2(conv7_dilated(s))),   s_next = bf16(snake_next(h)),
 * s = bf16(snake1(h)) (the previous launch's second output), h the fp32 residual stream (in place), all time-major [T, C].
 * The x window of a row block sits in LDS once for all 7 taps and the 7-tap conv's output never leaves the chip.  C 96 | 192,
 * 7 taps, dilation <= 9 (omni_codec_res_unit_supported); other shapes take two omni_gemm_tile launches.  s_next must not alias s.
 *   w1 bf16 fragment-major [C, 7 C] (K index = tap * C + channel), b1 fp32 [C]; snake2_* fp32 [C] (exp(alpha), 1 / (exp(beta) + eps));
 *   w2 bf16 fragment-major [C, C], b2 fp32 [C]; next_* = the SnakeBeta in front of the next consumer of h. */
typedef struct omni_res_unit {
    const void* s; float* h; void* s_next;
    const void* w1; const float* b1; const float* snake2_alpha; const float* snake2_inv_beta;
    const void* w2; const float* b2; const float* next_alpha; const float* next_inv_beta;
    int T, C, dilation;
} omni_res_unit;
int omni_codec_res_unit_supported(int C, int taps, int dilation);
int omni_codec_res_unit(const omni_res_unit* u, void* stream);

#ifdef __cplusplus
}
#endif
#endif
