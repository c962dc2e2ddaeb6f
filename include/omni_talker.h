/*
 * omni_talker.h -- C-ABI of the MI355X-native talker AR decode path.
 *
 * Drop-in boundary for the hot path of heiervang-technologies/ht-vllm-omni
 * (R/ = the reference tree, V/ = R/vllm_omni).  The reference has no native code
 * (SURVEY F1): every arithmetic op on this path is reached through vllm==0.18.0
 * Python call sites.  Each entry point below names the reference call site whose
 * work it replaces.  Plain pointers and sizes only; all pointers are DEVICE
 * pointers unless marked host; `stream` is a hipStream_t passed as void*.
 * Every function is asynchronous on `stream`, performs no allocation and no host
 * synchronisation (safe under hipStreamBeginCapture), and returns 0 on success or
 * a negative OMNI_E* code (message via omni_last_error()).  Nothing aborts the
 * process (V/worker error convention, SURVEY 8b "Error conventions").
 *
 * dtypes: bf16 = 16-bit brain float bit pattern; kv_dtype selects the KV-cache
 * storage: OMNI_KV_BF16 (vLLM cache_dtype "auto"), OMNI_KV_FP16, OMNI_KV_FP8 (OCP e4m3fn,
 * "fp8"), OMNI_KV_INT8 (build-defined, no reference semantics: SURVEY F3).
 *
 * KV cache layout (one allocation per layer), the stacked layout that
 * V/distributed/omni_connectors/utils/kv_utils.py:52-55 accepts:
 *     [2][num_blocks][block_size][n_kv_heads][head_dim]  (+ for INT8 a float
 *     scale array [2][num_blocks][block_size][n_kv_heads])
 * slot = block_table[row][pos / block_size] * block_size + pos % block_size.
 * block_size is a power of two (vLLM's 8 ... 256): the attention kernels split a position by shift and mask and refuse
 * any other value.
 */
#ifndef OMNI_TALKER_H
#define OMNI_TALKER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMNI_OK 0
#define OMNI_EINVAL (-1)   /* bad argument / unsupported shape */
#define OMNI_EHIP (-2)     /* HIP launch error */

#define OMNI_KV_BF16 0
#define OMNI_KV_FP8 1
#define OMNI_KV_INT8 2
#define OMNI_KV_FP16 3   /* IEEE half storage (BASELINE config #2's wording): a bf16 model's K / V are exactly representable unless
                          * |x| > 65504 or < 2^-14; same results as OMNI_KV_BF16 for in-range values                            */

/* GEMM epilogues */
#define OMNI_EPI_BF16 0        /* out bf16 [M,N] = bf16(acc + bias)                         */
#define OMNI_EPI_SILU_MUL 1    /* W = [gate rows | up rows] (2N rows); out bf16 [M,N]       */
#define OMNI_EPI_F32 2         /* out fp32 [M,N]                                            */
#define OMNI_EPI_F32_BF16RND 3 /* out fp32 [M,N] holding bf16-rounded values (logits)       */
#define OMNI_EPI_RESID 4       /* omni_gemm_resid only: r += bf16(acc + bias), sum(r^2) slabs */
#define OMNI_EPI_SILU_MUL_GU8 5 /* as SILU_MUL for a fragment-major W whose rows were interleaved first: rows 16t..16t+7 =
                                 * gate[8t..8t+7], rows 16t+8..16t+15 = up[8t..8t+7] -- one 16-row MFMA tile then yields 8
                                 * finished act columns, so any number of tiles per workgroup divides the chip evenly */

const char* omni_last_error(void);
int omni_abi_version(void);   /* 5 (round 6: omni_ar_peers.tile_flags, omni_talker_set_chains modes 2 / 3, a zero-filled attention workspace) */

/* ------------------------------------------------------------------ per-op entry points */

/* Fused residual-add + RMSNorm (vLLM fused_add_rms_norm reached inside Qwen3Model layers,
 * V/model_executor/models/qwen3_tts/qwen3_tts_talker.py:341; HF numerics, see oracle).
 *   delta    : bf16 [rows,hidden] or NULL            (branch output to add)
 *   residual : bf16 [rows,hidden] in/out or NULL     (residual <- bf16(residual + delta))
 *   x        : bf16 [rows,hidden] input when residual == NULL
 *   out      : bf16 [rows,hidden] = w * bf16(v * rsqrt(mean(v^2)+eps)), v = residual (or x)   */
int omni_rmsnorm(const void* x, const void* delta, void* residual, const void* w, void* out,
                 int rows, int hidden, float eps, void* stream);

/* Skinny-M bf16 GEMM  out[M,N] = x[M,K] . W[N,K]^T (+bias[N]), fp32 accumulate on MFMA
 * (vLLM QKVParallelLinear / RowParallelLinear / MergedColumnParallelLinear / ParallelLMHead
 * apply(), reached from qwen3_tts_talker.py:341,344-351,431).  M <= 64.
 *   ldx = row stride of x in elements; mask (EPI_F32*) = uint8 [N] allowed flags or NULL:
 *   disallowed columns are written as -inf (qwen3_tts_talker.py:435).                        */
int omni_gemm_bf16(const void* x, int ldx, const void* w, const void* bias, void* out,
                   int M, int N, int K, int epilogue, const uint8_t* mask, void* stream);

/* Operand layouts for omni_gemm_bf16_ex.  FRAG = fragment-major: the 16-row x 32-k tile of one MFMA operand is 512
 * contiguous elements in lane order (element (row,k) at
 *   ((((row/16) * (K/32) + k/32) * 4 + (k%32)/8) * 16 + row%16) * 8 + k%8 ),
 * so one wave-level load is 1 KB contiguous instead of 64 scattered 16-B accesses.  Weights are shuffled once at
 * load time; activations are written in this layout by the producing kernel (norm, attention, SiLU epilogue). */
#define OMNI_LAYOUT_W_FRAG 1
#define OMNI_LAYOUT_X_FRAG 2     /* x rows padded to a multiple of 16 (ldx ignored)                 */
#define OMNI_LAYOUT_OUT_FRAG 4   /* bf16 epilogues only: out is the next GEMM's x (K' = N)          */
int omni_gemm_bf16_ex(const void* x, int ldx, const void* w, const void* bias, void* out, int M, int N, int K,
                      int epilogue, const uint8_t* mask, int layout, void* stream);

/* act bf16 [T, inter] = bf16(bf16(silu(gate)) * up) for gate_up bf16 [T, 2*inter] = [gate | up]: the SiluAndMul between
 * gate_up_proj and down_proj of vLLM's Qwen3MLP, for the prefill rows whose gate_up GEMM runs on hipBLASLt (the decode
 * GEMM fuses it as OMNI_EPI_SILU_MUL).                                                                                */
int omni_silu_mul(const void* gate_up, void* out, int T, int inter, void* stream);

/* Thinker -> talker projections of the Qwen3-Omni talker's prompt builder (reference: qwen3_omni.py:650-676 _get_tts_embed,
 * 975-992 _get_talker_user_parts, 994-1060 _get_talker_assistant_parts; module = HF Qwen3OmniMoeTalkerResizeMLP):
 * out bf16 [T, H_out] = linear_fc2(silu(linear_fc1(x))), x bf16 [T, H_in] row-major, weights bf16 row-major [I, H_in] /
 * [H_out, I] with bf16 biases (may be NULL); each op rounds to bf16 as the bf16 torch modules do.  act_ws: bf16
 * [min(T, 64), I] scratch.  omni_silu: out = bf16(silu(x)) over n bf16 elements (16-byte aligned, in place allowed).       */
int omni_silu(const void* x, void* out, long long n, void* stream);
int omni_resize_mlp(const void* x, const void* fc1_w, const void* fc1_b, const void* fc2_w, const void* fc2_b, void* act_ws,
                    void* out, int T, int H_in, int I, int H_out, void* stream);

/* Sparse-MoE MLP of the Qwen3-Omni talker backbone (the reference runs vLLM's FusedMoE through
 * V/model_executor/models/qwen3_omni/qwen3_omni_moe_talker.py; arithmetic here = HF Qwen3OmniMoeTalkerTextSparseMoeBlock,
 * the algorithm the oracle restates), decode batch sizes: T <= 64 tokens, top_k <= 8, E <= 256 experts.
 *   omni_moe_route:   logits bf16 [T, E] (= x . W_router^T, e.g. omni_gemm_bf16) -> fp32 softmax -> top_k experts per token
 *                     (ties: lower index), topk_idx int32 [T, k], topk_w bf16 [T, k] (renormalised when norm_topk_prob).
 *   omni_moe_experts: per hit expert e: act = silu(x_e . Wg_e^T) * (x_e . Wu_e^T); y = bf16(act . Wd_e^T) * weight; then per
 *                     token out = sum of its y in ascending expert order (bf16 accumulate) + bf16(sigmoid(bf16(x . w_shared_gate))
 *                     * shared) when shared != NULL (shared bf16 [T, H] = the shared expert's MLP output).
 *                     w_gate_up bf16 [E, 2I, H] = [gate rows | up rows] and w_down bf16 [E, H, I], each expert matrix
 *                     fragment-major (OMNI_LAYOUT_W_FRAG); workspaces act_ws bf16 [T * k, I], y_ws bf16 [T * k, H].     */
int omni_moe_route(const void* logits, int T, int E, int top_k, int norm_topk_prob, int32_t* topk_idx, void* topk_w, void* stream);
int omni_moe_experts(const void* x, const int32_t* topk_idx, const void* topk_w, const void* w_gate_up, const void* w_down,
                     const void* shared, const void* w_shared_gate, void* act_ws, void* y_ws, void* out, int T, int H, int I,
                     int E, int top_k, void* stream);
/* The same block for a tensor- / expert-parallel rank and for fp8 expert weights (BASELINE configs #4 / #5; the reference
 * runs vLLM FusedMoE under tensor_parallel_size, V/model_executor/models/qwen3_omni/qwen3_moe.py:8,152-161):
 *   E_local, e0 : this rank holds experts [e0, e0 + E_local) of the router's numbering (expert parallel); slots routed to
 *                 other ranks' experts contribute zero -- `out` is then a PARTIAL sum the caller all-reduces.  Tensor
 *                 parallel (every expert's intermediate dimension split) needs no special case: pass the shards and I_local.
 *   s_gate_up [E_local, 2I] / s_down [E_local, H] fp32, both or neither: with them w_gate_up / w_down hold fp8 e4m3fn bytes
 *                 (same fragment-major element order), dequantised in registers to bf16(fp8 * row scale) -- bit for bit the
 *                 matrix of a weight-only-dequantised bf16 model -- before the same bf16 MFMA (half the HBM bytes).       */
int omni_moe_experts_ex(const void* x, const int32_t* topk_idx, const void* topk_w, const void* w_gate_up, const float* s_gate_up,
                        const void* w_down, const float* s_down, const void* shared, const void* w_shared_gate, void* act_ws,
                        void* y_ws, void* out, int T, int H, int I, int E_local, int e0, int top_k, void* stream);

/* The same block on the norm-free residual stream (OMNI_EPI_RESID's conventions): instead of `out`,
 *   partial_frag == NULL: resid_frag (bf16 [T, H] fragment-major) += the block's output in place (bf16 add) and part
 *                         (fp32 [H / 16][64]) receives the per-slab sums of squares of the new residual -- the next RMSNorm is
 *                         folded into the GEMMs that read it (omni_gemm_xnorm);
 *   partial_frag != NULL: a tensor- / expert-parallel rank's partial output, fragment-major, for omni_allreduce_resid
 *                         (which adds the ranks' partials into the residual and writes the slabs). */
int omni_moe_experts_resid(const void* x, const int32_t* topk_idx, const void* topk_w, const void* w_gate_up, const float* s_gate_up,
                           const void* w_down, const float* s_down, const void* shared, const void* w_shared_gate, void* act_ws,
                           void* y_ws, void* resid_frag, float* part, void* partial_frag, int T, int H, int I, int E_local, int e0,
                           int top_k, void* stream);

/* Large-M bf16 GEMM on the matrix cores (csrc/gemm_prefill.hip): out[M, N] = epilogue(x[M, K] . W[N, K]^T), fp32 accumulate.
 * The prefill GEMMs of the talker (replaces the F.linear calls vLLM's Qwen3 layers make under
 * vllm_omni/worker/gpu_model_runner.py:1305-1328 for prompt tokens) and every convolution of the Code2Wav decoder
 * (tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:174-224: causal conv1d / transposed conv on TIME-major activations).
 *   x        bf16 row-major, x_rows rows of ldx elements.  K is K / seg_len segments; segment s of output row m reads columns
 *            [0, seg_len) of x row (m + row_off + s * seg_rows); rows outside [0, x_rows) read as zero (causal padding).
 *            Plain GEMM: seg_len = 0 (or K), seg_rows = row_off = 0.  conv1d, kernel k, dilation d, C_in channels:
 *            K = k * C_in, seg_len = C_in, seg_rows = d, row_off = -(k - 1) * d.
 *   w        bf16 [N, K] fragment-major (OMNI_LAYOUT_W_FRAG); N % 16 == 0, K % 32 == 0, seg_len % 32 == 0.
 *   bias / scale  fp32 [N] or NULL.  y = act(acc + bias) * scale (+ resid), all fp32.
 *   act      OMNI_TILE_ACT_NONE | _GELU (erf) | _SILU_MUL_GU8 (w rows interleaved as OMNI_EPI_SILU_MUL_GU8; out is [M, N / 2],
 *            = bf16(bf16(SiLU(bf16 gate)) * bf16 up), omni_silu_mul's rounding points; no other output).
 *   resid    fp32 [M, ldr] or NULL (may alias out_f32: the residual stream updated in place).
 *   out_f32  fp32 [M, ldf] or NULL: y.   out  bf16 [M, ldo] or NULL: bf16(y).
 *   out2     bf16 [M, ldo2] or NULL: bf16(snake(y)), snake(y) = y + inv_beta[n] * sin^2(alpha[n] * y) (the SnakeBeta in front of
 *            the next conv, fp32 [N] parameters as omni_snake_beta).  At least one output.                                    */
#define OMNI_TILE_ACT_NONE 0
#define OMNI_TILE_ACT_GELU 1
#define OMNI_TILE_ACT_SILU_MUL_GU8 2
typedef struct omni_tile_gemm {
    const void* x; int64_t x_rows; int ldx;
    int seg_len, seg_rows, row_off;
    const void* w; const float* bias; const float* scale;
    int act;
    const float* resid; int ldr;
    float* out_f32; int ldf;
    void* out; int ldo;
    void* out2; int ldo2; const float* snake_alpha; const float* snake_inv_beta;
    int M, N, K;
    int tile_hint;   /* 0: automatic; 1: the large output tiles (256 x 256 ...); 2: the 128 x 64 tile of small-M problems;
                        3 / 4: 256 columns x 224 / 192 rows (N % 256 == 0); 5 / 6 / 7: the two-group form of the 256-column tile at
                        256 / 224 / 192 rows (plain GEMMs only; what 0 picks for them; same results) */
    /* grouped (batched) launch, groups > 1: group g computes out rows [g * out_group_rows, + M) from x rows [g * x_group_rows, + M)
     * and the matrix w + g * w_group_elems (elements) -- the expert-sorted [E, cap, H] batch of the MoE prefill against the
     * decode step's per-expert fragment-major weights (replaces the batched GEMMs vLLM's FusedMoE issues under
     * V/model_executor/models/qwen3_omni/qwen3_moe.py:8,152-161).  group_rows (device int32 [groups], or NULL = M each) is the
     * live row count per group: rows past it read as zero and are not written, whole tiles past it are skipped. */
    int groups; int64_t x_group_rows, w_group_elems, out_group_rows; const int32_t* group_rows;
} omni_tile_gemm;
int omni_gemm_tile(const omni_tile_gemm* g, void* stream);

/* SnakeBeta activation of the Code2Wav decoder (next stage after the talker):
 *   out[b, c, t] = x + inv_beta[c] * sin^2(x * exp_alpha[c]),  x / out [B, C, T] contiguous fp32 (is_bf16 = 0) or bf16,
 *   exp_alpha = exp(alpha), inv_beta = 1 / (exp(beta) + 1e-9) fp32 [C].
 * Replaces the Triton kernel of tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:617-700 (SnakeBeta._triton_forward). */
int omni_snake_beta(const void* x, const float* exp_alpha, const float* inv_beta, void* out, int B, int C, int T, int is_bf16,
                    void* stream);

/* The norm-free residual stream of the decode step: the residual r lives fragment-major ([rows16, N], frag_off) next to
 * per-row sum-of-squares slabs partials[nparts][64] fp32, so that the reference's fused_add_rms_norm launch between
 * two linears (Qwen3DecoderLayer, V/model_executor/models/qwen3_tts/qwen3_tts_talker.py:297-311 via vLLM Qwen3Model)
 * disappears: the producing GEMM adds into r and emits its workgroups' shares of sum(r^2), the consuming GEMM applies
 * x = norm_w * bf16(r * rsqrt(sum / K + eps)) to every operand fragment it loads.  Same arithmetic as omni_rmsnorm
 * (fp32 statistics, bf16 roundings in the HF order); the slab sum order is fixed, so results are run-to-run
 * deterministic.
 *   omni_gemm_resid: r_io = bf16((accumulate ? r_io : 0) + bf16(x . W^T + bias)); partials[N/16][64]; *nparts_out = N/16
 *                    (x row-major or fragment-major per layout; N % 32 == 0).
 *   omni_gemm_xnorm: out = epilogue((norm_w * bf16(r * rstd)) . W^T); r and W fragment-major; normed_out (bf16 [M,K]
 *                    row-major or NULL) receives the normalised rows; out_frag as OMNI_LAYOUT_OUT_FRAG.           */
int omni_gemm_resid(const void* x, int ldx, const void* w, const void* bias, void* r_io, int accumulate, float* partials,
                    int* nparts_out, int M, int N, int K, int layout, void* stream);
int omni_gemm_xnorm(const void* r, const float* partials, int nparts, const void* norm_w, float eps, void* normed_out,
                    const void* w, void* out, int M, int N, int K, int epilogue, const uint8_t* mask, int out_frag,
                    void* stream);

/* Per-head q/k RMSNorm + neox RoPE + KV-cache write with quantisation
 * (vLLM Qwen3Attention q_norm/k_norm + rotary_emb + reshape_and_cache; slot mapping built at
 * V/worker/gpu_ar_model_runner.py:239-244, passed via set_forward_context 283-292).
 *   qkv       : bf16 [T, (Hq+2Hkv)*D]      q_out : bf16 [T, Hq*D]
 *   positions : int32 [T]                  cos_sin : bf16 [max_pos][2][D/2] (cos | sin, host-built)
 *   slot_mapping : int64 [T], -1 = padded token (skipped)
 *   k_cache/v_cache : base of the K / V half of one layer's cache; kv_scales: INT8 only.     */
int omni_qknorm_rope_kvwrite(const void* qkv, const void* qnorm_w, const void* knorm_w,
                             const int32_t* positions, const void* cos_sin, const int64_t* slot_mapping,
                             void* q_out, void* k_cache, void* v_cache, float* k_scales, float* v_scales,
                             int T, int q_heads, int kv_heads, int head_dim, float eps,
                             int kv_dtype, float k_scale, float v_scale, void* stream);

/* The same with M-RoPE position ids (vLLM MRotaryEmbedding.forward with `mrope_section`; ids from
 * OmniMRotaryEmbedding.get_input_positions_tensor, V/model_executor/layers/rotary_embedding/mrope.py:64-109): positions3 int32
 * [3, T] = temporal / height / width id of every token, mrope_axis uint8 [64] (device) = the axis whose id rotary pair p uses
 * (chunked sections [24, 20, 20]: 0 x 24, 1 x 20, 2 x 20; interleaved (Qwen3-Omni): p % 3 for p < 60 (axes h, w need
 * p < 3 * section), else 0).  Three identical rows give omni_qknorm_rope_kvwrite's result bit for bit. */
int omni_qknorm_mrope_kvwrite(const void* qkv, const void* qnorm_w, const void* knorm_w, const int32_t* positions3,
                              const uint8_t* mrope_axis, const void* cos_sin, const int64_t* slot_mapping, void* q_out,
                              void* k_cache, void* v_cache, float* k_scales, float* v_scales, int T, int q_heads, int kv_heads,
                              int head_dim, float eps, int kv_dtype, float k_scale, float v_scale, void* stream);

/* Slot mapping for a uniform decode batch, computed on device (vLLM BlockTable.compute_slot_mapping,
 * reached from gpu_ar_model_runner.py:239-244):  slot[r] = bt[r][pos/bs]*bs + pos%bs, -1 when r >= B. */
int omni_slot_mapping(const int32_t* block_table, int bt_stride, const int32_t* positions,
                      int64_t* slot_mapping, int B, int B_padded, int block_size, void* stream);

/* Paged-attention decode, query_len = 1 (vLLM attention backend forward inside _model_forward,
 * gpu_ar_model_runner.py:299-308; metadata built 246-258).
 *   q : bf16 [B, Hq*D]   out : bf16 [B, Hq*D]   block_table : int32 [B, bt_stride]
 *   seq_lens : int32 [B] (context length INCLUDING the current token)
 *   workspace: fp32, >= omni_paged_attn_workspace_bytes(...), or NULL = no KV split.  ABI v5: hand it over ZERO-FILLED -- its first 2 KB
 *   are reserved for arrival counters (the in-launch merge of KV splits, an A/B arm of the debug library; every launch leaves them
 *   zero), the rest is scratch                                                                                                        */
int omni_paged_attn_decode(const void* q, const void* k_cache, const void* v_cache,
                           const float* k_scales, const float* v_scales,
                           const int32_t* block_table, int bt_stride, const int32_t* seq_lens,
                           void* out, void* workspace, int B, int q_heads, int kv_heads, int head_dim,
                           int block_size, int kv_dtype, float k_scale, float v_scale, float sm_scale,
                           int max_seq_len, void* stream);
int64_t omni_paged_attn_workspace_bytes(int B, int q_heads, int head_dim, int max_seq_len);

/* omni_qknorm_rope_kvwrite + omni_paged_attn_decode in ONE launch for a uniform decode batch: takes the
 * raw qkv projection, writes the new token's K/V into the cache (slot computed from block_table and
 * positions, returned in slot_out[B] when non-NULL) and attends over seq_lens[b] keys incl. the new one. */
int omni_attn_decode_fused(const void* qkv, const void* qnorm_w, const void* knorm_w, const int32_t* positions,
                           const void* cos_sin, float eps, void* k_cache, void* v_cache, float* k_scales,
                           float* v_scales, const int32_t* block_table, int bt_stride, const int32_t* seq_lens,
                           int64_t* slot_out, void* out, void* workspace, int B, int q_heads, int kv_heads,
                           int head_dim, int block_size, int kv_dtype, float k_scale, float v_scale,
                           float sm_scale, int max_seq_len, void* stream);

/* Causal paged-attention for prefill / mixed batches (correctness path; the MFMA prefill
 * kernel is the "next" row of SURVEY 8f).  Token t of request r = req_of_tok[t] at absolute
 * position positions[t] attends cache positions 0..positions[t] of r.                         */
int omni_paged_attn_prefill(const void* q, const void* k_cache, const void* v_cache,
                            const float* k_scales, const float* v_scales,
                            const int32_t* block_table, int bt_stride, const int32_t* req_of_tok,
                            const int32_t* positions, void* out, int T, int q_heads, int kv_heads,
                            int head_dim, int block_size, int kv_dtype, float k_scale, float v_scale,
                            float sm_scale, void* stream);

/* Embedding row gather  out[t] = table[ids[t]]  (embed_input_ids, qwen3_tts_talker.py:411-412,637). */
int omni_embed(const int32_t* ids, const void* table, void* out, int T, int hidden, int vocab, void* stream);

/* Sampler (vLLM Sampler reached at gpu_ar_model_runner.py:455; params
 * V/model_executor/stage_configs/qwen3_tts.yaml:27-34).  logits fp32 [B, ld]; V columns used.
 *   greedy != 0 : first argmax.  Otherwise: repetition penalty over seen[B,V] (uint8, may be
 *   NULL) -> /temperature -> top-k (ties kept) -> top-p (0 < top_p < 1: of the candidates sorted by
 *   (value desc, index asc) keep those whose preceding cumulative softmax mass is < top_p, the rule of
 *   qwen3_omni_moe_code_predictor_mtp.py:463-469; needs 0 < top_k <= 1024, and the cut sees at most 1024
 *   candidates: a row with more than 1024 - top_k exact ties AT the k-th value is cut among its first 1024
 *   candidates in index order) -> Gumbel-max with the hash
 *   RNG of the oracle keyed by (seed, steps[b], column).  out_ids int32 [B]; if seen != NULL the sampled id is
 *   marked.  steps int32 [B] (device) is incremented when inc_steps != 0.                     */
int omni_sample(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k, float top_p,
                float rep_penalty, uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul,
                int step_add, int inc_steps, int32_t* out_ids, void* stream);

/* Per-request sampling parameters as per-row device arrays [B] (vLLM keeps one SamplingParams and one seeded
 * torch.Generator per request: V/worker/gpu_model_runner.py:315-319, sampler call gpu_ar_model_runner.py:455).  A NULL
 * member falls back to the launch-wide scalar next to it; the arrays are read inside the (graph-captured) sampler
 * launch, so one captured step serves any mix of requests.  greedy != 0 = argmax (temperature 0 in vLLM terms);
 * top_p in (0,1) needs 0 < top_k <= 1024 on that row (the host validates at admission). */
typedef struct omni_row_sampling {
    const int32_t* greedy;
    const float* temperature;
    const int32_t* top_k;
    const float* top_p;
    const float* rep_penalty;
    const uint32_t* seed;
} omni_row_sampling;

/* omni_sample with every parameter per row (all six arrays required). */
int omni_sample_rows(const float* logits, int ld, int B, int V, const omni_row_sampling* rows, uint8_t* seen,
                     int32_t* steps, int step_mul, int step_add, int inc_steps, int32_t* out_ids, void* stream);

/* One-shot all-reduce of the tensor-parallel decode step over peer-mapped buffers, fused with the residual add and the
 * sum-of-squares slabs of the norm-free stream (the reference initialises vLLM's tensor-parallel group at
 * V/worker/gpu_ar_worker.py:69-75 and all-reduces inside RowParallelLinear; SURVEY 5.8 / 8e: 56 messages of <= 256 KiB per
 * step are latency-bound on a ring).  One process per GPU: every rank allocates its partial buffers and its flag words with
 * omni_ar_alloc (fine-grained device memory + hipIpc handle), exchanges the 64-byte handles out of band
 * (torch.distributed) and maps the peers' with omni_ar_open.
 *   data[p]  : rank p's partial [M16, H] bf16, FRAGMENT-major (written by the GEMM before the call: OMNI_LAYOUT_OUT_FRAG)
 *   flags[p] : rank p's uint32[8] arrival words; epoch: this rank's uint32[2] (epoch, ticket); error: this rank's int32,
 *              set when a peer failed to arrive within the spin bound (the kernel then carries on: never a hung GPU)
 * omni_allreduce_resid: sum over ranks in rank order (fp32, one rounding: identical bits on every rank);
 *   r_io (fragment-major residual stream, may be NULL) = bf16((accumulate ? r_io : 0) + sum), partials[H/16][pstride] slabs of
 *   sum(r^2) (may be NULL), out_rowmajor (bf16 [M, H], may be NULL) = the sum itself.  M <= 64, H % 32 == 0. */
typedef struct omni_ar_peers {
    int world, rank;
    const void* data[8];
    uint32_t* flags[8];
    uint32_t* epoch;
    int32_t* error;
    /* ABI v5: per-tile arrival flags for the all-reduce INSIDE the backbone's persistent launches (csrc/bb_chain.hip): tile_flags[p] = rank
     * p's array of world x 256 zero-initialised uint32 (index [source rank][tile]) in the same kind of memory as flags[p].  All NULL: the
     * tensor-parallel backbone stays on the launch-per-op path with omni_allreduce_resid launches (a one-rank group needs none). */
    uint32_t* tile_flags[8];
} omni_ar_peers;
int omni_allreduce_resid(const omni_ar_peers* peers, void* r_io, int accumulate, float* partials, int pstride, void* out_rowmajor,
                         int M, int H, void* stream);
int omni_ar_alloc(int64_t bytes, void** ptr, void* ipc_handle64);   /* zero-filled; handle = hipIpcMemHandle_t bytes */
int omni_ar_open(const void* ipc_handle64, void** ptr);
int omni_ar_close(void* ptr);
int omni_ar_free(void* ptr);

/* ------------------------------------------------------------------ talker engine */

typedef struct omni_layer_weights {
    const void* ln1;    /* bf16 [H]               */
    const void* wqkv;   /* bf16 [(Hq+2Hkv)D, H]   */
    const void* qnorm;  /* bf16 [D]               */
    const void* knorm;  /* bf16 [D]               */
    const void* wo;     /* bf16 [H, Hq*D]         */
    const void* ln2;    /* bf16 [H]               */
    const void* wgu;    /* bf16 [2I, H] = [gate | up]; under desc.frag_layout: OMNI_EPI_SILU_MUL_GU8 row order */
    const void* wdown;  /* bf16 [H, I]            */
    /* sparse-MoE MLP instead of wgu / wdown when desc.moe_experts > 0 (backbone layers of the Omni talker) */
    const void* moe_router;          /* bf16 [E, H]                                   */
    const void* moe_gate_up;         /* bf16 [E, 2*Im, H], each expert fragment-major  */
    const void* moe_down;            /* bf16 [E, H, Im],   each expert fragment-major  */
    const void* moe_shared_gate_up;  /* bf16 [2*Is, H]  ([gate | up] rows)             */
    const void* moe_shared_down;     /* bf16 [H, Is]    (layout as wdown)              */
    const void* moe_shared_gate;     /* bf16 [H]                                       */
    /* ABI v2: fp8 e4m3fn expert weights (desc.moe_w8): moe_gate_up / moe_down hold bytes, these the per-row fp32 scales */
    const float* moe_gate_up_scale;  /* fp32 [E_local, 2*Im]                           */
    const float* moe_down_scale;     /* fp32 [E_local, H]                              */
} omni_layer_weights;

typedef struct omni_talker_desc {
    /* backbone dims (per TP rank) */
    int hidden, layers, q_heads, kv_heads, head_dim, inter, vocab;
    int codebook, num_code_groups;
    float eps;
    /* code predictor dims */
    int cp_hidden, cp_layers, cp_q_heads, cp_kv_heads, cp_head_dim, cp_inter;
    int has_cp_projection;
    int frag_layout;   /* != 0: every GEMM weight below is fragment-major (OMNI_LAYOUT_W_FRAG); activations follow */
    /* backbone MLP = sparse MoE when moe_experts > 0 (omni_moe_route / omni_moe_experts).  With fused_norm the router weight
       must be fragment-major too (it takes the fused-norm prologue and leaves the normalised rows for the expert kernels) */
    int moe_experts, moe_top_k, moe_inter, moe_shared_inter, moe_norm_topk;
    int fused_norm;    /* != 0 (needs frag_layout, the folded tables, single rank): the decode step keeps the residual
                          stream fragment-major and folds every RMSNorm into its neighbouring GEMMs (omni_gemm_resid /
                          omni_gemm_xnorm); the per-phase attn_out / mlp_out buffers are then NOT produced */
    int cp_fused_norm; /* the same for the code predictor alone: it is replicated and has no all-reduce inside, so tensor-
                          parallel ranks and MoE backbones (fused_norm == 0) still run it on the norm-free stream */
    /* runtime */
    int max_batch, block_size, kv_dtype, max_model_len, bt_stride;
    float k_scale, v_scale;
    float masked_logit;  /* value written for logits the mask removes: 0 = -inf (Qwen3-TTS compute_logits, qwen3_tts_talker.py:424-443);
                            the Qwen3-Omni talker writes -1e9 (qwen3_omni.py:1143-1149) -- same picks, finite logits */
    /* weights (device) */
    const void* embed;                 /* bf16 [vocab, H]                    */
    const omni_layer_weights* layer;   /* HOST array [layers]                */
    const void* final_norm;            /* bf16 [H]                           */
    const void* lm_head;               /* bf16 [vocab, H]                    */
    const uint8_t* allowed_mask;       /* uint8 [vocab]                      */
    const void* cos_sin;               /* bf16 [max_model_len][2][D/2]       */
    const void* cp_proj_w;             /* bf16 [Hc, H] or NULL               */
    const void* cp_proj_b;             /* bf16 [Hc]   or NULL                */
    const omni_layer_weights* cp_layer;/* HOST array [cp_layers]             */
    const void* cp_norm;               /* bf16 [Hc]                          */
    const void* cp_lm_head;            /* bf16 [Q-1][codebook][Hc]           */
    const void* cp_embed;              /* bf16 [Q-1][codebook][H]            */
    const void* cp_cos_sin;            /* bf16 [Q+1][2][Dc/2]                */
    /* optional constant-folded projections (bit-identical to projecting at run time):
     * cp_proj_table[g][c] = small_to_mtp_projection(cp_embed[g][c]), cp_e0_table[c] = projection(embed[c]) */
    const void* cp_proj_table;         /* bf16 [Q-1][codebook][Hc] or NULL   */
    const void* cp_e0_table;           /* bf16 [vocab][Hc] or NULL           */
    /* KV caches (device), HOST arrays [layers] of K-half / V-half base pointers */
    void* const* k_cache;
    void* const* v_cache;
    float* const* k_scales;            /* INT8 only, else NULL               */
    float* const* v_scales;
    /* scratch (device), sized by omni_talker_scratch_bytes() */
    void* scratch;
    int64_t scratch_bytes;
    /* ABI v2: tensor-parallel ranks with peer-mapped partial buffers (both NULL: the host all-reduces omni_talker_attn_out /
     * omni_talker_mlp_out between the phase calls).  Same flags / epoch / error words in both, different data buffers: the
     * o_proj and down_proj all-reduces alternate.  With them a tensor-parallel rank runs fused_norm = 1 and
     * omni_talker_decode_step is the whole step (the all-reduces are launches of the step, captured with it). */
    const omni_ar_peers* ar_attn;
    const omni_ar_peers* ar_mlp;
    /* sparse-MoE backbone on a tensor- / expert-parallel rank: moe_experts stays the ROUTER's expert count; this rank holds
     * experts [moe_e0, moe_e0 + moe_experts_local) (0 local = all of them), each with moe_inter columns (already the shard);
     * moe_w8 != 0: fp8 expert weights + scales in omni_layer_weights */
    int moe_e0, moe_experts_local, moe_w8;
    /* ABI v3: != 0 lets the code predictor's layer stack run as persistent launches (csrc/cp_chain.hip) where the shape is
     * supported.  The grid of such a launch must be co-resident (256 workgroups, one per CU): leave it 0 for engines whose
     * steps run CONCURRENTLY with another engine's on the same GPU (parallel graph branches). */
    int cp_chain;
    /* ABI v4: rows of `cos_sin` (0 = max_model_len).  M-RoPE models build the table longer than max_model_len: a prompt's rotary
     * ids may run ahead of its token count (video temporal ids), and every later position is index + mrope_position_delta.  The
     * decode kernels clamp positions[b] + rope_delta[b] into the table (backstop; the host refuses requests that would leave it). */
    int rope_rows;
} omni_talker_desc;

typedef struct omni_talker omni_talker;

int64_t omni_talker_scratch_bytes(const omni_talker_desc* desc);
omni_talker* omni_talker_create(const omni_talker_desc* desc);   /* copies the descriptor */
void omni_talker_destroy(omni_talker* t);
/* The code predictor's layer stack runs as persistent launches whose stages wait for each other on flag words with BOUNDED
 * spins (csrc/cp_chain.hip; reference: the decoder loop of qwen3_tts_code_predictor_vllm.py:480-561).  A spin that runs out
 * (grid not co-resident) is recorded in a sticky device word and the launch finishes without waiting: results of that step
 * are wrong, the GPU never hangs.  Returns that word (0 = no wait ever timed out; else 16 * layer + stage + 1 of the first
 * one) or a negative OMNI_E* code; synchronises the device (call it per output hand-over, not per launch).  reset == 1
 * clears the word and the flags; reset == 2 SETS the word (fault injection for tests of the host's fall-back: the next
 * chain launches stop waiting, exactly as after a real time-out).                                                        */
int omni_talker_chain_error(omni_talker* t, int reset);
/* ABI v4.  Switch the persistent chains of an engine on / off at run time (the launch-per-op path computes the same bits):
 * what a host does after a chain flag wait timed out (omni_step_io.status) -- reset the words, turn the chains off, re-capture
 * its graphs, redo the step: a degraded stage, not a dead one.  The GPU may be shared with another process then.        */
int omni_talker_set_chains(omni_talker* t, int on);
/* ABI v5.  on == 2: the HALF grid -- the backbone's persistent launches run on 128 workgroups (each plays two of the 256 stage workgroups),
 * so that two engines' launches are co-resident on one GPU (a co-located second stage; two tensor-parallel ranks of a test on one GPU); the
 * code predictor runs launch per op then.  Same bits as on == 1 and on == 0.
 * on == 3: as 1, but a tensor-parallel rank's backbone keeps omni_allreduce_resid launches between launch-per-op GEMMs instead of the
 * all-reduce inside its persistent launches (what tp_comm.check_backbone_chain selects when its start-up comparison fails on any rank). */
/* Which persistent chains the LAST decode-step call of this engine launched (host-side record, no device access):
 * bit 0 the code-predictor chain (cp_chain.hip), bit 1 the backbone chain (bb_chain.hip).  0 = launch-per-op path.      */
int omni_talker_chains_ran(const omni_talker* t);
/* ABI v4.  Per-layer fp8 KV scales (host arrays [layers], all > 0) -- the result of a `calculate_kv_scales` pass: the reference
 * runs its first forward eager so that vLLM's attention layers can set k_scale = max|k| / 200, v_scale = max|v| / 100 from that
 * pass (V/worker/gpu_ar_model_runner.py:122,269-275).  The decode step's attention launches read the scales
 * from a device table (so graphs captured BEFORE the call see them), the engine's eager prefill launches take them as arguments.
 * Until the call every layer uses desc.k_scale / desc.v_scale.  Synchronises `stream`.                                      */
int omni_talker_set_kv_scales(omni_talker* t, const float* k_scale, const float* v_scale, void* stream);

/* Per-step device buffers (persistent, graph-stable addresses).  Row r = batch slot r. */
typedef struct omni_step_io {
    int B;                        /* live decode rows, 1..max_batch                        */
    int32_t* input_ids;           /* [B] in: last sampled layer-0 id; out: newly sampled    */
    int32_t* positions;           /* [B] in: position of the token computed this step; +1   */
    int32_t* seq_lens;            /* [B] in: context length incl. this token; +1            */
    const int32_t* block_table;   /* [B, bt_stride]                                         */
    int64_t* slot_mapping;        /* [B] out: slots written this step (bit-exact parity)    */
    void* last_hidden;            /* bf16 [B,H] in: h[t]; out: h[t+1] (postprocess)         */
    const void* text_step;        /* bf16 [B,H] text-step vector of this step               */
    void* inputs_embeds;          /* bf16 [B,H] out: x[t+1] fed to the backbone             */
    int64_t* audio_codes;         /* [B,Q] out: frame [c0..c15][t]                          */
    float* logits;                /* [B,vocab] out                                          */
    uint8_t* seen;                /* [B,vocab] repetition-penalty bitmap or NULL            */
    int32_t* steps;               /* [B] per-request generated-token counters (RNG key)     */
    /* sampling */
    int greedy;                   /* talker layer-0 sampler                                 */
    float temperature; int top_k; float rep_penalty; uint32_t seed;
    int cp_greedy;                /* code predictor sub-steps                               */
    float cp_temperature; int cp_top_k;
    int advance;                  /* !=0: positions/seq_lens += 1 after the step (on device)*/
    float top_p, cp_top_p;        /* nucleus cut after top-k (>= 1 or <= 0: off)            */
    /* ABI v2 */
    const int32_t* num_live;      /* device int32 or NULL (= B): rows [*num_live, B) of a padded graph bucket are INERT --
                                     no KV-cache write, no slot_mapping / last_hidden / input_ids / seen / steps / positions /
                                     seq_lens update (they may hold live prefill rows of the persistent batch)            */
    omni_row_sampling rows;       /* layer-0 sampler parameters per row; rows.seed also keys the code predictor's noise    */
    /* ABI v3 */
    const int32_t* rope_delta;    /* device int32 [B] or NULL: the backbone's rotary position of row b is positions[b] +
                                     rope_delta[b] (the request's mrope_position_delta: after a prompt whose M-RoPE ids ran
                                     ahead of / behind its token count, vLLM get_next_input_positions); cache slots and the
                                     attention context still follow positions[] / seq_lens[] */
    /* ABI v4 */
    int32_t* status;              /* device int32 [4] or NULL, written by the LAST launch of the step (so that it rides in the
                                     host's per-step copy of input_ids): [0] the sticky chain error word (omni_talker_chain_error
                                     without the device synchronisation), [1] the sticky error word of the peer all-reduce (0 on a
                                     single rank), [2] omni_talker_chains_ran of this step, [3] 0                          */
} omni_step_io;

/* The four phases of one decode step.  With TP > 1 the host
 * all-reduces (RCCL) `omni_talker_attn_out` after layer_attn and `omni_talker_mlp_out`
 * after layer_mlp; with TP == 1 omni_talker_decode_step runs everything.
 *   mtp        : _preprocess decode branch + _talker_mtp_forward + talker_mtp
 *                (gpu_model_runner.py:1211-1303, qwen3_tts_talker.py:615-647,1594-1642,
 *                 qwen3_tts_code_predictor_vllm.py:480-561)
 *   layer_attn : RMSNorm -> qkv_proj -> q/k-norm+RoPE+KV write -> paged attention -> o_proj
 *   layer_mlp  : (+residual) RMSNorm -> gate_up_proj -> SiLU*mul -> down_proj
 *   finish     : final norm, compute_logits (qwen3_tts_talker.py:424-443), sampler
 *                (gpu_ar_model_runner.py:455), postprocess (qwen3_tts_talker.py:649-655)     */
int omni_talker_mtp(omni_talker* t, const omni_step_io* io, void* stream);
int omni_talker_layer_attn(omni_talker* t, const omni_step_io* io, int layer, void* stream);
int omni_talker_layer_mlp(omni_talker* t, const omni_step_io* io, int layer, void* stream);
int omni_talker_finish(omni_talker* t, const omni_step_io* io, void* stream);
int omni_talker_decode_step(omni_talker* t, const omni_step_io* io, void* stream);
/* every layer + finish, without the mtp phase: the backbone half of a step on whatever residual stream the last step left
 * (timing attribution only: bench.py roofline.breakdown, scripts/ab_knobs.py)                                             */
int omni_talker_backbone_step(omni_talker* t, const omni_step_io* io, void* stream);
/* ABI v4, timing attribution only (bench.py roofline.families): launch a subset of the step's parts -- 1 the mtp phase (code
 * predictor + input assembly), 2 the backbone's paged-attention launches, 4 the rest of the backbone stack (2 and 4 apart only
 * where the backbone runs as attention + chain launches), 8 final norm + lm_head + sampler.  A part run alone reads whatever the
 * buffers hold: its TIME is the step's, its outputs are not.  15 = omni_talker_decode_step.                                   */
int omni_talker_step_part(omni_talker* t, const omni_step_io* io, int parts, void* stream);
void* omni_talker_attn_out(omni_talker* t);   /* bf16 [max_batch,H] (TP all-reduce buffer)  */
void* omni_talker_mlp_out(omni_talker* t);    /* bf16 [max_batch,H]                         */

/* Prefill / mixed batch through the backbone only (T tokens, several requests):
 * x bf16 [T,H] -> hidden bf16 [T,H] (final-normed); writes KV at slot_mapping.
 * Correctness path for TTFA head (SURVEY 8f rank 2 is its MFMA replacement).                */
int omni_talker_prefill(omni_talker* t, const void* x, const int32_t* positions,
                        const int32_t* req_of_tok, const int64_t* slot_mapping,
                        const int32_t* block_table, void* hidden_out, int T, void* stream);

/* The same prefill path one phase at a time for <= max_batch rows (tensor-parallel hosts all-reduce
 * omni_talker_attn_out / omni_talker_mlp_out between the phases, exactly as in the decode step). */
int omni_talker_rows_begin(omni_talker* t, const void* x, int rows, void* stream);
int omni_talker_rows_attn(omni_talker* t, int layer, int rows, const int32_t* positions, const int64_t* slot_mapping,
                          const int32_t* block_table, const int32_t* req_of_tok, void* stream);
int omni_talker_rows_mlp(omni_talker* t, int layer, int rows, void* stream);
int omni_talker_rows_end(omni_talker* t, void* hidden_out, int rows, void* stream);

/* compute_logits on arbitrary rows: hidden bf16 [R,H] -> logits fp32 [R,vocab] (masked). */
int omni_talker_logits(omni_talker* t, const void* hidden, float* logits, int R, int round_bf16, void* stream);

/* Code predictor alone (parity tests): layer0 ids + layer0 embeds + last hidden -> codes.
 * (qwen3_tts_code_predictor_vllm.py:480-561).  cp_logits fp32 [B,Q-1,codebook] or NULL.     */
int omni_talker_code_predictor(omni_talker* t, const int32_t* layer0_ids, const void* layer0_embed,
                               const void* last_hidden, int64_t* codes, float* cp_logits, int B,
                               int greedy, float temperature, int top_k, float top_p, uint32_t seed,
                               const int32_t* steps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNI_TALKER_H */
