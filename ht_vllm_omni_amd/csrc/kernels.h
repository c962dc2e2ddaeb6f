// Internal (non-ABI) launchers shared between translation units.
#pragma once
#include <stdint.h>

#include "../../include/omni_talker.h"

// the step's status words (omni_step_io.status), copied by the step's last launch: dst[0] = *src0, dst[1] = *src1, dst[2] = ran
struct omni_step_status { const int32_t* src0; const int32_t* src1; int32_t* dst; int ran; };
int k_sample(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k, float top_p, float rep_penalty,
             uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul, int step_add, int inc_steps, int32_t* out_ids,
             int out_stride, void* stream, int32_t* inc0 = nullptr, int32_t* inc1 = nullptr /* per-row counters to bump */,
             const omni_row_sampling* rows = nullptr /* per-row parameters override the scalars */,
             const int32_t* num_live = nullptr /* device: rows >= *num_live are skipped entirely */,
             const omni_step_status* status = nullptr);
int k_embed(const int32_t* ids, int ids_stride, const void* table, void* out, int T, int hidden, int vocab,
            void* stream);
// rows of `table` (ids == NULL: rows 0..T-1) -> fragment-major residual stream + slab 0 of the sum(r^2) partials;
// zero_slabs > 1: slabs 1 .. zero_slabs - 1 of these rows are cleared (rows that join a stream with that many slabs);
// pstride = rows per slab
int k_gather_frag(const int32_t* ids, int ids_stride, const void* table, void* r_out, float* part_out, int T, int hidden,
                  int vocab, void* stream, int zero_slabs = 1, int pstride = 64);
// the residual-stream GEMMs with an explicit slab stride (64 | 128 rows per slab; M <= pstride)
int k_gemm_resid(const void* x, int ldx, const void* w, const void* bias, void* r_io, int accumulate, float* partials,
                 int* nparts_out, int M, int N, int K, int layout, int pstride, void* stream);
int k_gemm_xnorm(const void* r, const float* partials, int nparts, const void* norm_w, float eps, void* normed_out,
                 const void* w, void* out, int M, int N, int K, int epilogue, const uint8_t* mask, int out_frag, int pstride,
                 void* stream, float mask_fill = -__builtin_inff(),
                 const int32_t* num_live = nullptr /* device: normed_out rows >= *num_live are not written */);
// omni_gemm_bf16_ex with the value written for masked-out logits
int k_gemm_bf16_ex(const void* x, int ldx, const void* w, const void* bias, void* out, int M, int N, int K, int epilogue,
                   const uint8_t* mask, int layout, void* stream, float mask_fill);
// code predictor, positions 0 and 1 of every row in one launch (dense private cache: row b owns block b): qkv rows b
// (position 0) and row1_off + b (position 1), fragment-major output rows likewise
int k_attn_pair01(const void* qkv, int row1_off, const void* qnorm_w, const void* knorm_w, const void* cos_sin, float eps,
                  void* k_cache, void* v_cache, void* out, int B, int q_heads, int kv_heads, int block_size, float sm_scale,
                  void* stream);
// sampler with optional fused gather: gather_out[b] = gather_table[picked id] (bf16 rows of gather_dim)
int k_sample_gather(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k, float top_p,
                    float rep_penalty, uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul, int step_add,
                    int inc_steps, int32_t* out_ids, int out_stride, const void* gather_table, void* gather_out,
                    int gather_dim, float* gather_part /* != NULL: gather_out fragment-major + sum-of-squares slab 0 */,
                    void* stream, int32_t* inc0 = nullptr, int32_t* inc1 = nullptr, const omni_row_sampling* rows = nullptr,
                    const int32_t* num_live = nullptr, const omni_step_status* status = nullptr);
// rmsnorm with out-of-place residual update: residual_out = bf16(residual + delta) (may alias residual)
// out (row-major) and/or out_frag (fragment-major, common.cuh frag_off) receive the normalised rows
int k_rmsnorm(const void* x, const void* delta, const void* residual, void* residual_out, const void* w, void* out,
              void* out_frag, int rows, int hidden, float eps, void* stream,
              const int32_t* out_live = nullptr /* device: rows >= *out_live skip the row-major `out` write */);
// internal attention entry points with a fragment-major output option
int k_attn_decode_fused(const void* qkv, const void* qnorm_w, const void* knorm_w, const int32_t* positions,
                        const void* cos_sin, float eps, void* k_cache, void* v_cache, float* k_scales, float* v_scales,
                        const int32_t* block_table, int bt_stride, const int32_t* seq_lens, int64_t* slot_out, void* out,
                        void* workspace, int B, int q_heads, int kv_heads, int head_dim, int block_size, int kv_dtype,
                        float k_scale, float v_scale, float sm_scale, int max_seq_len, int out_frag,
                        int dense_pos /* >= 0: every row at this position of its own block (no index loads); else -1 */, void* stream,
                        const int32_t* num_live = nullptr /* device: rows >= *num_live write no KV / slot */,
                        const int32_t* rope_delta = nullptr /* device [B]: rotary position = positions[b] + rope_delta[b] */,
                        int rope_rows = 0 /* rows of cos_sin: positions + rope_delta is clamped into the table (0: no clamp) */,
                        const float* scale_dev = nullptr /* fp8 KV: device {k_scale, v_scale} read at run time instead of the arguments */);
int k_paged_attn_prefill(const void* q, const void* k_cache, const void* v_cache, const float* k_scales,
                         const float* v_scales, const int32_t* block_table, int bt_stride, const int32_t* req_of_tok,
                         const int32_t* positions, void* out, int T, int q_heads, int kv_heads, int head_dim, int block_size,
                         int kv_dtype, float k_scale, float v_scale, float sm_scale, int out_frag, void* stream);
// causal prefill attention on the matrix cores (prefill_attn.hip); head_dim 128, q_heads / kv_heads in {1, 2, 4}
bool k_prefill_mfma_supported(int q_heads, int kv_heads, int head_dim);
int k_prefill_mfma(const void* q, const void* k_cache, const void* v_cache, const float* k_scales, const float* v_scales,
                   const int32_t* block_table, int bt_stride, const int32_t* req_of_tok, const int32_t* positions, void* out,
                   int T, int q_heads, int kv_heads, int block_size, int kv_dtype, float k_scale, float v_scale, float sm_scale,
                   int out_frag, void* stream);
// the code predictor as persistent launches (cp_chain.hip): passes (buffer positions) g0 .. g1 - 1 of the layer stack, and with
// `head` the head GEMM + sampler (+ gather of the next input row) of every pass too; supported = the released predictor shape
// (1024 wide, 16 q / 8 kv heads, 3072 intermediate, 2048-entry codebooks) on a GPU with >= 256 CUs
struct omni_chain_head {
    float* logits; int logits_ld, logits_pass;     // row b of group g at logits + b * logits_ld + (g - 1) * logits_pass
    int greedy, top_k; float temperature, top_p; uint32_t seed;
    const int32_t* steps; const uint32_t* row_seed;
    int32_t* codes;                                 // [B][Q]
    // round 6: the step's input assembly and the backbone's first qkv as the TAIL of the all-pass launch (cp_chain.hip): the row owner of the
    // last sampler stage sums the 16 code embeddings + the text step into the backbone's residual stream (mtp_finalize_kernel's arithmetic),
    // then one more stage computes layer 0's qkv rows.  NULL: both stay launches of their own
    const struct omni_chain_tail* tail;
};
struct omni_chain_tail {
    const int32_t* input_ids; const void* embed; int vocab; const void* cp_embed; const void* text_step;
    void* x_out; void* resid; float* part; int64_t* audio_codes; int H;
    const void* wqkv; const void* ln1; void* qkv; int NQ;
};
// the tail's qkv stage is instantiated for the backbone widths of bb_chain.hip's BB_SHAPES; every row group must be live (B > 48)
bool k_cp_chain_tail_supported(const omni_talker_desc& d, int B);
bool k_cp_pair_supported(const omni_talker_desc& d, int B, int greedy, int top_k, float top_p);
int k_cp_pair(const omni_talker_desc& d, const omni_layer_weights* layers, uint16_t* const* k_cache, uint16_t* const* v_cache, int B, int np_in,
              uint16_t* resid, float* part, uint16_t* qkv, uint16_t* attn, uint16_t* act, uint32_t* flags, int32_t* err, const omni_chain_head* head,
              void* stream);
bool k_cp_chain_supported(const omni_talker_desc& d, int pos);
bool k_cp_chain_all_supported(const omni_talker_desc& d, int g0, int greedy, int top_k, float top_p);
int k_cp_chain(const omni_talker_desc& d, const omni_layer_weights* layers, uint16_t* const* k_cache, uint16_t* const* v_cache, int B,
               int g0, int g1, int np_in, uint16_t* resid, float* part, uint16_t* qkv, uint16_t* attn, uint16_t* act, uint32_t* flags,
               int32_t* err, const omni_chain_head* head, void* stream);
// the backbone between two attention launches (o_proj -> gate_up -> down_proj -> next layer's qkv) as one persistent launch
// (bb_chain.hip); supported = a dense shape whose (q width, hidden, intermediate) triple is instantiated there (BB_SHAPES) and whose tile
// grids fit 256 workgroups, at 1-64 rows on a single rank; or the 0.6B shape (bb_chain_small_kernel)
bool k_bb_chain_supported(const omni_talker_desc& d, int B, const omni_ar_peers* ar = nullptr, bool half = false);
bool k_bb_chain_small(const omni_talker_desc& d);       // the 0.6B shape: the segment on the code predictor's 16-row stage set, any batch
// `head` (last layer only, next == NULL): the launch ends in the talker's head -- final norm folded into the lm_head GEMM -> logits, h[t + 1]
struct omni_bb_head { float* logits; void* last_hidden; const int32_t* num_live; float mask_fill; };
struct omni_bb_ar { const omni_ar_peers* attn; const omni_ar_peers* mlp; };      // tensor-parallel rank: the peers of the o_proj / down_proj all-reduces
bool k_bb_chain_head_supported(const omni_talker_desc& d);
int k_bb_chain(const omni_talker_desc& d, const omni_layer_weights& w, const omni_layer_weights* next, const void* attn, void* resid, float* part,
               void* act, void* qkv, int B, float eps, uint32_t* flags, int32_t* err, void* stream, bool small = false,
               const omni_bb_head* head = nullptr, const omni_bb_ar* ar = nullptr, bool half = false);
// the sparse-MoE layer between its attention launch and its expert GEMMs (o_proj -> router | shared gate_up -> shared down + routing) as
// one persistent launch (moe_chain.hip); supported = the Omni talker's shape on a single rank, <= 64 rows
bool k_moe_chain_supported(const omni_talker_desc& d, int B, bool has_ar);
int k_moe_chain(const omni_talker_desc& d, const omni_layer_weights& w, const void* attn, void* resid, float* part, void* normed_rm, void* logits,
                void* act, void* shared, int32_t* topk_idx, void* topk_w, int B, uint32_t* flags, int32_t* err, void* stream);
// ... and its tail: the expert GEMM launches (moe.hip), then { combine into the residual stream -> the NEXT layer's qkv } as one persistent launch
int k_moe_experts_phases(const void* x, const int32_t* topk_idx, const void* topk_w, const void* w_gate_up, const float* s_gate_up, const void* w_down,
                         const float* s_down, void* act_ws, void* y_ws, int T, int H, int I, int E_local, int e0, int top_k, void* stream);
bool k_moe_tail_enabled();
int k_moe_tail(const omni_talker_desc& d, const omni_layer_weights& w, const omni_layer_weights& next, const void* y_ws, const void* normed_rm,
               const void* shared, const int32_t* topk_idx, void* resid, float* part, void* qkv, int B, uint32_t* flags, int32_t* err, void* stream);
// the WHOLE backbone -- qkv(0), then attention -> o_proj -> gate_up -> down_proj -> next qkv per layer -- as one persistent launch
// (bb_all.hip): the layer pointers live in a device table filled at engine creation
size_t k_bb_all_table_bytes(int layers);
int k_bb_all_table(void* table_dev, const omni_talker_desc& d, const omni_layer_weights* layers, void* const* k_cache, void* const* v_cache,
                   float* const* k_scales, float* const* v_scales);
int k_bb_all_set_scales(void* table_dev, int layers, const float* k, const float* v, void* stream);
bool k_bb_all_supported(const omni_talker_desc& d, int B, bool has_ar);
int k_bb_all(const omni_talker_desc& d, const void* table_dev, const omni_step_io* io, void* attn, void* resid, float* part, void* act, void* qkv,
             uint32_t* flags, int32_t* err, void* stream);
// the same segment as a loader / consumer engine (bb_engine.hip): 4 extra waves stream the weights into an LDS FIFO by LDS-DMA
bool k_bb_engine_enabled();
int k_bb_engine(const omni_layer_weights& w, const omni_layer_weights* next, const void* attn, void* resid, float* part, void* act, void* qkv,
                int B, float eps, uint32_t* flags, int32_t* err, void* stream);
