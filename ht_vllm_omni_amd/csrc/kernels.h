// Internal (non-ABI) launchers shared between translation units.
#pragma once
#include <stdint.h>

int k_sample(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k, float rep_penalty,
             uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul, int step_add, int inc_steps, int32_t* out_ids,
             int out_stride, void* stream);
int k_embed(const int32_t* ids, int ids_stride, const void* table, void* out, int T, int hidden, int vocab,
            void* stream);
bool k_gemm_rn_supported(int K);
// sampler with optional fused gather: gather_out[b] = gather_table[picked id] (bf16 rows of gather_dim)
int k_sample_gather(const float* logits, int ld, int B, int V, int greedy, float temperature, int top_k,
                    float rep_penalty, uint8_t* seen, uint32_t seed, int32_t* steps, int step_mul, int step_add,
                    int inc_steps, int32_t* out_ids, int out_stride, const void* gather_table, void* gather_out,
                    int gather_dim, void* stream);
// rmsnorm with out-of-place residual update: residual_out = bf16(residual + delta) (may alias residual)
int k_rmsnorm(const void* x, const void* delta, const void* residual, void* residual_out, const void* w, void* out,
              int rows, int hidden, float eps, void* stream);
