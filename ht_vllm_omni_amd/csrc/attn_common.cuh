// Per-head q / k RMSNorm + neox RoPE shared by the attention kernels (paged_attn.hip) and the code-predictor chain
// (cp_chain.hip): one arithmetic, the same bits in both.
#pragma once
#include "common.cuh"

#define LOG2E 1.4426950408889634f

// one wave, one 128-wide head; lane owns elements l and l + 64 (the RoPE pair): x0 / x1 = the raw bf16 values
__device__ __forceinline__ void head_norm_rope_vals(float x0, float x1, const uint16_t* nw, const uint16_t* cs, float eps,
                                                    int lane, float& y0, float& y1) {
    const float ss = wave_sum(x0 * x0 + x1 * x1);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 128.0f) + eps);
    const float n0 = bfround(bf2f(nw[lane]) * bfround(x0 * rstd));
    const float n1 = bfround(bf2f(nw[lane + 64]) * bfround(x1 * rstd));
    const float c = bf2f(cs[lane]), s = bf2f(cs[64 + lane]);
    y0 = bfround(bfround(n0 * c) + bfround(-n1 * s));
    y1 = bfround(bfround(n1 * c) + bfround(n0 * s));
}
__device__ __forceinline__ void head_norm_rope(const uint16_t* src, const uint16_t* nw, const uint16_t* cs, float eps,
                                               int lane, float& y0, float& y1) {
    head_norm_rope_vals(bf2f(src[lane]), bf2f(src[lane + 64]), nw, cs, eps, lane, y0, y1);
}
