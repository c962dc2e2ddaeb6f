// MFMA operand helpers shared by the skinny GEMM (gemm.hip) and the persistent code-predictor chain (cp_chain.hip):
// the two must produce the same bits, so the arithmetic lives in one place.
#pragma once
#include "common.cuh"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x4 ld16(const uint16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x4 ld16_nt(const uint16_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// PRO_XNORM: the x operand is the fragment-major RESIDUAL stream r; the RMSNorm is applied to each fragment as it is
// consumed: x = w * bf16(r * rstd), rstd from the producer's per-workgroup partial sums (talker_oracle.rms_norm)
__device__ __forceinline__ u32x4 xnorm_frag(u32x4 v, u32x4 nw, float rstd) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lo = bf_lo(nw[e]) * bfround(bf_lo(v[e]) * rstd);
        const float hi = bf_hi(nw[e]) * bfround(bf_hi(v[e]) * rstd);
        v[e] = pack_bf2(lo, hi);
    }
    return v;
}

// SiLU(gate) * up with the bf16 rounding points of HF's bf16 modules (each op rounds)
__device__ __forceinline__ float silu_mul_bf16(float gate_acc, float up_acc) {
    const float gt = bfround(gate_acc);
    const float up = bfround(up_acc);
    const float sl = bfround(gt / (1.0f + expf(-gt)));
    return sl * up;
}
