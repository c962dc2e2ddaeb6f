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
// Written on (lo, hi) PAIRS of one dword: v_pk_mul_f32 + v_cvt_pk_bf16_f32 produce the packed result in place -- the scalar
// form made hipcc pair elements ACROSS dwords and re-interleave the halves afterwards (6 VALU per element; this: 4-5).  Same
// operations on the same values: bit-identical.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t xnorm_pair(uint32_t v, f32x2 w2, f32x2 rstd2) {
    f32x2 a = {bf_lo(v), bf_hi(v)};
    a = a * rstd2;
    const uint32_t p = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2_t));      // bf16(r * rstd), both halves
    f32x2 r = {bf_lo(p), bf_hi(p)};
    r = r * w2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2_t));
}
__device__ __forceinline__ u32x4 xnorm_frag(u32x4 v, u32x4 nw, float rstd) {
    const f32x2 rstd2 = {rstd, rstd};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = xnorm_pair(v[e], (f32x2){bf_lo(nw[e]), bf_hi(nw[e])}, rstd2);
    return v;
}

// PRO 3 (chain_gemm.cuh): the fragment times the norm weight only -- x = bf16(w * r), the row's rstd multiplies the fp32 sums later
__device__ __forceinline__ u32x4 xw_frag(u32x4 v, u32x4 nw) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f32x2 a = {bf_lo(v[e]), bf_hi(v[e])};
        a = a * (f32x2){bf_lo(nw[e]), bf_hi(nw[e])};
        v[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2_t));
    }
    return v;
}

// ---- the sum(r^2) slabs of a row -> one wave partial, in an order that does NOT depend on the rows per tile (round 6: with the rstd applied to
// the fp32 sums -- PRO 3 -- a last-bit difference of rstd between two tile shapes shows in the outputs; the chains and the launch path may
// pick different row tiles for the same GEMM, e.g. the pair pass below 64 rows).  Canonical order for np <= 128 slabs: wave w owns slabs
// w + 8 k (k < 16); quarter j of them (k = j + 4 i) is summed in ascending i; the wave's partial is (q0 + q1) + (q2 + q3); the row total is the
// eight wave partials in wave order.  A wave holds CPW = 64 / XROWS sub-channels per row (lane / XROWS): each sums 4 / CPW quarters, the
// butterfly steps (lane ^ 32, lane ^ 16) pair them as the formula does.
template <int XROWS>
struct SlabOrder {
    static_assert(XROWS == 16 || XROWS == 32 || XROWS == 64, "slab reduction: 16-, 32- or 64-row tiles");
    static constexpr int CPW = 64 / XROWS, QPS = 4 / CPW, PE = 4 * QPS;
    // slab index of entry e of sub-channel `sub` of wave `wave`
    static __device__ __forceinline__ int index(int wave, int sub, int e) { return wave + 8 * ((sub * QPS + e / 4) + 4 * (e % 4)); }
    static __device__ __forceinline__ float reduce(const float (&pv)[PE]) {
        float q[QPS];
#pragma unroll
        for (int j = 0; j < QPS; ++j) q[j] = ((pv[4 * j] + pv[4 * j + 1]) + pv[4 * j + 2]) + pv[4 * j + 3];
        // (each quarter pinned to a register of its own before the cross-quarter additions: left alone, hipcc packs the quarters into register
        //  pairs and adds lo + hi with v_pk_add_f32 op_sel:[0,1] -- the operand form that is corrupted beside another wave's MFMAs,
        //  profiles/r04_pkfma_probe.txt, tests/test_build_rules.py)
#pragma unroll
        for (int j = 0; j < QPS; ++j) asm volatile("" : "+v"(q[j]));
        float s_ = q[0];
        if constexpr (QPS == 4) {
            float lo = q[0] + q[1], hi = q[2] + q[3];
            asm volatile("" : "+v"(lo), "+v"(hi));
            s_ = lo + hi;
        } else if constexpr (QPS == 2) {
            s_ = q[0] + q[1];
        }
        if (CPW == 4) s_ = xor16_sum(s_);        // sub-channels 0 | 1 and 2 | 3 first: (q0 + q1), (q2 + q3)
        if (CPW >= 2) s_ = xor32_sum(s_);        // ... then the two halves of the wave
        return s_;
    }
};

// SiLU(gate) * up with the bf16 rounding points of HF's bf16 modules (each op rounds)
__device__ __forceinline__ float silu_mul_bf16(float gate_acc, float up_acc) {
    const float gt = bfround(gate_acc);
    const float up = bfround(up_acc);
    const float sl = bfround(gt / (1.0f + expf(-gt)));
    return sl * up;
}
