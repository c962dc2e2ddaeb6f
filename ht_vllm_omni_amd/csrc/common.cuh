// Shared device helpers for the gfx950 (CDNA4, wave64) talker kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/omni_talker.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

// Policy knobs: compile-time constants in libomni_talker.so; run-time variables with omni_debug_* setters only in
// libomni_talker_debug.so (built with -DOMNI_DEBUG_HOOKS from the same sources + debug.hip; include/omni_talker_debug.h)
#ifdef OMNI_DEBUG_HOOKS
#define OMNI_KNOB static int
#else
#define OMNI_KNOB static constexpr int
#endif

// ---- error plumbing (host)
void omni_set_error(const char* fmt, ...);
#define OMNI_CHECK_ARG(cond, ...)                \
    do {                                         \
        if (!(cond)) {                           \
            omni_set_error(__VA_ARGS__);         \
            return OMNI_EINVAL;                  \
        }                                        \
    } while (0)
#define OMNI_CHECK_LAUNCH(name)                                                   \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            omni_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return OMNI_EHIP;                                                     \
        }                                                                         \
    } while (0)

// ---- bf16 <-> f32 (bit-exact RNE; v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (bf16_t)f); }
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }
// one v_cvt_pk_bf16_f32 (the scalar form costs two conversions, a shift and an or; same round-to-nearest-even values)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }

// ---- IEEE half <-> f32 (OMNI_KV_FP16 storage)
__device__ __forceinline__ float h2f(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float h_lo(uint32_t w) { return h2f((uint16_t)(w & 0xFFFFu)); }
__device__ __forceinline__ float h_hi(uint32_t w) { return h2f((uint16_t)(w >> 16)); }
#define OMNI_KV_IS16(KV) ((KV) == OMNI_KV_BF16 || (KV) == OMNI_KV_FP16)

// ---- lane exchanges inside a 16-lane row as DPP modifiers (~1 VALU op; hipcc lowers __shfl_xor to ds_bpermute_b32, an
// LDS-pipe round trip of ~120 cycles that a one-wave-per-SIMD kernel cannot hide).  Full waves only (a disabled source
// lane reads as 0).  For REDUCTIONS any pairing that merges disjoint groups works, so the 8- and 16-lane steps use the
// mirror controls: after xor1 + xor2 every lane holds its quad's total, half_mirror (i <-> 7 - i) brings in the other quad
// of the 8-lane group, mirror (i <-> 15 - i) the other half of the row.
#define OMNI_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define OMNI_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define OMNI_DPP_HALF_MIRROR 0x141  // row_half_mirror
#define OMNI_DPP_MIRROR 0x140       // row_mirror
#define OMNI_DPP_ROR8 0x128         // row_ror:8  == lane ^ 8 inside the row
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum / max over the 8-lane group (lane >> 3) and over the 16-lane row, result in every lane of the group
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_f<OMNI_DPP_XOR1>(v);
    v += dpp_f<OMNI_DPP_XOR2>(v);
    v += dpp_f<OMNI_DPP_HALF_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float row16_sum(float v) { v = group8_sum(v); return v + dpp_f<OMNI_DPP_MIRROR>(v); }
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<OMNI_DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<OMNI_DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<OMNI_DPP_HALF_MIRROR>(v));
    return fmaxf(v, dpp_f<OMNI_DPP_MIRROR>(v));
}

// ---- exchanges ACROSS rows: gfx950's v_permlane16_swap / v_permlane32_swap (VALU, no LDS pipe).  With both operands = v,
// permlane16_swap leaves {rows 0,0,2,2 of v} in the first result and {rows 1,1,3,3} in the second; permlane32_swap
// {lanes 0-31 twice} and {lanes 32-63 twice}: their sum / max is the lane ^ 16 / lane ^ 32 butterfly step, and picking the
// first result on the odd rows (upper half) and the second on the even rows (lower half) is the plain exchange.
// sum of four squares with the contraction written out (hipcc fuses `a * a + b * b + ...` as it pleases: the residual epilogues of gemm.hip
// and chain_gemm.cuh and the all-reduce launch must put the SAME bits into a sum(r^2) slab)
__device__ __forceinline__ float sq4_sum(float a, float b, float c, float d) { return fmaf(d, d, fmaf(c, c, fmaf(b, b, a * a))); }
__device__ __forceinline__ float xor16_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xchg16(float v, bool odd_row) {      // value of lane ^ 16; odd_row = (lane & 16) != 0
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(odd_row ? r[0] : r[1]);
}
__device__ __forceinline__ float xchg32(float v, bool upper) {        // value of lane ^ 32; upper = (lane & 32) != 0
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(upper ? r[0] : r[1]);
}

// ---- wave-wide argmax of (value, index) pairs, larger value first, smaller index on ties; every lane gets the winner
__device__ __forceinline__ void argmax_merge(float& v, int& idx, float ov, int oi) {
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
}
template <int CTRL>
__device__ __forceinline__ void argmax_dpp(float& v, int& idx) {
    const float ov = dpp_f<CTRL>(v);
    const int oi = __builtin_amdgcn_update_dpp(0, idx, CTRL, 0xF, 0xF, true);
    argmax_merge(v, idx, ov, oi);
}
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
    argmax_dpp<OMNI_DPP_XOR1>(v, idx);
    argmax_dpp<OMNI_DPP_XOR2>(v, idx);
    argmax_dpp<OMNI_DPP_HALF_MIRROR>(v, idx);
    argmax_dpp<OMNI_DPP_MIRROR>(v, idx);
    {
        const auto rv = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        const auto ri = __builtin_amdgcn_permlane16_swap((unsigned)idx, (unsigned)idx, false, false);
        float a = __uint_as_float(rv[0]); int ai = (int)ri[0];
        argmax_merge(a, ai, __uint_as_float(rv[1]), (int)ri[1]);
        v = a; idx = ai;
    }
    {
        const auto rv = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        const auto ri = __builtin_amdgcn_permlane32_swap((unsigned)idx, (unsigned)idx, false, false);
        float a = __uint_as_float(rv[0]); int ai = (int)ri[0];
        argmax_merge(a, ai, __uint_as_float(rv[1]), (int)ri[1]);
        v = a; idx = ai;
    }
}

// ---- wave64 reductions (all 64 lanes hold the result): 4 DPP steps inside the rows + 2 permlane swaps, no LDS traffic
__device__ __forceinline__ float wave_sum(float v) { return xor32_sum(xor16_sum(row16_sum(v))); }
__device__ __forceinline__ float wave_max(float v) { return xor32_max(xor16_max(row16_max(v))); }

// ---- OCP e4m3fn <-> f32 (saturating, round-nearest-even: clamp then hardware cvt)
#define OMNI_FP8_MAX 448.0f
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -OMNI_FP8_MAX), OMNI_FP8_MAX);
    b = fminf(fmaxf(b, -OMNI_FP8_MAX), OMNI_FP8_MAX);
    c = fminf(fmaxf(c, -OMNI_FP8_MAX), OMNI_FP8_MAX);
    d = fminf(fmaxf(d, -OMNI_FP8_MAX), OMNI_FP8_MAX);
    int p = 0;
    p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, p, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (uint32_t)p;
}
__device__ __forceinline__ void unpack_fp8x4(uint32_t w, float* o) {
    f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false);
    f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
    o[0] = lo[0]; o[1] = lo[1]; o[2] = hi[0]; o[3] = hi[1];
}

// ---- the oracle's counter-based uniform (oracle/talker_oracle.py hash_uniform)
__device__ __forceinline__ float hash_uniform(uint32_t seed, uint32_t step, uint32_t idx) {
    uint32_t x = (idx * 0x9E3779B1u) ^ seed;
    x += step * 0x85EBCA77u;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// ---- fragment-major layout of a [rows, K] bf16 matrix (rows padded to 16, K % 32 == 0): the 16-row x 32-k tile an
// MFMA 16x16x32 operand consumes is 512 contiguous elements in LANE order (lane = 16 * kchunk + row % 16, 8 elements per
// lane), tiles ordered [row tile][k step].  One wave-level 16-B load of a fragment = 1 KB contiguous.
__host__ __device__ __forceinline__ size_t frag_off(int row, int k, int K) {
    return ((((size_t)(row >> 4) * (K >> 5) + (k >> 5)) * 4 + ((k & 31) >> 3)) * 16 + (row & 15)) * 8 + (k & 7);
}
