// Skinny-M (M <= 64) bf16 GEMM for the decode step: out[M,N] = x[M,K] . W[N,K]^T.
//
// HBM-bound weight streaming (64 FLOP/B << MFMA ridge): W is read exactly once, straight from
// HBM into MFMA A-operand registers (no LDS round trip: each W element feeds one wave only);
// x (<= 64 x K, L2-resident) is the B operand.  v_mfma_f32_16x16x32_bf16:
//   A lane(r = l&15, q = l>>4) = W[n0 + r][k0 + 8q .. +8)      (16 rows x 64 B per load)
//   B lane(c = l&15, q)        = x[m0 + c][k0 + 8q .. +8)
//   D[n][m]: lane holds m = l&15, n = 4*(l>>4) + reg.
// One workgroup = 8 waves = one group of NT 16-row n-tiles; the waves split K (k-steps
// interleaved w, w+8, ...) and combine through LDS, so every CU has 8+ KB of W in flight per
// load round.  fp32 accumulate, one rounding to bf16 (oracle: talker_oracle.linear).
#include "common.cuh"

#define GEMM_WAVES 8
#define GEMM_THREADS (GEMM_WAVES * 64)

template <int MT, int NT, int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_skinny_kernel(
    const uint16_t* __restrict__ x, int ldx, const uint16_t* __restrict__ W, const uint16_t* __restrict__ bias,
    void* __restrict__ out, int M, int N, int K, const uint8_t* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [WAVES][NT*MT*4][64]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;

    // row base of each n-tile of this workgroup
    const uint16_t* wrow[NT];
    if (EPI == OMNI_EPI_SILU_MUL) {
        // tile 0 = gate rows, tile 1 = the matching up rows (W = [gate | up], N = inter)
        const int n0 = blockIdx.x * 16;
        wrow[0] = W + (size_t)(n0 + r) * K + 8 * q;
        if (NT > 1) wrow[NT - 1] = W + (size_t)(N + n0 + r) * K + 8 * q;
    } else {
        const int n0 = blockIdx.x * 16 * NT;
#pragma unroll
        for (int j = 0; j < NT; ++j) wrow[j] = W + (size_t)(n0 + j * 16 + r) * K + 8 * q;
    }
    const uint16_t* xrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = i * 16 + r;
        m = m < M ? m : M - 1;   // rows past M: valid address, result discarded
        xrow[i] = x + (size_t)m * ldx + 8 * q;
    }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = K >> 5;          // k-steps of 32
    constexpr int U = 4;
    int s = wave;
    for (; s + (U - 1) * GEMM_WAVES < nsteps; s += U * GEMM_WAVES) {
        uint4 a[U][NT], b[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k0 = (s + u * GEMM_WAVES) << 5;
#pragma unroll
            for (int j = 0; j < NT; ++j) a[u][j] = *reinterpret_cast<const uint4*>(wrow[j] + k0);
#pragma unroll
            for (int i = 0; i < MT; ++i) b[u][i] = *reinterpret_cast<const uint4*>(xrow[i] + k0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, a[u][j]), __builtin_bit_cast(bf16x8, b[u][i]), acc[j][i], 0, 0, 0);
    }
    for (; s < nsteps; s += GEMM_WAVES) {
        const int k0 = s << 5;
        uint4 a[NT], b[MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) a[j] = *reinterpret_cast<const uint4*>(wrow[j] + k0);
#pragma unroll
        for (int i = 0; i < MT; ++i) b[i] = *reinterpret_cast<const uint4*>(xrow[i] + k0);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, b[i]), acc[j][i], 0, 0, 0);
    }

    // ---- combine the 8 K-partials through LDS: lds[wave][e][lane], e = (j*MT+i)*4+reg
    constexpr int E = NT * MT * 4;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) lds[(wave * E + (j * MT + i) * 4 + g) * 64 + lane] = acc[j][i][g];
    __syncthreads();

    // item = (m-tile i, [n-tile j], lane l): 4 consecutive n (reg 0..3) of one row m
    constexpr int NTO = (EPI == OMNI_EPI_SILU_MUL) ? 1 : NT;      // output n-tiles per group
    constexpr int ITEMS = NTO * MT * 64;
    for (int it = threadIdx.x; it < ITEMS; it += GEMM_THREADS) {
        const int l = it & 63;
        const int t = it >> 6;
        const int i = t % MT, j = t / MT;
        const int m = i * 16 + (l & 15);
        if (m >= M) continue;
        float v[4], v2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float sum = 0.f, sum2 = 0.f;
#pragma unroll
            for (int w = 0; w < GEMM_WAVES; ++w) {
                sum += lds[(w * E + (j * MT + i) * 4 + g) * 64 + l];
                if (EPI == OMNI_EPI_SILU_MUL) sum2 += lds[(w * E + ((NT - 1) * MT + i) * 4 + g) * 64 + l];
            }
            v[g] = sum;
            v2[g] = sum2;
        }
        if (EPI == OMNI_EPI_SILU_MUL) {
            const int n = blockIdx.x * 16 + 4 * (l >> 4);
            uint32_t p[2];
            float o[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // HF: down(silu(gate(x)) * up(x)) with bf16 tensors: each op rounds to bf16
                const float gt = bfround(v[g]);
                const float up = bfround(v2[g]);
                const float sl = bfround(gt / (1.0f + expf(-gt)));
                o[g] = sl * up;
            }
            p[0] = pack_bf2(o[0], o[1]);
            p[1] = pack_bf2(o[2], o[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + (size_t)m * N + n) = make_uint2(p[0], p[1]);
        } else {
            const int n = blockIdx.x * 16 * NT + j * 16 + 4 * (l >> 4);
            if (bias) {
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] += bf2f(bias[n + g]);
            }
            if (EPI == OMNI_EPI_BF16) {
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + (size_t)m * N + n) =
                    make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            } else {
                float4 o;
                float* po = reinterpret_cast<float*>(&o);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float y = (EPI == OMNI_EPI_F32_BF16RND) ? bfround(v[g]) : v[g];
                    if (mask && !mask[n + g]) y = -INFINITY;
                    po[g] = y;
                }
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)m * N + n) = o;
            }
        }
    }
}

template <int MT, int NT, int EPI>
static int launch_gemm(const void* x, int ldx, const void* w, const void* bias, void* out, int M, int N, int K,
                       const uint8_t* mask, hipStream_t st) {
    const int groups = (EPI == OMNI_EPI_SILU_MUL) ? N / 16 : N / (16 * NT);
    const size_t lds = (size_t)GEMM_WAVES * NT * MT * 4 * 64 * sizeof(float);
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, EPI>), dim3(groups), dim3(GEMM_THREADS), lds, st,
                       (const uint16_t*)x, ldx, (const uint16_t*)w, (const uint16_t*)bias, out, M, N, K, mask);
    OMNI_CHECK_LAUNCH("omni_gemm_bf16");
    return OMNI_OK;
}

template <int NT, int EPI>
static int dispatch_mt(const void* x, int ldx, const void* w, const void* bias, void* out, int M, int N, int K,
                       const uint8_t* mask, hipStream_t st) {
    if (M <= 16) return launch_gemm<1, NT, EPI>(x, ldx, w, bias, out, M, N, K, mask, st);
    if (M <= 32) return launch_gemm<2, NT, EPI>(x, ldx, w, bias, out, M, N, K, mask, st);
    return launch_gemm<4, NT, EPI>(x, ldx, w, bias, out, M, N, K, mask, st);
}

extern "C" int omni_gemm_bf16(const void* x, int ldx, const void* w, const void* bias, void* out, int M, int N,
                              int K, int epilogue, const uint8_t* mask, void* stream) {
    OMNI_CHECK_ARG(x && w && out, "omni_gemm_bf16: null pointer");
    OMNI_CHECK_ARG(M >= 1 && M <= 64, "omni_gemm_bf16: M=%d outside 1..64", M);
    OMNI_CHECK_ARG(N > 0 && N % 16 == 0, "omni_gemm_bf16: N=%d not a multiple of 16", N);
    OMNI_CHECK_ARG(K > 0 && K % 32 == 0, "omni_gemm_bf16: K=%d not a multiple of 32", K);
    OMNI_CHECK_ARG(ldx >= K && ldx % 8 == 0, "omni_gemm_bf16: ldx=%d (need >= K, multiple of 8)", ldx);
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case OMNI_EPI_BF16:
            OMNI_CHECK_ARG(mask == nullptr, "omni_gemm_bf16: mask needs an fp32 epilogue");
            return dispatch_mt<1, OMNI_EPI_BF16>(x, ldx, w, bias, out, M, N, K, nullptr, st);
        case OMNI_EPI_SILU_MUL:
            OMNI_CHECK_ARG(bias == nullptr && mask == nullptr, "omni_gemm_bf16: silu_mul takes no bias/mask");
            return dispatch_mt<2, OMNI_EPI_SILU_MUL>(x, ldx, w, nullptr, out, M, N, K, nullptr, st);
        case OMNI_EPI_F32:
            return dispatch_mt<1, OMNI_EPI_F32>(x, ldx, w, bias, out, M, N, K, mask, st);
        case OMNI_EPI_F32_BF16RND:
            return dispatch_mt<1, OMNI_EPI_F32_BF16RND>(x, ldx, w, bias, out, M, N, K, mask, st);
        default:
            omni_set_error("omni_gemm_bf16: unknown epilogue %d", epilogue);
            return OMNI_EINVAL;
    }
}
