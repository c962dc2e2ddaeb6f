// Skinny-M (M <= 64) bf16 GEMM for the decode step: out[M,N] = x[M,K] . W[N,K]^T, with the
// residual-add + RMSNorm of the producing layer optionally fused into the prologue.
//
// HBM-bound weight streaming (64 FLOP/B << MFMA ridge): W is read exactly once, straight from
// HBM into MFMA A-operand registers with non-temporal loads (each W element feeds one wave only,
// and keeping it out of L2 leaves the activations and kernel code L2-resident); x (<= 64 x K,
// L2-resident) is the B operand.  v_mfma_f32_16x16x32_bf16:
//   A lane(r = l&15, q = l>>4) = W[n0 + r][k0 + 8q .. +8)      (16 rows x 64 B per load)
//   B lane(c = l&15, q)        = x[m0 + c][k0 + 8q .. +8)
//   D[n][m]: lane holds m = l&15, n = 4*(l>>4) + reg.
// Workgroup = 8 waves = one group of NT 16-row n-tiles x MT 16-row m-tiles; grid = (n groups,
// m splits) sized to >= 256 workgroups.  The 8 waves split K (k-steps interleaved w, w+8, ...),
// loads are software-pipelined two groups deep, partial sums combine through LDS.
// fp32 accumulate, one rounding to bf16 (oracle: talker_oracle.linear / rms_norm).
//
// PRO_RN prologue (K = hidden <= 2048, K % 256 == 0): every workgroup rebuilds the normalised
// activation rows it needs from the residual stream:
//   r = bf16(resid + delta);  x = w * bf16(r * rsqrt(mean(r^2) + eps))
// (64 x K elements, ~2 us of VALU, instead of a separate 5-6 us kernel launch); the workgroups
// with blockIdx.x == 0 write r to resid_out -- a DIFFERENT buffer than resid: other workgroups are
// still reading the input -- and x to normed_out when the caller wants the normalised rows.
#include "common.cuh"
#include "kernels.h"

#define GEMM_WAVES 8
#define GEMM_THREADS (GEMM_WAVES * 64)
#define GEMM_U 4          // k-steps per pipelined load group
#define RN_MAX_STEPS 8    // PRO_RN: k-steps per wave (K <= 32 * 8 * 8 = 2048)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld16(const uint16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x4 ld16_nt(const uint16_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct GemmArgs {
    const uint16_t* x; int ldx;
    const uint16_t* W; const uint16_t* bias; void* out;
    int M, N, K;
    const uint8_t* mask;
    // PRO_RN
    const uint16_t* resid; uint16_t* resid_out; const uint16_t* delta; const uint16_t* norm_w; float eps; uint16_t* normed_out;
    int wshuf, xshuf, oshuf;   // fragment-major layouts (OMNI_LAYOUT_*), see common.cuh frag_off
};

template <int MT, int NT, int PRO, int EPI, bool NTL, int KS>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_skinny_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [WAVES][NT*MT*4][64]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int K = a.K, N = a.N;
    const int m_base = blockIdx.y * (MT * 16);
    const int Mloc = min(a.M - m_base, MT * 16);     // valid rows of this workgroup (>= 1)

    // Operand addressing.  Row-major: lane (r, q) reads 16 B of row r at k-offset 8q, k-step stride 32 elements -- 64
    // separate cache accesses per wave-level load (adjacent lanes sit on different 2K-byte-strided rows: measured
    // 64.8 TCP accesses per load, texture addresser 71 % busy at M = 64).  Fragment-major (frag_off): the 16 x 32 tile
    // of one MFMA operand is 1 KB contiguous in lane order, one wave-level load = 8 full lines, k-step stride 512.
    const int nsteps = K >> 5;                                    // k-steps of 32
    const int wstep = a.wshuf ? 512 : 32;
    const int xstep = a.xshuf ? 512 : 32;
    const uint16_t* wrow[NT];
    {
        int row0[NT];
        if (EPI == OMNI_EPI_SILU_MUL) {
            // tiles [0, NT/2) = gate rows, tiles [NT/2, NT) = the matching up rows (W = [gate | up], N = inter)
            const int n0 = blockIdx.x * 16 * (NT / 2);
#pragma unroll
            for (int j = 0; j < NT / 2; ++j) {
                row0[j] = n0 + j * 16;
                row0[NT / 2 + j] = N + n0 + j * 16;
            }
        } else {
            const int n0 = blockIdx.x * 16 * NT;
#pragma unroll
            for (int j = 0; j < NT; ++j) row0[j] = n0 + j * 16;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
            wrow[j] = a.wshuf ? a.W + ((size_t)(row0[j] >> 4) * nsteps) * 512 + lane * 8
                              : a.W + (size_t)(row0[j] + r) * K + 8 * q;
    }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ntw = (nsteps - wave + GEMM_WAVES - 1) / GEMM_WAVES; // k-steps of this wave (may be 0)

    if (PRO == 1) {
        // ---------------- fused residual-add + RMSNorm prologue -----------------
        u32x4 wf[RN_MAX_STEPS][NT];
#pragma unroll
        for (int s = 0; s < RN_MAX_STEPS; ++s)
            if (s < ntw) {
                const int k0 = (wave + s * GEMM_WAVES) * wstep;
#pragma unroll
                for (int j = 0; j < NT; ++j) wf[s][j] = NTL ? ld16_nt(wrow[j] + k0) : ld16(wrow[j] + k0);
            }
        u32x4 xr[MT][RN_MAX_STEPS];
        float ss[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int m = i * 16 + r;
            m = m_base + (m < Mloc ? m : Mloc - 1);
            const uint16_t* rp = a.resid + (size_t)m * K + 8 * q;
            const uint16_t* dp = a.delta ? a.delta + (size_t)m * K + 8 * q : nullptr;
            float s2 = 0.f;
#pragma unroll
            for (int s = 0; s < RN_MAX_STEPS; ++s)
                if (s < ntw) {
                    const int k0 = (wave + s * GEMM_WAVES) << 5;
                    u32x4 v = ld16(rp + k0);
                    if (dp) {
                        const u32x4 d = ld16(dp + k0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = pack_bf2(bf_lo(v[e]) + bf_lo(d[e]), bf_hi(v[e]) + bf_hi(d[e]));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = bf_lo(v[e]), hi = bf_hi(v[e]);
                        s2 = fmaf(lo, lo, s2);
                        s2 = fmaf(hi, hi, s2);
                    }
                    xr[i][s] = v;
                }
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            ss[i] = s2;
        }
        // row sums across the 8 waves: lds[wave][MT*16]
        if (q == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i) lds[wave * (MT * 16) + i * 16 + r] = ss[i];
        }
        __syncthreads();
        float rstd[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < GEMM_WAVES; ++w) t += lds[w * (MT * 16) + i * 16 + r];
            rstd[i] = 1.0f / sqrtf(t / (float)K + a.eps);
        }
        __syncthreads();      // lds is reused by the epilogue
        const bool writer = blockIdx.x == 0;
#pragma unroll
        for (int s = 0; s < RN_MAX_STEPS; ++s)
            if (s < ntw) {
                const int k0 = (wave + s * GEMM_WAVES) << 5;
                const u32x4 nw = ld16(a.norm_w + k0 + 8 * q);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int ml = i * 16 + r;
                    const bool row_ok = ml < Mloc;
                    const size_t off = (size_t)(m_base + ml) * K + k0 + 8 * q;
                    u32x4 v = xr[i][s];
                    if (writer && row_ok && a.resid_out) *reinterpret_cast<u32x4*>(a.resid_out + off) = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = bf_lo(nw[e]) * bfround(bf_lo(v[e]) * rstd[i]);
                        const float hi = bf_hi(nw[e]) * bfround(bf_hi(v[e]) * rstd[i]);
                        v[e] = pack_bf2(lo, hi);
                    }
                    if (writer && row_ok && a.normed_out) *reinterpret_cast<u32x4*>(a.normed_out + off) = v;
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j][i] = mfma16(wf[s][j], v, acc[j][i]);
                }
            }
    } else {
        // ---------------- plain x operand, K known at compile time (KS = k-steps per wave = K / 256) --------------
        // Fully static schedule: every W load of the wave goes out first (KS KB in flight), x fragments run XW k-steps
        // ahead in a register ring; no runtime guard anywhere, so hipcc keeps counted vmcnt waits.
        if constexpr (KS > 0) {
            const uint16_t* xr_[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                int m = i * 16 + r;
                m = m_base + (m < Mloc ? m : Mloc - 1);
                xr_[i] = a.xshuf ? a.x + ((size_t)((m_base >> 4) + i) * nsteps) * 512 + lane * 8 + wave * xstep
                                 : a.x + (size_t)m * a.ldx + 8 * q + wave * xstep;
            }
            constexpr int XW = (MT == 4) ? 4 : 8;
            constexpr int XWe = XW < KS ? XW : KS;
            u32x4 Wr[KS][NT], X[XWe][MT];
#pragma unroll
            for (int t = 0; t < KS; ++t)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    Wr[t][j] = NTL ? ld16_nt(wrow[j] + (wave + t * GEMM_WAVES) * wstep) : ld16(wrow[j] + (wave + t * GEMM_WAVES) * wstep);
#pragma unroll
            for (int t = 0; t < XWe; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i) X[t][i] = ld16(xr_[i] + t * GEMM_WAVES * xstep);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < KS; ++t) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = mfma16(Wr[t][j], X[t % XWe][i], acc[j][i]);
                if (t + XWe < KS) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) X[t % XWe][i] = ld16(xr_[i] + (t + XWe) * GEMM_WAVES * xstep);
                }
            }
        } else {
        // ---------------- plain x operand: deep W prefetch ring, x one step ahead -----------------
        // Every wave keeps DEPTH k-steps (DEPTH KB) of W outstanding (Little's law at ~3 us loaded HBM latency);
        // x is L2-resident and fetched one k-step ahead.  (Measured alternatives, scripts/bench_ops.py: issuing a
        // whole chunk of x fragments up front -- 218 VGPRs -- was 10-40 % slower; see DESIGN.md.)
        const uint16_t* xrow[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int m = i * 16 + r;
            m = m_base + (m < Mloc ? m : Mloc - 1);   // rows past M: valid address, result discarded
            xrow[i] = a.xshuf ? a.x + ((size_t)((m_base >> 4) + i) * nsteps) * 512 + lane * 8
                              : a.x + (size_t)m * a.ldx + 8 * q;
        }
        constexpr int DEPTH = (NT == 1) ? 16 : (NT == 2 ? 8 : 4);
        u32x4 Wr[DEPTH][NT];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d < ntw) {
                const int k0 = (wave + d * GEMM_WAVES) * wstep;
#pragma unroll
                for (int j = 0; j < NT; ++j) Wr[d][j] = NTL ? ld16_nt(wrow[j] + k0) : ld16(wrow[j] + k0);
            }
        u32x4 X[2][MT];
        if (ntw > 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i) X[0][i] = ld16(xrow[i] + wave * xstep);
        }
        for (int t0 = 0; t0 < ntw; t0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int t = t0 + d;
                if (t < ntw) {
                    if (t + 1 < ntw) {
                        const int k1 = (wave + (t + 1) * GEMM_WAVES) * xstep;
#pragma unroll
                        for (int i = 0; i < MT; ++i) X[(d + 1) & 1][i] = ld16(xrow[i] + k1);
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[j][i] = mfma16(Wr[d][j], X[d & 1][i], acc[j][i]);
                    if (t + DEPTH < ntw) {
                        const int k2 = (wave + (t + DEPTH) * GEMM_WAVES) * wstep;
#pragma unroll
                        for (int j = 0; j < NT; ++j) Wr[d][j] = NTL ? ld16_nt(wrow[j] + k2) : ld16(wrow[j] + k2);
                    }
                }
            }
        }
        }   // KS == 0
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
    constexpr int NTO = (EPI == OMNI_EPI_SILU_MUL) ? NT / 2 : NT;
    constexpr int ITEMS = NTO * MT * 64;
    for (int it = threadIdx.x; it < ITEMS; it += GEMM_THREADS) {
        const int l = it & 63;
        const int t = it >> 6;
        const int i = t % MT, j = t / MT;
        const int ml = i * 16 + (l & 15);
        if (ml >= Mloc) continue;
        const int m = m_base + ml;
        float v[4], v2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float sum = 0.f, sum2 = 0.f;
#pragma unroll
            for (int w = 0; w < GEMM_WAVES; ++w) {
                sum += lds[(w * E + (j * MT + i) * 4 + g) * 64 + l];
                if (EPI == OMNI_EPI_SILU_MUL) sum2 += lds[(w * E + ((NT / 2 + j) * MT + i) * 4 + g) * 64 + l];
            }
            v[g] = sum;
            v2[g] = sum2;
        }
        if (EPI == OMNI_EPI_SILU_MUL) {
            const int n = blockIdx.x * 16 * (NT / 2) + j * 16 + 4 * (l >> 4);
            float o[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // HF: down(silu(gate(x)) * up(x)) with bf16 tensors: each op rounds to bf16
                const float gt = bfround(v[g]);
                const float up = bfround(v2[g]);
                const float sl = bfround(gt / (1.0f + expf(-gt)));
                o[g] = sl * up;
            }
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + (a.oshuf ? frag_off(m, n, N) : (size_t)m * N + n)) =
                make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
        } else {
            const int n = blockIdx.x * 16 * NT + j * 16 + 4 * (l >> 4);
            if (a.bias) {
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] += bf2f(a.bias[n + g]);
            }
            if (EPI == OMNI_EPI_BF16) {
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + (a.oshuf ? frag_off(m, n, N) : (size_t)m * N + n)) =
                    make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            } else {
                float4 o;
                float* po = reinterpret_cast<float*>(&o);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float y = (EPI == OMNI_EPI_F32_BF16RND) ? bfround(v[g]) : v[g];
                    if (a.mask && !a.mask[n + g]) y = -INFINITY;
                    po[g] = y;
                }
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + (size_t)m * N + n) = o;
            }
        }
    }
}

static int g_gemm_silu_nt4 = 0;   // 32+32-column SiLU tiles: measured slower than 16+16 once the layouts are fragment-major
static int g_gemm_nt = 1, g_gemm_rn = 0, g_gemm_wgs = 256, g_gemm_static = 0;   // static-K schedule: same time, +20 % fetch (profiles/r01_pmc_gemm_traffic.csv)
extern "C" void omni_debug_set(int nt, int rn, int wgs) { g_gemm_nt = nt & 1; g_gemm_static = (nt >> 1) & 1; g_gemm_silu_nt4 = !((nt >> 2) & 1); g_gemm_rn = rn; g_gemm_wgs = wgs; }

template <int MT, int NT, int PRO, int EPI, int KS>
static int launch_gemm_ks(const GemmArgs& a, int m_splits, hipStream_t st) {
    const int groups = (EPI == OMNI_EPI_SILU_MUL) ? a.N / (8 * NT) : a.N / (16 * NT);
    size_t lds = (size_t)GEMM_WAVES * NT * MT * 4 * 64 * sizeof(float);
    if (lds > 65536) {   // NT = 4, MT = 4: 128 KB of the CU's 160 KB (one workgroup per CU)
        static bool done_t = false, done_f = false;
        if (!done_t) { (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<MT, NT, PRO, EPI, true, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); done_t = true; }
        if (!done_f) { (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<MT, NT, PRO, EPI, false, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); done_f = true; }
    }
    // non-temporal W loads only when each W byte is read by exactly one workgroup (no m-split)
    if (m_splits == 1 && g_gemm_nt)
        hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, PRO, EPI, true, KS>), dim3(groups, m_splits), dim3(GEMM_THREADS), lds, st, a);
    else
        hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, PRO, EPI, false, KS>), dim3(groups, m_splits), dim3(GEMM_THREADS), lds, st, a);
    OMNI_CHECK_LAUNCH("omni_gemm_bf16");
    return OMNI_OK;
}

template <int MT, int NT, int PRO, int EPI>
static int launch_gemm(const GemmArgs& a, int m_splits, hipStream_t st) {
    if (PRO == 0 && g_gemm_static && a.K % 256 == 0) {
        switch (a.K / 256) {       // hidden / intermediate sizes of the talker shapes (and their TP shards)
            case 4: return launch_gemm_ks<MT, NT, PRO, EPI, (PRO == 0 ? 4 : 0)>(a, m_splits, st);
            case 8: return launch_gemm_ks<MT, NT, PRO, EPI, (PRO == 0 ? 8 : 0)>(a, m_splits, st);
            case 12: return launch_gemm_ks<MT, NT, PRO, EPI, (PRO == 0 ? 12 : 0)>(a, m_splits, st);
            case 24: return launch_gemm_ks<MT, NT, PRO, EPI, (PRO == 0 ? (MT * NT <= 2 ? 24 : 0) : 0)>(a, m_splits, st);
            default: break;
        }
    }
    return launch_gemm_ks<MT, NT, PRO, EPI, 0>(a, m_splits, st);
}

template <int NT, int PRO, int EPI>
static int dispatch_mt(const GemmArgs& a, hipStream_t st) {
    // m-tiles per workgroup: split M over grid.y until the grid has >= ~256 workgroups
    const int mt_total = (a.M + 15) / 16;
    const int groups = (EPI == OMNI_EPI_SILU_MUL) ? a.N / (8 * NT) : a.N / (16 * NT);
    int splits = (g_gemm_wgs + groups - 1) / groups;
    if (splits > mt_total) splits = mt_total;
    if (splits < 1) splits = 1;
    int mt = (mt_total + splits - 1) / splits;       // 1..4
    if (mt == 3) mt = 4;
    splits = (mt_total + mt - 1) / mt;
    if (mt == 1) return launch_gemm<1, NT, PRO, EPI>(a, splits, st);
    if (mt == 2) return launch_gemm<2, NT, PRO, EPI>(a, splits, st);
    return launch_gemm<4, NT, PRO, EPI>(a, splits, st);
}

static int check_common(const GemmArgs& a) {
    OMNI_CHECK_ARG(a.W && a.out, "omni_gemm_bf16: null pointer");
    OMNI_CHECK_ARG(a.M >= 1 && a.M <= 64, "omni_gemm_bf16: M=%d outside 1..64", a.M);
    OMNI_CHECK_ARG(a.N > 0 && a.N % 16 == 0, "omni_gemm_bf16: N=%d not a multiple of 16", a.N);
    OMNI_CHECK_ARG(a.K > 0 && a.K % 32 == 0, "omni_gemm_bf16: K=%d not a multiple of 32", a.K);
    return OMNI_OK;
}

template <int PRO>
static int dispatch_epi(const GemmArgs& a, int epilogue, hipStream_t st) {
    switch (epilogue) {
        case OMNI_EPI_BF16:
            OMNI_CHECK_ARG(a.mask == nullptr, "omni_gemm_bf16: mask needs an fp32 epilogue");
            return dispatch_mt<1, PRO, OMNI_EPI_BF16>(a, st);
        case OMNI_EPI_SILU_MUL:
            OMNI_CHECK_ARG(a.bias == nullptr && a.mask == nullptr, "omni_gemm_bf16: silu_mul takes no bias/mask");
            if (g_gemm_silu_nt4 && a.N % 32 == 0 && a.M > 32 && a.N / 32 >= 160)
                return dispatch_mt<4, PRO, OMNI_EPI_SILU_MUL>(a, st);
            return dispatch_mt<2, PRO, OMNI_EPI_SILU_MUL>(a, st);
        case OMNI_EPI_F32:
            return dispatch_mt<1, PRO, OMNI_EPI_F32>(a, st);
        case OMNI_EPI_F32_BF16RND:
            return dispatch_mt<1, PRO, OMNI_EPI_F32_BF16RND>(a, st);
        default:
            omni_set_error("omni_gemm_bf16: unknown epilogue %d", epilogue);
            return OMNI_EINVAL;
    }
}

extern "C" int omni_gemm_bf16_ex(const void* x, int ldx, const void* w, const void* bias, void* out, int M, int N,
                                 int K, int epilogue, const uint8_t* mask, int layout, void* stream) {
    GemmArgs a{};
    a.wshuf = (layout & OMNI_LAYOUT_W_FRAG) != 0;
    a.xshuf = (layout & OMNI_LAYOUT_X_FRAG) != 0;
    a.oshuf = (layout & OMNI_LAYOUT_OUT_FRAG) != 0;
    OMNI_CHECK_ARG(!a.oshuf || epilogue == OMNI_EPI_BF16 || epilogue == OMNI_EPI_SILU_MUL,
                   "omni_gemm_bf16: fragment-major output needs a bf16 epilogue");
    OMNI_CHECK_ARG(!a.oshuf || N % 32 == 0, "omni_gemm_bf16: fragment-major output needs N %% 32 == 0");
    a.x = (const uint16_t*)x; a.ldx = ldx; a.W = (const uint16_t*)w; a.bias = (const uint16_t*)bias; a.out = out;
    a.M = M; a.N = N; a.K = K; a.mask = mask;
    int rc = check_common(a);
    if (rc != OMNI_OK) return rc;
    OMNI_CHECK_ARG(x, "omni_gemm_bf16: null x");
    OMNI_CHECK_ARG(a.xshuf || (ldx >= K && ldx % 8 == 0), "omni_gemm_bf16: ldx=%d (need >= K, multiple of 8)", ldx);
    return dispatch_epi<0>(a, epilogue, (hipStream_t)stream);
}

extern "C" int omni_gemm_bf16(const void* x, int ldx, const void* w, const void* bias, void* out, int M, int N,
                              int K, int epilogue, const uint8_t* mask, void* stream) {
    return omni_gemm_bf16_ex(x, ldx, w, bias, out, M, N, K, epilogue, mask, 0, stream);
}

// the fused prologue multiplies the activation traffic by the number of workgroups (every workgroup re-reads
// resid + delta: 2 x 64 x K x 2 B): measured 2-3x SLOWER than a separate norm launch at B = 64 -> off by default
static bool rn_shape_ok(int K) { return K % 256 == 0 && K <= 32 * GEMM_WAVES * RN_MAX_STEPS; }
bool k_gemm_rn_supported(int K) { return g_gemm_rn && rn_shape_ok(K); }

extern "C" int omni_gemm_resid_norm(const void* resid, const void* delta, void* resid_out, const void* norm_w, float eps,
                                    void* normed_out, const void* w, const void* bias, void* out, int M, int N, int K,
                                    int epilogue, const uint8_t* mask, void* stream) {
    GemmArgs a{};
    a.W = (const uint16_t*)w; a.bias = (const uint16_t*)bias; a.out = out;
    a.M = M; a.N = N; a.K = K; a.mask = mask;
    a.resid = (const uint16_t*)resid; a.resid_out = (uint16_t*)resid_out; a.delta = (const uint16_t*)delta; a.norm_w = (const uint16_t*)norm_w; a.eps = eps;
    a.normed_out = (uint16_t*)normed_out;
    int rc = check_common(a);
    if (rc != OMNI_OK) return rc;
    OMNI_CHECK_ARG(resid && norm_w, "omni_gemm_resid_norm: null pointer");
    OMNI_CHECK_ARG(resid_out != resid && (normed_out == nullptr || normed_out != resid),
                   "omni_gemm_resid_norm: outputs must not alias resid (other workgroups still read it)");
    OMNI_CHECK_ARG(rn_shape_ok(K), "omni_gemm_resid_norm: K=%d unsupported (multiple of 256, <= 2048)", K);
    return dispatch_epi<1>(a, epilogue, (hipStream_t)stream);
}
