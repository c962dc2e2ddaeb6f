// Code2Wav decoder residual unit as ONE launch (Qwen3TTSTokenizerV2DecoderDecoderResidualUnit,
// tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:726-742), for the high-rate blocks (C = 96 | 192 channels, 10^5 - 10^6 rows):
//     h <- h + conv1x1( snake2( conv7_dilated( s ) ) ),     s_next = snake_next(h)
// with s = snake1(h) in bf16 (the previous launch's second output), h the fp32 residual stream, time-major [T, C].
// As separate omni_gemm_tile launches the unit moves 16 B per element through HBM and fetches every x row from L2 once per tap
// (7 x); here
//   * the row block's x window (its BM rows + the 6 * dilation halo rows above them) is read ONCE into LDS, row-major with a 16-byte
//     row pad (pitch 2 C + 16 B: the 16 rows of an MFMA operand land on 16 different 16-byte bank groups for every tap shift), and
//     all 7 taps take their operand fragments from it by row offset;
//   * the 7-tap conv's output tile never leaves the chip: bias + snake2 -> bf16 -> LDS in operand-fragment order -> the 1x1 conv's
//     MFMAs (its C x C weight straight from L2 into registers);
//   * epilogue: + bias + h (fp32) -> h in place, bf16(snake_next(h)) for the next unit: 12 B per element of HBM traffic.
// W1 (fragment-major [C, 7 C], K index = tap * C + channel) streams through an LDS ring by LDS-DMA like omni_gemm_tile's W.
// 8 waves, a wave owns 6 n-tiles x 4 m-tiles (96 columns x 64 rows): C = 96: 1 x 8 waves, BM = 512; C = 192: 2 x 4 waves, BM = 256.
// Same k order per output element as the two-launch path (taps outer, channels inner; then the 1x1 conv's channels): bit-identical
// results (tests/test_gpu_code2wav.py).
#include "common.cuh"
#include "kernels.h"
#include "../../include/omni_codec.h"

#define RU_WAVES 8
#define RU_THREADS (RU_WAVES * 64)
#define RU_TAPS 7
#define RU_HALO_MAX 54               // 6 * dilation, dilation <= 9
#define RU_OOB 0x80000000u

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void ru_lds_t;

struct ResUnitArgs {
    const uint16_t* s; float* h; uint16_t* s_next;
    const uint16_t* w1; const float* b1; const float* a2; const float* ib2;      // conv7 + the snake behind it
    const uint16_t* w2; const float* b2; const float* an; const float* ibn;      // conv1x1 + the snake of the next consumer
    int T, dil;
};

__device__ __forceinline__ f32x4 ru_mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int C, int WM_, int NBUF_>
struct RuGeom {
    static constexpr int NT = C / 16, KS = C / 32, WAVES_N = C / 96, WAVES_M = RU_WAVES / WAVES_N, WN = 6, WM = WM_;
    static constexpr int BM = WAVES_M * WM * 16;
    static constexpr int PITCH = 2 * C + 16;                                      // window row pitch (bytes)
    static constexpr int WIN_ROWS = BM + RU_HALO_MAX;
    static constexpr int WIN_BYTES = (WIN_ROWS * PITCH + 1023) / 1024 * 1024;
    static constexpr int NSTEPS = RU_TAPS * KS;                                   // 32-deep k-steps of the 7-tap conv
    static constexpr int NLW = (NT + RU_WAVES - 1) / RU_WAVES;                    // W1 LDS-DMA pieces per wave and k-step
    static constexpr int NBUF = NBUF_;
    static constexpr int RING_BYTES = NBUF * NT * 1024;
    static constexpr int TT_BYTES = BM * C * 2;                                   // stage-2 operand (overlays the window)
    static constexpr int IPITCH = 6 * 64 + 16;                                    // epilogue image row: the wave's 96 columns x fp32 + pad
    static constexpr int EPI_BYTES = RU_WAVES * 16 * IPITCH;                      // one m-tile (16 rows) per wave and pass; overlays the window
    static constexpr int LDS_BYTES = WIN_BYTES + RING_BYTES;
    static_assert(TT_BYTES <= WIN_BYTES && EPI_BYTES <= WIN_BYTES && LDS_BYTES <= 160 * 1024, "LDS budget");
};

// <C, m tiles per wave, ring depth>: <96, 2, 2> = 256-row blocks in 76 KB of LDS and <= 128 registers: TWO workgroups per CU, so that one
// block's HBM phases (window load, stream read / write) overlap the other's MFMA phase; <192, 4, 3> = 256-row blocks, one per CU
template <int C, int WM_, int NBUF_>
__global__ __launch_bounds__(RU_THREADS, (WM_ == 2 ? 4 : 2)) void res_unit_kernel(const ResUnitArgs a) {
    using G = RuGeom<C, WM_, NBUF_>;
    constexpr int NT = G::NT, KS = G::KS, WN = G::WN, WM = G::WM, P = G::PITCH, NBUF = G::NBUF, NLW = G::NLW, NSTEPS = G::NSTEPS;
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    uint8_t* win = lds;
    uint8_t* ring = lds + G::WIN_BYTES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave % G::WAVES_N, wm = wave / G::WAVES_N;
    const int c = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * G::BM;
    const int halo = (RU_TAPS - 1) * a.dil;

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, 0, C * RU_TAPS * C * 2, 0x00020000);
    // ---- W1 ring: k-step u = tap * KS + ks; tile (n16, u) of the fragment-major matrix at (n16 * NSTEPS + u) KB
    // Wave w stages tiles (w + 8 i) mod NT: where NT is no multiple of 8 a few tiles are fetched twice (identical bytes to the same
    // LDS address) so that every wave has the same number of pieces in flight for the counted waits
    unsigned wbase[NLW];
    int wtile[NLW];
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        wtile[i] = (wave + RU_WAVES * i) % NT;
        wbase[i] = (unsigned)(wtile[i] * NSTEPS * 1024 + lane * 16);
    }
    auto stage = [&](int u) {
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            const unsigned off = u < NSTEPS ? wbase[i] + (unsigned)u * 1024u : RU_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (ru_lds_t*)(ring + (u % NBUF) * (NT * 1024) + wtile[i] * 1024), 16, off, 0, 0, 0);
        }
    };
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u) stage(u);

    // ---- x window: rows m0 - halo .. m0 + BM - 1 of s -> LDS rows 0 .. halo + BM - 1 (16-byte chunks, zero outside [0, T)); all of a
    // thread's chunks (<= WCH) are in flight before the first LDS write: one memory latency, not one per batch
    {
        constexpr int CH = C / 8;                                                 // chunks per row
        constexpr int WCH = (G::WIN_ROWS * CH + RU_THREADS - 1) / RU_THREADS;
        const int total = (halo + G::BM) * CH;
        u32x4 v[WCH];
#pragma unroll
        for (int k = 0; k < WCH; ++k) {
            const int id = k * RU_THREADS + (int)threadIdx.x;
            const int lr = id / CH, lc = id - lr * CH;
            const int g = m0 - halo + lr;
            v[k] = u32x4{0u, 0u, 0u, 0u};
            if (id < total && g >= 0 && g < a.T) v[k] = *reinterpret_cast<const u32x4*>(a.s + (size_t)g * C + lc * 8);
        }
#pragma unroll
        for (int k = 0; k < WCH; ++k) {
            const int id = k * RU_THREADS + (int)threadIdx.x;
            const int lr = id / CH, lc = id - lr * CH;
            if (id < total) *reinterpret_cast<u32x4*>(win + lr * P + lc * 16) = v[k];
        }
    }
    f32x4 acc[WN][WM];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NBUF - 2) * NLW) : "memory");      // window + k-step 0 in place

    // ---- stage 1: the 7-tap conv.  Per k-step: the wave's 4 x fragments from the window (row offset = tap * dilation), its 6 W
    // fragments from the ring, 24 MFMAs; k-step u + NBUF - 1 staged at the top, NBUF - 2 k-steps in flight across the barrier
    const uint8_t* xw = win + (wm * WM * 16 + c) * P + q * 16;
    for (int tap = 0; tap < RU_TAPS; ++tap) {
        const uint8_t* xt = xw + tap * a.dil * P;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int u = tap * KS + ks;
            stage(u + NBUF - 1);
            u32x4 xf[WM], wf[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) xf[i] = *reinterpret_cast<const u32x4*>(xt + i * 16 * P + ks * 64);
            const uint8_t* wr = ring + (u % NBUF) * (NT * 1024) + (wn * WN) * 1024 + lane * 16;
#pragma unroll
            for (int j = 0; j < WN; ++j) wf[j] = *reinterpret_cast<const u32x4*>(wr + j * 1024);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int i = 0; i < WM; ++i) acc[j][i] = ru_mfma(wf[j], xf[i], acc[j][i]);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NBUF - 2) * NLW) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's tail pieces (zeros) have landed; the barrier above retired all window reads

    // ---- stage 2 operand: t = bf16(snake2(acc + b1)) in MFMA operand-fragment order: m-tile gi, k-step n >> 5, lane (chunk of 8
    // columns) * 16 + row.  Lane (row c, q) holds columns 16 j + 4 q .. + 3 of tile j: 8 bytes of chunk 2 (j & 1) + (q >> 1)
    uint8_t* tt = lds;                                     // overlays the window: every wave passed the last barrier
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = (wn * WN + j) * 16 + 4 * q;
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.b1 + n);
        const f32x4 al = *reinterpret_cast<const f32x4*>(a.a2 + n);
        const f32x4 ib = *reinterpret_cast<const f32x4*>(a.ib2 + n);
        const int ks2 = n >> 5, chunk = ((n & 31) >> 3), sub = (n & 7) >> 2;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = acc[j][i][e] + b[e];
                const float sn = __sinf(y * al[e]);
                v[e] = y + ib[e] * sn * sn;
            }
            uint2 pk;
            pk.x = pack_bf2(v[0], v[1]);
            pk.y = pack_bf2(v[2], v[3]);
            const int gi = wm * WM + i;
            *reinterpret_cast<uint2*>(tt + (gi * KS + ks2) * 1024 + (chunk * 16 + c) * 16 + sub * 8) = pk;
            acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // epilogue geometry: one m-tile (16 rows) x the wave's 96 columns per pass: a row's 12 lanes cover 384 contiguous bytes of the fp32
    // stream (whole 128-byte lines; passes over column halves wrote 192-byte pieces: 3.0 instead of 3.8 TB/s)
    constexpr int IP = G::IPITCH;
    constexpr int EIT = 16 * 12 / 64;                      // row-side iterations per pass (8 columns per lane each)
    uint8_t* img = lds + wave * (16 * IP);
    const int mw0 = m0 + wm * WM * 16;
    const int nw0 = wn * WN * 16;
    f32x4 hv[2][EIT][2];                                   // the stream values of two passes in flight
    auto load_h = [&](int pass) {
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
            const int idx = lane + 64 * it;
            const int row = idx / 12, ch = idx - row * 12;
            const int m = mw0 + pass * 16 + row;
            hv[pass & 1][it][0] = hv[pass & 1][it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < a.T) {
                const float* hp = a.h + (size_t)m * C + nw0 + ch * 8;
                hv[pass & 1][it][0] = *reinterpret_cast<const f32x4*>(hp);
                hv[pass & 1][it][1] = *reinterpret_cast<const f32x4*>(hp + 4);
            }
        }
    };
    // ---- stage 2: the 1x1 conv: K = C; W2 fragments (tile (n16, ks) at (n16 * KS + ks) KB) straight from L2
    {
        const uint16_t* w2 = a.w2 + (size_t)(wn * WN) * KS * 512 + lane * 8;
        u32x4 wf[WN], wnx[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) wf[j] = *reinterpret_cast<const u32x4*>(w2 + (size_t)j * KS * 512);
        load_h(0);                                         // lands behind the 1x1 conv's MFMAs
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int j = 0; j < WN; ++j) wnx[j] = *reinterpret_cast<const u32x4*>(w2 + ((size_t)j * KS + ks + 1) * 512);
            }
            u32x4 xf[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) xf[i] = *reinterpret_cast<const u32x4*>(tt + ((wm * WM + i) * KS + ks) * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int i = 0; i < WM; ++i) acc[j][i] = ru_mfma(wf[j], xf[i], acc[j][i]);
            if (ks + 1 < KS) {
#pragma unroll
                for (int j = 0; j < WN; ++j) wf[j] = wnx[j];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // all waves are done reading t: the image may overlay it

    // ---- epilogue: y = acc + b2 + h (fp32) -> h; bf16(snake_next(y)) -> s_next; one m-tile per pass through a per-wave fp32 LDS image
    // [16 rows][96 columns]; pass p + 1's stream loads are issued before pass p's stores
#pragma unroll
    for (int pass = 0; pass < WM; ++pass) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.b2 + nw0 + j * 16 + 4 * q);
            *reinterpret_cast<f32x4*>(img + c * IP + (j * 16 + 4 * q) * 4) = acc[j][pass] + b;
        }
        if (pass + 1 < WM) load_h(pass + 1);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
            const int idx = lane + 64 * it;
            const int row = idx / 12, ch = idx - row * 12;
            const int m = mw0 + pass * 16 + row, n = nw0 + ch * 8;
            if (m >= a.T) continue;
            float* hp = a.h + (size_t)m * C + n;
            const f32x4 y0 = *reinterpret_cast<const f32x4*>(img + row * IP + ch * 32) + hv[pass & 1][it][0];
            const f32x4 y1 = *reinterpret_cast<const f32x4*>(img + row * IP + ch * 32 + 16) + hv[pass & 1][it][1];
            *reinterpret_cast<f32x4*>(hp) = y0;
            *reinterpret_cast<f32x4*>(hp + 4) = y1;
            const f32x4 al0 = *reinterpret_cast<const f32x4*>(a.an + n), al1 = *reinterpret_cast<const f32x4*>(a.an + n + 4);
            const f32x4 ib0 = *reinterpret_cast<const f32x4*>(a.ibn + n), ib1 = *reinterpret_cast<const f32x4*>(a.ibn + n + 4);
            f32x4 z0, z1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s0 = __sinf(y0[e] * al0[e]), s1 = __sinf(y1[e] * al1[e]);
                z0[e] = y0[e] + ib0[e] * s0 * s0;
                z1[e] = y1[e] + ib1[e] * s1 * s1;
            }
            u32x4 o;
            o[0] = pack_bf2(z0[0], z0[1]); o[1] = pack_bf2(z0[2], z0[3]);
            o[2] = pack_bf2(z1[0], z1[1]); o[3] = pack_bf2(z1[2], z1[3]);
            *reinterpret_cast<u32x4*>(a.s_next + (size_t)m * C + n) = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int C, int WM_, int NBUF_>
static int launch_unit(const ResUnitArgs& a, hipStream_t st) {
    using G = RuGeom<C, WM_, NBUF_>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(res_unit_kernel<C, WM_, NBUF_>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) { omni_set_error("omni_codec_res_unit: LDS attribute: %s", hipGetErrorString(e)); return OMNI_EHIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL((res_unit_kernel<C, WM_, NBUF_>), dim3((a.T + G::BM - 1) / G::BM), dim3(RU_THREADS), G::LDS_BYTES, st, a);
    OMNI_CHECK_LAUNCH("omni_codec_res_unit");
    return OMNI_OK;
}

extern "C" int omni_codec_res_unit_supported(int C, int taps, int dilation) {
    return (C == 96 || C == 192) && taps == RU_TAPS && dilation >= 1 && (RU_TAPS - 1) * dilation <= RU_HALO_MAX;
}

extern "C" int omni_codec_res_unit(const omni_res_unit* u, void* stream) {
    OMNI_CHECK_ARG(u && u->s && u->h && u->s_next && u->w1 && u->b1 && u->snake2_alpha && u->snake2_inv_beta && u->w2 && u->b2 &&
                   u->next_alpha && u->next_inv_beta, "omni_codec_res_unit: null argument");
    OMNI_CHECK_ARG(u->T > 0 && omni_codec_res_unit_supported(u->C, RU_TAPS, u->dilation), "omni_codec_res_unit: T=%d C=%d dilation=%d (C 96 | 192, dilation <= 9)", u->T, u->C, u->dilation);
    OMNI_CHECK_ARG(u->s != u->s_next, "omni_codec_res_unit: s_next aliases s (neighbouring row blocks read a halo of s)");
    ResUnitArgs a;
    a.s = (const uint16_t*)u->s; a.h = u->h; a.s_next = (uint16_t*)u->s_next;
    a.w1 = (const uint16_t*)u->w1; a.b1 = u->b1; a.a2 = u->snake2_alpha; a.ib2 = u->snake2_inv_beta;
    a.w2 = (const uint16_t*)u->w2; a.b2 = u->b2; a.an = u->next_alpha; a.ibn = u->next_inv_beta;
    a.T = u->T; a.dil = u->dilation;
    return u->C == 96 ? launch_unit<96, 4, 4>(a, (hipStream_t)stream) : launch_unit<192, 4, 3>(a, (hipStream_t)stream);
}
