// Large-M bf16 GEMM on the matrix cores: out[M, N] = epilogue(x[M, K] . W[N, K]^T), fp32 accumulate.
//
// The prefill-side counterpart of gemm.hip's weight-streaming skinny kernel: at M = 512 ... 600 k rows the op is MFMA-bound, so a
// workgroup owns a 256-row x BN-column output tile, stages 64-deep slices of both operands in LDS and reuses every fragment
// 4 - 8 times from registers.  Built for two callers:
//   * talker prefill (engine.prefill_native): the four per-layer GEMMs on the SAME fragment-major weights the decode step
//     streams (OMNI_LAYOUT_W_FRAG; gate_up in its 8-row interleave with SiLU(gate) * up fused) -- weights are stored once;
//   * Code2Wav decoder (codec.hip / code2wav.py): activations are TIME-major [T, C], so a causal dilated conv1d is this GEMM
//     over overlapping row windows of x -- K = taps x C_in split into `seg_len`-wide segments, segment s of output row m
//     reading x row m + row_off + s * seg_rows -- with no im2col buffer, and a transposed conv of stride s / kernel 2s is
//     the GEMM [T, 2 C_in] x [2 C_in, s * C_out] whose row-major output IS the up-sampled [T * s, C_out] signal.
//
// Data movement.  Both operands reach LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per wave instruction, no VGPR
// staging): W fragments are contiguous 1 KB runs of the fragment-major matrix; x fragments are gathered by the per-lane
// SOURCE address (lane (c, q) fetches 16 B of row c at k-offset 8q), which leaves the LDS image fragment-major too --
// every ds_read_b128 is lane-linear, conflict-free, no swizzle.  Rows outside [0, x_rows) (the causal left padding, the
// M tail) and k-steps past K are out of range of the buffer descriptor and read as zero.  Two LDS buffers; the loads of
// slice t + 1 are issued before the MFMAs of slice t, one counted wait + barrier per slice.
// MFMA v_mfma_f32_16x16x32_bf16 with W as the A operand: D[n][m], lane holds m = l & 15, n = 4 (l >> 4) + reg.
// Epilogue: y = (acc + bias) -> GELU -> * scale (+ fp32 residual) in fp32, transposed through LDS (one m-tile per pass) so that global
// stores are whole row pieces; on the way out SiLU(gate) * up for the interleaved gate_up layout, and up to three stores of y: fp32 (a residual
// STREAM kept in fp32: snake's sin(alpha x) turns a bf16 ulp of x ~ 16 into a phase error of 0.1 rad), bf16 (the next GEMM's
// operand) and bf16(snake(y)) = y + inv_beta * sin^2(alpha * y) -- the activation in front of the NEXT conv, so that no
// stand-alone activation pass runs over the 24 kHz-rate tensors.
// Workgroups are dealt to the 8 XCDs in contiguous chunks of the (n block, m block) order, m fastest: the workgroups that
// share an L2 share a W panel.
#include "common.cuh"
#include "kernels.h"

#define TG_WAVES 8
#define TG_THREADS (TG_WAVES * 64)
#define TG_OOB 0x80000000u           // byte offset beyond any descriptor (num_records < 2^31): reads as zero
#define TG_LDS_MAX (160 * 1024)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

struct TileArgs {
    const uint16_t* x; int64_t x_rows; int ldx;
    int seg_len, seg_rows, row_off;
    const uint16_t* W; const float* bias; const float* scale;
    const float* resid; int ldr;
    float* out_f32; int ldf;
    uint16_t* out; int ldo;
    uint16_t* out2; int ldo2; const float* snake_alpha; const float* snake_inv_beta;
    int M, N, K;
    int act;                         // OMNI_TILE_ACT_*
    int mblocks, nblocks, band;
    // grouped (batched) launch: blockIdx.y = group; group g reads x rows [g * gx_rows, + its row count), the matrix
    // W + g * gw_elems, and writes output rows from g * gout_rows.  group_rows (device, optional) = live rows per group:
    // tiles past it leave at once, rows past it read as zero and are not stored
    long long gx_rows, gw_elems, gout_rows;
    const int32_t* group_rows;
};

__device__ __forceinline__ f32x4 tg_mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float tg_gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

__device__ __forceinline__ bool tile_plain(const TileArgs& a) {
    return !a.bias && !a.scale && a.act != OMNI_TILE_ACT_GELU && !a.resid && !a.out_f32 && !a.out2 && a.out;
}

// Epilogue of a tile kernel: `acc[j][i]` = this wave's WN n-tiles x WM m-tiles (D[n][m] fragments), first row mw0, first column nw0; `lds` is dead.
// PLAIN = no bias, scale, GELU, residual, fp32 or snake output (the talker prefill's four GEMMs): the same values (acc + 0 and * 1 change
// nothing) without ~20 wave-uniform branches per pass -- they, not the stores, were most of an 11 k-cycle epilogue (profiles/r05_gemm_pp_stamps.txt).
template <int WN, int WM, bool GU8, bool PLAIN>
__device__ __forceinline__ void tile_epilogue(const TileArgs& a, f32x4 (&acc)[WN][WM], uint8_t* lds, int wave, int lane, int mw0, int nw0) {
    const int c = lane & 15, q = lane >> 4;
    // ---- epilogue.  Register side: y = act(acc + bias) * scale in fp32, transposed through this wave's LDS image in fp32, ONE m-tile
    // (16 rows x the wave's WN * 16 columns) per pass.  Row side: 8 columns per lane, a row's lanes cover WN * 64 contiguous bytes
    // of an fp32 output (whole 128-byte lines); + fp32 residual (the pass's residual loads are issued before its image is written:
    // the epilogue of a residual GEMM is an HBM phase, not a chain of load -> add -> store round trips), then up to three stores of
    // the SAME fp32 value: fp32 (the residual stream), bf16 (the next GEMM's operand), bf16(snake(y)) (the next conv's operand).
    // (Two images per wave, pass p + 1's written before pass p's is read back, measured SLOWER: 5.9 k -> 8.5 k cycles per epilogue.)
    constexpr int IPITCH = WN * 64 + 16;                  // bytes per image row
    constexpr int PER_ROW = GU8 ? WN : WN * 2;            // lane items per image row: a [8 gate | 8 up] tile, or 8 columns
    constexpr int RIT = (16 * PER_ROW + 63) / 64;         // row-side iterations per pass
    uint8_t* img = lds + wave * (16 * IPITCH);
    f32x4 bq[WN], sq[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = nw0 + j * 16 + 4 * q;
        bq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        sq[j] = f32x4{1.f, 1.f, 1.f, 1.f};
        if (!PLAIN && n < a.N) {
            if (a.bias) bq[j] = *reinterpret_cast<const f32x4*>(a.bias + n);
            if (a.scale) sq[j] = *reinterpret_cast<const f32x4*>(a.scale + n);
        }
    }
    const bool gelu = !PLAIN && a.act == OMNI_TILE_ACT_GELU;
    auto write_image = [&](const int pass) {
        if (PLAIN) {
#pragma unroll
            for (int j = 0; j < WN; ++j) *reinterpret_cast<f32x4*>(img + c * IPITCH + (j * 16 + 4 * q) * 4) = acc[j][pass];
        } else if (gelu) {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = tg_gelu(acc[j][pass][e] + bq[j][e]) * sq[j][e];
                *reinterpret_cast<f32x4*>(img + c * IPITCH + (j * 16 + 4 * q) * 4) = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (acc[j][pass][e] + bq[j][e]) * sq[j][e];
                *reinterpret_cast<f32x4*>(img + c * IPITCH + (j * 16 + 4 * q) * 4) = v;
            }
        }
    };
    f32x4 rv[RIT][2];
#pragma unroll
    for (int pass = 0; pass < WM; ++pass) {
        const int mp0 = mw0 + pass * 16;
        if (!PLAIN && !GU8 && a.resid) {
#pragma unroll
            for (int it = 0; it < RIT; ++it) {
                const int idx = lane + 64 * it;
                const int row = idx / PER_ROW, ch = idx - row * PER_ROW;
                const int m = mp0 + row, n = nw0 + ch * 8;
                rv[it][0] = rv[it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < 16 * PER_ROW && m < a.M && n < a.N) {
                    const float* rp = a.resid + (size_t)m * a.ldr + n;
                    rv[it][0] = *reinterpret_cast<const f32x4*>(rp);
                    rv[it][1] = *reinterpret_cast<const f32x4*>(rp + 4);
                }
            }
        }
        write_image(pass);
        __builtin_amdgcn_wave_barrier();
        if (GU8) {
            // tile j of a row = [8 gate | 8 up] -> 8 act columns at n / 2: bf16(bf16(SiLU(bf16 gate)) * bf16 up), the rounding points of
            // torch's F.silu(g) * u on bf16 tensors (omni_silu_mul, oracle silu_mul)
#pragma unroll
            for (int it = 0; it < RIT; ++it) {
                const int idx = lane + 64 * it;
                const int row = idx / PER_ROW, j = idx - row * PER_ROW;
                const int m = mp0 + row, n = nw0 + j * 16;
                if (idx >= 16 * PER_ROW || m >= a.M || n >= a.N) continue;
                const float* gp = reinterpret_cast<const float*>(img + row * IPITCH + j * 64);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g0 = bfround(gp[2 * e]), g1 = bfround(gp[2 * e + 1]);
                    const float u0 = bfround(gp[8 + 2 * e]), u1 = bfround(gp[8 + 2 * e + 1]);
                    const float s0 = bfround(g0 / (1.0f + expf(-g0))), s1 = bfround(g1 / (1.0f + expf(-g1)));   // torch: F.silu(bf16) is bf16
                    o[e] = pack_bf2(s0 * u0, s1 * u1);
                }
                *reinterpret_cast<u32x4*>(a.out + (size_t)m * a.ldo + (n >> 1)) = o;
            }
        } else {
#pragma unroll
            for (int it = 0; it < RIT; ++it) {
                const int idx = lane + 64 * it;
                const int row = idx / PER_ROW, ch = idx - row * PER_ROW;
                const int m = mp0 + row, n = nw0 + ch * 8;
                if (idx >= 16 * PER_ROW || m >= a.M || n >= a.N) continue;
                f32x4 y0 = *reinterpret_cast<const f32x4*>(img + row * IPITCH + ch * 32);
                f32x4 y1 = *reinterpret_cast<const f32x4*>(img + row * IPITCH + ch * 32 + 16);
                if (PLAIN) {
                    u32x4 o;
                    o[0] = pack_bf2(y0[0], y0[1]); o[1] = pack_bf2(y0[2], y0[3]);
                    o[2] = pack_bf2(y1[0], y1[1]); o[3] = pack_bf2(y1[2], y1[3]);
                    *reinterpret_cast<u32x4*>(a.out + (size_t)m * a.ldo + n) = o;
                    continue;
                }
                if (a.resid) {
                    y0 += rv[it][0];
                    y1 += rv[it][1];
                }
                if (a.out_f32) {
                    float* op = a.out_f32 + (size_t)m * a.ldf + n;
                    *reinterpret_cast<f32x4*>(op) = y0;
                    *reinterpret_cast<f32x4*>(op + 4) = y1;
                }
                if (a.out) {
                    u32x4 o;
                    o[0] = pack_bf2(y0[0], y0[1]); o[1] = pack_bf2(y0[2], y0[3]);
                    o[2] = pack_bf2(y1[0], y1[1]); o[3] = pack_bf2(y1[2], y1[3]);
                    *reinterpret_cast<u32x4*>(a.out + (size_t)m * a.ldo + n) = o;
                }
                if (a.out2) {
                    const f32x4 al0 = *reinterpret_cast<const f32x4*>(a.snake_alpha + n), al1 = *reinterpret_cast<const f32x4*>(a.snake_alpha + n + 4);
                    const f32x4 ib0 = *reinterpret_cast<const f32x4*>(a.snake_inv_beta + n), ib1 = *reinterpret_cast<const f32x4*>(a.snake_inv_beta + n + 4);
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
                    *reinterpret_cast<u32x4*>(a.out2 + (size_t)m * a.ldo2 + n) = o;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Geometry of one instantiation: 8 waves (two per SIMD: while one issues its LDS-DMA pieces -- 60 - 180 cycles each -- or waits
// for fragments, the other feeds the matrix pipe) as WAVES_N x (8 / WAVES_N); a wave owns WN n-tiles x WM m-tiles of 16 x 16
// (acc = WN * WM * 4 of its 256 registers); BN = WAVES_N * WN * 16, BM = (8 / WAVES_N) * WM * 16.
template <int WAVES_N, int WN, int WM>
struct TileGeom {
    static constexpr int WAVES_M = TG_WAVES / WAVES_N;
    static constexpr int NTILES = WAVES_N * WN, MTILES = WAVES_M * WM;
    static constexpr int BN = NTILES * 16, BM = MTILES * 16;
    static constexpr int SLICE = (NTILES + MTILES) * 1024;                   // one 32-deep slice of both operands
    static constexpr int NLW = (NTILES + TG_WAVES - 1) / TG_WAVES, NLX = (MTILES + TG_WAVES - 1) / TG_WAVES;   // LDS-DMA per wave and slice
    static constexpr bool W_RAGGED = NTILES % TG_WAVES != 0;                 // some waves issue a dummy W load (uniform counts)
    static constexpr bool X_RAGGED = MTILES % TG_WAVES != 0;                 // likewise for x (the 64-row tile of the small-M geometry)
    static constexpr bool RAGGED = W_RAGGED || X_RAGGED;
    static constexpr int NBUF = (TG_LDS_MAX - (RAGGED ? TG_WAVES * 1024 : 0)) / SLICE >= 4 ? 4 : 3;
    static constexpr int EPI_BYTES = TG_WAVES * 16 * (WN * 64 + 16);           // fp32 image of one m-tile x the wave's columns
    static constexpr int RING_BYTES = NBUF * SLICE + (RAGGED ? TG_WAVES * 1024 : 0);
    static constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
    static_assert(WN % 2 == 0 && LDS_BYTES <= TG_LDS_MAX, "geometry");
};

template <int WAVES_N, int WN, int WM, bool GU8>
__global__ __launch_bounds__(TG_THREADS) void gemm_tile_kernel(const TileArgs a_in) {
    using G = TileGeom<WAVES_N, WN, WM>;
    TileArgs a = a_in;
    long long x_lo = 0, x_hi = a.x_rows;                                      // rows of x this launch may read
    if (gridDim.y > 1 || a.group_rows) {
        const int grp = blockIdx.y;
        if (a.group_rows) a.M = min(a.M, a.group_rows[grp]);
        x_lo = grp * a.gx_rows;
        x_hi = min(x_lo + a.M, a.x_rows);
        a.W += (size_t)grp * a.gw_elems;
        const size_t orow = (size_t)grp * a.gout_rows;
        if (a.out) a.out += orow * a.ldo;
        if (a.out2) a.out2 += orow * a.ldo2;
        if (a.out_f32) a.out_f32 += orow * a.ldf;
        if (a.resid) a.resid += orow * a.ldr;
    }
    constexpr int NTILES = G::NTILES, SLICE = G::SLICE, NLW = G::NLW, NLX = G::NLX, NBUF = G::NBUF;
    constexpr int INFLIGHT = NBUF - 2;                                        // slices still landing when slice t + 1 is awaited
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int c = lane & 15, q = lane >> 4;

    // XCD-aware tile order (bijective for any grid size): the hardware deals workgroup ids round-robin over the 8 XCDs
    int tile;
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3, per = nwg >> 3, rem = nwg & 7;
        tile = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    }
    // tile order: bands of `band` m blocks, n-major inside a band, so that an XCD's contiguous chunk of ~nwg / 8 tiles is a compact
    // 2-D block (~sqrt x sqrt) of the tile grid: it touches the fewest distinct W and x panels (L2 misses go to the MALL / HBM)
    int n_blk, m_blk;
    {
        const int per_band = a.band * a.nblocks;
        const int b = tile / per_band, r = tile - b * per_band;
        const int h = min(a.band, a.mblocks - b * a.band);            // rows of this (possibly last, shorter) band
        n_blk = r / h;
        m_blk = b * a.band + (r - n_blk * h);
    }
    const int m0 = m_blk * G::BM, n0 = n_blk * G::BN;
    if (m0 >= a.M) return;                                                    // a group with fewer rows than the launch's M (uniform)
    const int nsteps = a.K >> 5;                                              // 32-deep slices

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (int)((size_t)a.N * a.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(a.x_rows * a.ldx * 2), 0x00020000);

    // ---- staging.  Slice image: W tile f at f KB, x tile g at (NTILES + g) KB.  Wave w issues W tiles w + 8 i (i < NLW; a wave
    // without an i-th tile loads zeros into a spare KB so that every wave has the same count in flight) and x tiles w + 8 i.
    unsigned wbase[NLW];
    int xrow[NLX], xbyte[NLX];
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int f = wave + TG_WAVES * i;
        const int n16 = (n0 >> 4) + f;
        wbase[i] = (f < NTILES && n16 * 16 < a.N) ? (unsigned)((size_t)n16 * nsteps * 1024 + lane * 16) : TG_OOB;
    }
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
        const int g = wave + TG_WAVES * i;
        xrow[i] = (int)x_lo + m0 + g * 16 + c + a.row_off;
        xbyte[i] = (xrow[i] * a.ldx + 8 * q) * 2;        // wraps out of the descriptor for rows < 0 (never used then)
    }
    int seg = 0, cs = 0;                                  // segment walk of the NEXT slice to stage: k = seg * seg_len + cs
    const int seg_row_bytes = a.seg_rows * a.ldx * 2;
    auto stage = [&](int slice) {                         // slices are staged in order, one call each
        const int buf = slice % NBUF;
        const bool live = slice < nsteps;
        const int shift = seg * a.seg_rows, add = seg * seg_row_bytes + cs * 2;
        uint8_t* base = lds + buf * SLICE + wave * 1024;
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            const bool mine = !G::W_RAGGED || wave + TG_WAVES * i < NTILES;
            const unsigned off = (live && mine && wbase[i] != TG_OOB) ? wbase[i] + (unsigned)slice * 1024u : TG_OOB;
            uint8_t* dst = mine ? base + i * (TG_WAVES * 1024) : lds + NBUF * SLICE + wave * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)dst, 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NLX; ++i) {
            const bool mine = !G::X_RAGGED || wave + TG_WAVES * i < G::MTILES;
            const int r = xrow[i] + shift;
            const unsigned off = (live && mine && r >= (int)x_lo && r < (int)x_hi) ? (unsigned)(xbyte[i] + add) : TG_OOB;
            uint8_t* dst = mine ? base + (NTILES + i * TG_WAVES) * 1024 : lds + NBUF * SLICE + wave * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)dst, 16, off, 0, 0, 0);
        }
        cs += 32;
        if (cs >= a.seg_len) { cs = 0; ++seg; }
    };

    f32x4 acc[WN][WM];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop.  Per slice a wave reads WM x fragments + WN W fragments and issues WN * WM MFMAs, in two halves of the W
    // tiles.  The wait for slice t + 1 and the workgroup barrier sit BETWEEN the halves of slice t; right after it the wave
    // reads the x fragments and the first W half of slice t + 1 into the other register set, so that LDS latency hides behind
    // the second half's MFMAs, and the second W half is read at the top of the next slice behind the first half's MFMAs.
    // NBUF-deep ring, slice t + NBUF - 1 staged at the top of slice t: NBUF - 2 slices stay in flight across every barrier
    // (counted vmcnt, never 0 inside the loop).  lgkmcnt(0) before the barrier retires this wave's reads of the buffer that
    // the next top-of-slice staging overwrites.
    constexpr int HW = WN / 2;
    auto lds_w = [&](int slice, int j) {
        return *reinterpret_cast<const u32x4*>(lds + (slice % NBUF) * SLICE + (wn * WN + j) * 1024 + lane * 16);
    };
    auto lds_x = [&](int slice, int i) {
        return *reinterpret_cast<const u32x4*>(lds + (slice % NBUF) * SLICE + (NTILES + wm * WM + i) * 1024 + lane * 16);
    };
    u32x4 xa[WM], xb[WM], w0[HW], w1[HW];
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s) stage(s);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(INFLIGHT * (NLW + NLX)) : "memory");
#pragma unroll
    for (int i = 0; i < WM; ++i) xa[i] = lds_x(0, i);
#pragma unroll
    for (int j = 0; j < HW; ++j) w0[j] = lds_w(0, j);

#define TG_SLICE(T, XCUR, XNEXT)                                                                             \
    {                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < HW; ++j) w1[j] = lds_w((T), HW + j);                           \
        stage((T) + NBUF - 1);                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int j = 0; j < HW; ++j)                                                       \
            _Pragma("unroll") for (int i = 0; i < WM; ++i) acc[j][i] = tg_mfma(w0[j], XCUR[i], acc[j][i]);   \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(INFLIGHT * (NLW + NLX)) : "memory"); \
        _Pragma("unroll") for (int i = 0; i < WM; ++i) XNEXT[i] = lds_x((T) + 1, i);                         \
        _Pragma("unroll") for (int j = 0; j < HW; ++j) w0[j] = lds_w((T) + 1, j);                            \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int j = 0; j < HW; ++j)                                                       \
            _Pragma("unroll") for (int i = 0; i < WM; ++i) acc[HW + j][i] = tg_mfma(w1[j], XCUR[i], acc[HW + j][i]); \
        __builtin_amdgcn_s_setprio(0);                                                                       \
    }
    for (int t = 0; t < nsteps; t += 2) {                // an odd slice count runs one slice of zeros (staged out of range)
        TG_SLICE(t, xa, xb)
        TG_SLICE(t + 1, xb, xa)
    }
#undef TG_SLICE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the ring is dead: its tail loads (zeros) landed

    if (tile_plain(a)) tile_epilogue<WN, WM, GU8, true>(a, acc, lds, wave, lane, m0 + wm * WM * 16, n0 + wn * WN * 16);
    else tile_epilogue<WN, WM, GU8, false>(a, acc, lds, wave, lane, m0 + wm * WM * 16, n0 + wn * WN * 16);
}

// ---- the two-group ("ping-pong") form of the 256-column tile.  Same images, same MFMA, same accumulation order as gemm_tile_kernel
// (bit-identical output); what differs is WHEN a wave does what.  The eight waves are two groups of four, one wave of each group per SIMD:
// group g owns the tile's m-tiles [MW g, + MW) (16 rows each; BM = 32 MW), wave column wc = wave & 3 owns n-tiles [4 wc, + 4).  A K-tile
// (64 deep: two 32-deep steps) is two phases of [12 LDS reads + 4 LDS-DMA pieces | barrier | 24 - 32 MFMAs | barrier], and group 1 runs ONE
// barrier interval behind group 0: while one wave of a SIMD issues its loads (an LDS-DMA piece costs its issuer 60 - 180 cycles) and waits
// for its fragments, the other one owns the matrix pipe.  In gemm_tile_kernel both waves of a SIMD load at the same time and then want the
// pipe at the same time.  Phase 2 t: [x low (t), W high (t) | x low (t) x all four n-tiles]; phase 2 t + 1: [x high (t), W low (t + 1) |
// x high (t) x all four] (the low W fragments live in two register sets).
// Staging: a K-tile's image is four 16 KB half-tiles (x: the low m-tiles of both groups | W: n-tiles 0-1 of every wave column | W: n-tiles
// 2-3 | x: the high m-tiles; a wave without a real piece loads zeros into a spare KB: uniform vmcnt), two 64 KB buffers.  Each phase stages
// the two half-tiles that the phase after the next one reads, and waits -- vmcnt(4) behind its own four pieces, in front of its first
// barrier -- for the two that the NEXT phase reads.  Order of a region's life: read in phase p (the reads retire at that wave's lgkmcnt(0)
// behind the phase's first barrier) -> re-staged in phase p + 2 -> waited for in p + 3 -> read in p + 4.  With the groups one interval
// apart every one of these edges still has a barrier that BOTH parties passed in between.
// Measured (scripts/probes/gemm_pp_stamps.hip, profiles/r05_gemm_pp_stamps.txt): 2 400 - 2 600 cycles per K-tile against 2 048 of MFMA at
// full rate (gemm_tile_kernel: ~2 900); four phases of 12 - 16 MFMAs: +1.5 %; the closing barrier 4 or 8 MFMAs before a section's end
// (both groups on the pipe for a moment): +9 %; unequal groups (240 / 208 rows): slower than the next taller equal pair.
//
// -DPP_STAMPS (the probe only): 1 / 2 = the shader clock at the section boundaries of K-tiles 8 ... 23 of one workgroup (first wave of each
// group; 2 adds a stamp between the loads and the first barrier), kept in LDS behind the ring (a global store would count in vmcnt);
// 3 = shader clock and the 100 MHz clock at kernel entry, loop start, loop end, kernel end, nothing inside the loop.
#ifdef PP_STAMPS
__device__ unsigned long long g_pp_stamps[2][1024];
__device__ int g_pp_stamp_wg;
#if PP_STAMPS == 3
#define PP_T()
#define PP_C(k) do { if (st_wave) { g_pp_stamps[grp][1 + 2 * (k)] = __builtin_amdgcn_s_memtime(); g_pp_stamps[grp][2 + 2 * (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define PP_T() do { if (st_on) { st_buf[st_n] = __builtin_amdgcn_s_memtime(); ++st_n; } } while (0)
#define PP_C(k)
#endif
#if PP_STAMPS == 2
#define PP_T1() PP_T()
#else
#define PP_T1()
#endif
#else
#define PP_T()
#define PP_T1()
#define PP_C(k)
#endif

template <int MW, bool GU8>
__global__ __launch_bounds__(TG_THREADS) void gemm_tile_pp_kernel(const TileArgs a_in) {
    const TileArgs& a = a_in;
    constexpr int KT_BYTES = 64 * 1024, HALF = 16 * 1024, SPARE = 2 * KT_BYTES;
    constexpr int LO = (MW + 1) / 2, HI = MW - LO;            // low / high m-tiles of a wave
    static_assert(MW >= 2 && MW <= 8, "geometry");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int c = lane & 15, q = lane >> 4;
#ifdef PP_STAMPS
    const bool st_wave = (int)blockIdx.x == g_pp_stamp_wg && wc == 0 && lane == 0;
#endif
    PP_C(0);
    int tile;
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3, per = nwg >> 3, rem = nwg & 7;
        tile = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    }
    int n_blk, m_blk;
    {
        const int per_band = a.band * a.nblocks;
        const int b = tile / per_band, r = tile - b * per_band;
        const int h = min(a.band, a.mblocks - b * a.band);
        n_blk = r / h;
        m_blk = b * a.band + (r - n_blk * h);
    }
    const int m0 = m_blk * (32 * MW), n0 = n_blk * 256;
    const int nsteps = a.K >> 5, nkt = (nsteps + 1) >> 1;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (int)((size_t)a.N * a.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(a.x_rows * a.ldx * 2), 0x00020000);

    // staging: piece e (0 / 1) of a half-tile = fragment f = wave + 8 e of its 16 = (slot t = f >> 1, k-step f & 1).  Slot t of an x half is
    // the t-th m-tile of [group 0's half | group 1's half]; slot t of a W half is n-tile 4 (t >> 1) + (t & 1) (+ 2 in the high half).
    unsigned wofs[2][2], xofs[2][2];
    bool xreal[2][2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int f = wave + 8 * e, t = f >> 1, ks = f & 1;
            const int n16 = (n0 >> 4) + 4 * (t >> 1) + (t & 1) + 2 * hi;
            wofs[hi][e] = n16 * 16 < a.N ? (unsigned)(((size_t)n16 * nsteps + ks) * 1024 + lane * 16) : TG_OOB;
            const int cnt = hi ? HI : LO;                     // per group
            const int mt = (t < cnt ? 0 : MW) + (hi ? LO : 0) + (t < cnt ? t : t - cnt);
            const int row = m0 + mt * 16 + c;
            xreal[hi][e] = t < 2 * cnt;
            xofs[hi][e] = (t < 2 * cnt && row < a.M && row < a.x_rows) ? (unsigned)(((size_t)row * a.ldx + ks * 32 + 8 * q) * 2) : TG_OOB;
        }
    auto stage = [&](int kt, int h, int buf) {                // h: 0 x low, 1 W low, 2 W high, 3 x high
        const bool live = kt < nkt;
        const bool is_w = h == 1 || h == 2;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const unsigned base = is_w ? wofs[h == 2][e] : xofs[h == 3][e];
            // an odd count of 32-deep steps: the last K-tile's second step lies past the row's K (x: the next row's data; W: the next n-tile's)
            const bool tail = ((wave + 8 * e) & 1) && 2 * kt + 1 >= nsteps;
            const unsigned off = (live && !tail && base != TG_OOB) ? base + (unsigned)kt * (is_w ? 2048u : 128u) : TG_OOB;
            const bool real = is_w || xreal[h == 3][e];
            uint8_t* dst = real ? lds + buf * KT_BYTES + h * HALF + (wave + 8 * e) * 1024 : lds + SPARE + wave * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(is_w ? rw : rx, (lds_void_t*)dst, 16, off, 0, 0, 0);
        }
    };
    const int t_lo = grp ? LO : 0, t_hi = grp ? HI : 0;       // this group's first slot in the x halves
    auto lds_x = [&](int buf, int hi, int i, int ks) {
        return *reinterpret_cast<const u32x4*>(lds + buf * KT_BYTES + (hi ? 3 : 0) * HALF + ((((hi ? t_hi : t_lo) + i) * 2 + ks) * 1024) + lane * 16);
    };
    auto lds_w = [&](int buf, int hi, int j, int ks) {
        return *reinterpret_cast<const u32x4*>(lds + buf * KT_BYTES + (hi ? 2 : 1) * HALF + (((wc * 2 + j) * 2 + ks) * 1024) + lane * 16);
    };

    // prologue: K-tile 0 whole and K-tile 1's low W -- what phase 0 expects to be in flight behind what it reads
    stage(0, 1, 0); stage(0, 0, 0); stage(0, 2, 0); stage(0, 3, 0); stage(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    if (grp) asm volatile("s_barrier" ::: "memory");          // group 1 starts one interval late

    f32x4 acc[4][MW];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MW; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 xr[LO][2], wl[2][2], wl2[2][2], wh[2][2];
#ifdef PP_STAMPS
    unsigned long long* st_buf = reinterpret_cast<unsigned long long*>(lds + SPARE + 8192 + grp * 4096);
    int st_n = 0;
    bool st_on = false;
#endif

#define PP_MFMA(WL, XCNT, XHI)                                                                                           \
    PP_T1();                                                                                                             \
    asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    PP_T();                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                     \
        _Pragma("unroll") for (int i = 0; i < (XCNT); ++i) {                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
                acc[j][(XHI) * LO + i] = tg_mfma(WL[j][ks], xr[i][ks], acc[j][(XHI) * LO + i]);                          \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
                acc[2 + j][(XHI) * LO + i] = tg_mfma(wh[j][ks], xr[i][ks], acc[2 + j][(XHI) * LO + i]);                  \
        }                                                                                                                \
    __builtin_amdgcn_s_setprio(0);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    PP_T();                                                                                                              \
    asm volatile("s_barrier" ::: "memory");                                                                              \
    PP_T();
#define PP_KTILE(KT, BUF, WL, WLNEXT)                                                                                    \
    {                                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < LO; ++i)                                                                   \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) xr[i][ks] = lds_x(BUF, 0, i, ks);                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) wh[j][ks] = lds_w(BUF, 1, j, ks);                           \
        stage((KT) + 1, 0, (BUF) ^ 1);                                                                                   \
        stage((KT) + 1, 2, (BUF) ^ 1);                                                                                   \
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        PP_MFMA(WL, LO, 0)                                                                                               \
        _Pragma("unroll") for (int i = 0; i < HI; ++i)                                                                   \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) xr[i][ks] = lds_x(BUF, 1, i, ks);                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) WLNEXT[j][ks] = lds_w((BUF) ^ 1, 0, j, ks);                 \
        stage((KT) + 1, 3, (BUF) ^ 1);                                                                                   \
        stage((KT) + 2, 1, BUF);                                                                                         \
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        PP_MFMA(WL, HI, 1)                                                                                               \
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wl[j][ks] = lds_w(0, 0, j, ks);
    PP_C(1);
    for (int u = 0; u < nkt; u += 2) {                        // an odd K-tile count runs one K-tile of zeros (staged out of range)
#ifdef PP_STAMPS
        st_on = st_wave && u >= 8 && u < 24;
        if (u == 8) PP_T();
#endif
        PP_KTILE(u, 0, wl, wl2)
        PP_KTILE(u + 1, 1, wl2, wl)
    }
#undef PP_KTILE
#undef PP_MFMA
    PP_C(2);
    if (!grp) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#if defined(PP_STAMPS) && PP_STAMPS != 3
    if (st_wave)
        for (int i = 0; i < st_n && i < 1023; ++i) g_pp_stamps[grp][1 + i] = st_buf[i];
    if (st_wave) g_pp_stamps[grp][0] = st_n;
#endif
    if (tile_plain(a)) tile_epilogue<4, MW, GU8, true>(a, acc, lds, wave, lane, m0 + grp * MW * 16, n0 + wc * 64);
    else tile_epilogue<4, MW, GU8, false>(a, acc, lds, wave, lane, m0 + grp * MW * 16, n0 + wc * 64);
#if defined(PP_STAMPS) && PP_STAMPS == 3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_C(3);
#endif
}

template <int MW, bool GU8>
static int launch_tile_pp(TileArgs a, hipStream_t st) {
    constexpr int BM = 32 * MW;
    a.mblocks = (a.M + BM - 1) / BM;
    a.nblocks = a.N / 256;
    {
        const int per_xcd = (a.mblocks * a.nblocks + 7) / 8;
        int band = 1;
        while (band * band < per_xcd) ++band;
        a.band = band > a.mblocks ? a.mblocks : band;
    }
#ifdef PP_STAMPS
    constexpr int LDS_BYTES = 128 * 1024 + TG_WAVES * 1024 + 8192 + 8192;
#else
    constexpr int LDS_BYTES = 128 * 1024 + TG_WAVES * 1024;
#endif
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tile_pp_kernel<MW, GU8>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) { omni_set_error("omni_gemm_tile: LDS attribute: %s", hipGetErrorString(e)); return OMNI_EHIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tile_pp_kernel<MW, GU8>), dim3(a.mblocks * a.nblocks), dim3(TG_THREADS), LDS_BYTES, st, a);
    OMNI_CHECK_LAUNCH("omni_gemm_tile");
    return OMNI_OK;
}

template <bool GU8>
static int launch_tile_pp_rows(int bm, const TileArgs& a, hipStream_t st) {
    if (bm == 256) return launch_tile_pp<8, GU8>(a, st);
    if (bm == 224) return launch_tile_pp<7, GU8>(a, st);
    return launch_tile_pp<6, GU8>(a, st);
}

template <int WAVES_N, int WN, int WM, bool GU8>
static int launch_tile(TileArgs a, hipStream_t st, int groups = 1) {
    using G = TileGeom<WAVES_N, WN, WM>;
    a.mblocks = (a.M + G::BM - 1) / G::BM;
    a.nblocks = (a.N + G::BN - 1) / G::BN;
    {
        const int per_xcd = (a.mblocks * a.nblocks + 7) / 8;
        int band = 1;
        while (band * band < per_xcd) ++band;                         // ~sqrt(tiles per XCD)
        a.band = band < 1 ? 1 : (band > a.mblocks ? a.mblocks : band);
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tile_kernel<WAVES_N, WN, WM, GU8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) { omni_set_error("omni_gemm_tile: LDS attribute: %s", hipGetErrorString(e)); return OMNI_EHIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tile_kernel<WAVES_N, WN, WM, GU8>), dim3(a.mblocks * a.nblocks, groups), dim3(TG_THREADS), G::LDS_BYTES, st, a);
    OMNI_CHECK_LAUNCH("omni_gemm_tile");
    return OMNI_OK;
}

extern "C" int omni_gemm_tile(const omni_tile_gemm* g, void* stream) {
    OMNI_CHECK_ARG(g && g->x && g->w && (g->out || g->out2 || g->out_f32), "omni_gemm_tile: null argument");
    OMNI_CHECK_ARG(g->M > 0 && g->N > 0 && g->N % 16 == 0 && g->K > 0 && g->K % 32 == 0, "omni_gemm_tile: M=%d N=%d K=%d (N %% 16, K %% 32)", g->M, g->N, g->K);
    const int seg_len = g->seg_len > 0 ? g->seg_len : g->K;
    OMNI_CHECK_ARG(seg_len % 32 == 0 && g->K % seg_len == 0, "omni_gemm_tile: seg_len=%d must divide K=%d and be a multiple of 32", seg_len, g->K);
    OMNI_CHECK_ARG(g->ldx % 8 == 0 && g->ldx >= seg_len, "omni_gemm_tile: ldx=%d (>= seg_len, multiple of 8)", g->ldx);
    OMNI_CHECK_ARG(g->x_rows > 0 && g->x_rows * (int64_t)g->ldx * 2 < (int64_t)TG_OOB, "omni_gemm_tile: x of %lld rows x %d exceeds the 2 GB descriptor", (long long)g->x_rows, g->ldx);
    OMNI_CHECK_ARG((int64_t)g->N * g->K * 2 < (int64_t)TG_OOB, "omni_gemm_tile: W exceeds the 2 GB descriptor");
    const bool gu8 = g->act == OMNI_TILE_ACT_SILU_MUL_GU8;
    OMNI_CHECK_ARG(g->act == OMNI_TILE_ACT_NONE || g->act == OMNI_TILE_ACT_GELU || gu8, "omni_gemm_tile: act=%d", g->act);
    OMNI_CHECK_ARG(g->tile_hint >= 0 && g->tile_hint <= 7, "omni_gemm_tile: tile_hint=%d", g->tile_hint);
    OMNI_CHECK_ARG(g->tile_hint < 3 || g->N % 256 == 0, "omni_gemm_tile: tile_hint=%d needs N %% 256 == 0", g->tile_hint);
    OMNI_CHECK_ARG(!gu8 || (g->out && !g->resid && !g->out2 && !g->out_f32), "omni_gemm_tile: SiLU-mul takes out only");
    OMNI_CHECK_ARG(!g->out2 || (g->snake_alpha && g->snake_inv_beta), "omni_gemm_tile: out2 needs the snake parameters");
    const int nout = gu8 ? g->N / 2 : g->N;
    OMNI_CHECK_ARG((!g->out || (g->ldo >= nout && g->ldo % 8 == 0)) && (!g->resid || (g->ldr >= nout && g->ldr % 4 == 0)) &&
                   (!g->out2 || (g->ldo2 >= nout && g->ldo2 % 8 == 0)) && (!g->out_f32 || (g->ldf >= nout && g->ldf % 4 == 0)),
                   "omni_gemm_tile: ldo / ldr / ldo2 / ldf (>= N; multiples of 8 (bf16) / 4 (fp32))");
    TileArgs a;
    a.x = (const uint16_t*)g->x; a.x_rows = g->x_rows; a.ldx = g->ldx;
    a.seg_len = seg_len; a.seg_rows = g->seg_rows; a.row_off = g->row_off;
    a.W = (const uint16_t*)g->w; a.bias = g->bias; a.scale = g->scale;
    a.resid = g->resid; a.ldr = g->ldr;
    a.out_f32 = g->out_f32; a.ldf = g->ldf;
    a.out = (uint16_t*)g->out; a.ldo = g->ldo;
    a.out2 = (uint16_t*)g->out2; a.ldo2 = g->ldo2; a.snake_alpha = g->snake_alpha; a.snake_inv_beta = g->snake_inv_beta;
    a.M = g->M; a.N = g->N; a.K = g->K; a.act = g->act;
    const int groups = g->groups > 1 ? g->groups : 1;
    OMNI_CHECK_ARG(groups <= 65535, "omni_gemm_tile: groups=%d", g->groups);
    OMNI_CHECK_ARG(groups == 1 || (seg_len == g->K && g->row_off == 0 && g->x_group_rows >= 0 && g->w_group_elems >= 0 && g->out_group_rows >= 0),
                   "omni_gemm_tile: a grouped launch is a plain GEMM per group (no conv window) with non-negative group strides");
    OMNI_CHECK_ARG(groups == 1 || (int64_t)groups * g->w_group_elems * 2 + (int64_t)g->N * g->K * 2 < ((int64_t)1 << 40), "omni_gemm_tile: W groups");
    a.gx_rows = groups > 1 ? g->x_group_rows : 0; a.gw_elems = groups > 1 ? g->w_group_elems : 0; a.gout_rows = groups > 1 ? g->out_group_rows : 0;
    a.group_rows = g->group_rows;
    hipStream_t st = (hipStream_t)stream;
    const int N = g->N;
    // <waves along n, n tiles per wave, m tiles per wave>: 256 x 256 | 192 x 256 | 128 x 512 | 96 x 512 output tiles, and 128 x 64
    // for problems whose big-tile grid would leave most of the 256 CUs idle (the Code2Wav transformer / ConvNeXt at a few hundred
    // frames, streaming chunks): there the K loop's latency is the cost, not MFMA throughput
    {
        const int big_n = N % 256 == 0 ? 256 : (N % 192 == 0 ? 192 : (N % 128 == 0 ? 128 : (N % 96 == 0 ? 96 : (N > 128 ? 256 : 128))));
        const int big_m = big_n >= 192 ? 256 : 512;
        const long long big_tiles = (long long)((N + big_n - 1) / big_n) * ((g->M + big_m - 1) / big_m) * groups;
        const bool small = g->tile_hint == 2 || (g->tile_hint == 0 && big_tiles < 96 && g->N >= 64);
        if (small) return gu8 ? launch_tile<4, 2, 2, true>(a, st, groups) : launch_tile<4, 2, 2, false>(a, st, groups);
    }
    // 256-column tiles come in three heights.  One workgroup per CU (the ring takes most of the LDS), so a grid runs in rounds of
    // 256 tiles and a round costs ~ the tile's rows: 6.4 k prompt tokens x 2048 columns are 208 tiles of 256 rows (one round, 48 CUs
    // idle) or 232 tiles of 224 rows (one round, 12.5 % shorter).  Same accumulation order in every geometry: bit-identical results.
    // A plain GEMM (no conv window, no groups) runs them in the two-group form (gemm_tile_pp_kernel: 5 - 8 % faster at every height).
    if (N % 256 == 0) {
        const bool pp_ok = groups == 1 && !g->group_rows && seg_len == g->K && g->row_off == 0;
        OMNI_CHECK_ARG(g->tile_hint < 5 || pp_ok, "omni_gemm_tile: tile_hint=%d (two-group tile) takes a plain GEMM", g->tile_hint);
        int bm = 256;
        if (g->tile_hint == 3 || g->tile_hint == 6) bm = 224;
        else if (g->tile_hint == 4 || g->tile_hint == 7) bm = 192;
        else if (g->tile_hint == 0 && groups == 1) {
            long long best = -1;
            for (int cand : {256, 224, 192}) {
                const long long tiles = (long long)((g->M + cand - 1) / cand) * (N / 256);
                const long long cost = ((tiles + 255) / 256) * (cand + 16);
                if (best < 0 || cost < best) { best = cost; bm = cand; }
            }
        }
        if (g->tile_hint >= 5 || (g->tile_hint == 0 && pp_ok)) return gu8 ? launch_tile_pp_rows<true>(bm, a, st) : launch_tile_pp_rows<false>(bm, a, st);
        if (bm == 224) return gu8 ? launch_tile<4, 4, 7, true>(a, st, groups) : launch_tile<4, 4, 7, false>(a, st, groups);
        if (bm == 192) return gu8 ? launch_tile<4, 4, 6, true>(a, st, groups) : launch_tile<4, 4, 6, false>(a, st, groups);
        return gu8 ? launch_tile<2, 8, 4, true>(a, st, groups) : launch_tile<2, 8, 4, false>(a, st, groups);
    }
    if (gu8) return launch_tile<1, 8, 4, true>(a, st, groups);
    if (N % 192 == 0) return launch_tile<2, 6, 4, false>(a, st, groups);
    if (N % 128 == 0) return launch_tile<1, 8, 4, false>(a, st, groups);
    if (N % 96 == 0) return launch_tile<1, 6, 4, false>(a, st, groups);
    return N > 128 ? launch_tile<2, 8, 4, false>(a, st, groups) : launch_tile<1, 8, 4, false>(a, st, groups);
}
