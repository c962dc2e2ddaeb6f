// Skinny-M (M <= 64) bf16 GEMM for the decode step: out[M,N] = x[M,K] . W[N,K]^T.
//
// HBM-bound weight streaming (64 FLOP/B << MFMA ridge): W is read exactly once, straight from
// HBM into MFMA A-operand registers with non-temporal loads (each W element feeds one wave only,
// and keeping it out of L2 leaves the activations and kernel code L2-resident); x (<= 64 x K,
// L2-resident) is the B operand.  v_mfma_f32_16x16x32_bf16:
//   A lane(r = l&15, q = l>>4) = W[n0 + r][k0 + 8q .. +8)      (16 rows x 64 B per load)
//   B lane(c = l&15, q)        = x[m0 + c][k0 + 8q .. +8)
//   D[n][m]: lane holds m = l&15, n = 4*(l>>4) + reg.
// Workgroup = 8 waves = one group of NT 16-row n-tiles x MT 16-row m-tiles; grid = (n groups,
// m splits).  The 8 waves split K (k-steps interleaved w, w+8, ...); when K % 512 == 0 (every talker
// shape) a wave runs the COUNTED schedule: two k-steps of W and x in flight (four for the residual-epilogue tiles), no
// predicated load, so each MFMA group waits only for its own operands (vmcnt(N), not vmcnt(0)); other K take the generic
// ring loop.  Partial sums combine through LDS.  M <= 64 (128 with 128-row slabs: the code predictor's pair pass).
// SiLU epilogues: OMNI_EPI_SILU_MUL pairs a gate tile with its up tile ([gate | up] rows, NT even); OMNI_EPI_SILU_MUL_GU8
// takes W with gate / up rows interleaved by 8, so ONE tile finishes 8 act columns and NT = 3 fills 256 workgroups.
// fp32 accumulate, one rounding to bf16 (oracle: talker_oracle.linear / rms_norm).
//
// The norm-free residual stream (omni_gemm_resid / omni_gemm_xnorm, see include/omni_talker.h):
//   EPI_RESID  r = bf16(r + bf16(acc + bias)) in place on the fragment-major stream + this workgroup's
//              share of sum(r^2) per row (slab [n group][64]);
//   PRO_XNORM  x = norm_w * bf16(r * rstd) applied to every operand fragment as it is loaded.  The
//              normalisation is redundant across n groups (every workgroup rebuilds the rows it
//              multiplies), so PRO_XNORM launches use FEW n groups of wide tiles (NT = 4) and split M
//              across grid.y instead: VALU per workgroup = rows_of_split x K x ~5 instructions.
#include "common.cuh"
#include "kernels.h"
#include "gemm_frag.cuh"

#define GEMM_WAVES 8
#define GEMM_THREADS (GEMM_WAVES * 64)

struct GemmArgs {
    const uint16_t* x; int ldx;
    const uint16_t* W; const uint16_t* bias; void* out;
    int M, N, K;
    const uint8_t* mask;
    // EPI_RESID: resid = the stream to add into (NULL: start a new one); PRO_XNORM: norm_w / eps / normed_out (row-major copy)
    const uint16_t* resid; const uint16_t* norm_w; float eps; uint16_t* normed_out;
    int wshuf, xshuf, oshuf;   // fragment-major layouts (OMNI_LAYOUT_*), see common.cuh frag_off
    // PRO_XNORM / EPI_RESID: per-row sum-of-squares slabs [np][64 rows] fp32 (deterministic: one slab per producer workgroup)
    const float* part_in; int np_in; float* part_out;
    int pstride;               // rows per slab (64; 128 for the code predictor's two-position pass)
    float mask_fill;           // logit written where mask[n] == 0 (-inf; the Omni talker writes -1e9)
    int counted;               // use the counted (unpredicated, 2-deep) schedule when K % 512 == 0
    const int32_t* num_live;   // PRO_XNORM normed_out: rows >= *num_live are not written (NULL: all M rows)
    int dbg_stage;             // debug library only: leave the kernel after stage N (launch-cost attribution), 0 = run all
};

template <int MT, int NT, int PRO, int EPI, bool NTL>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_skinny_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [WAVES][NT*MT*4][64]
#ifdef OMNI_DEBUG_HOOKS
#define DBG_STAGE(n) if (a.dbg_stage == (n)) return
#else
#define DBG_STAGE(n)
#endif
    DBG_STAGE(1);                                                  // 1: empty kernel with this launch geometry
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
        static_assert(NT != 3 || EPI == OMNI_EPI_SILU_MUL_GU8, "NT = 3 only for the interleaved gate_up layout");
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

    // PRO_XNORM, part 1: this thread's share of the sum(r^2) slabs of the workgroup's OWN rows (row = t % rows, every
    // NCH-th slab; slabs are [n group][64 rows], so a wave-level load is contiguous) -- issued BEFORE the W ring so that
    // it returns first (loads complete in order) and the reduction overlaps the W latency.  Loads are unconditional
    // (clamped slab index): a predicated load would force hipcc to drain every outstanding load at each wait.
    // PRO 3 (round 6; the chains' DEFER form, chain_gemm.cuh): the row's rstd is NOT folded into the operand fragments -- x = bf16(w * r)
    // -- but multiplies the fp32 sums in the epilogue: the slab reduction leaves the critical path between the operand loads and the first
    // MFMA (one LDS round trip + a workgroup barrier per norm-fused GEMM).  One rounding point moves: rstd * sum(W * bf16(w r)) instead of
    // sum(W * bf16(w * bf16(r rstd))) -- one bf16 rounding per element instead of two; gated on the accuracy tests (tests/test_gpu_parity_full.py
    // three-way statement), not on the oracle's bits.  Taken for the GEMMs whose normalised rows are nobody's output (qkv, gate_up).
    constexpr bool NORM = PRO == 2 || PRO == 3, DEFER = PRO == 3;
    constexpr int XROWS = MT * 16, NCH = GEMM_THREADS / XROWS;
    typedef SlabOrder<XROWS> SO;                                                  // canonical slab order (gemm_frag.cuh): SO::PE slabs per thread cover np <= 128
    float pv[SO::PE];
    float psum = 0.f;
    if (NORM) {
        const int row = lane % XROWS, sub = lane / XROWS, ch = threadIdx.x / XROWS;
        const float* pp = a.part_in + m_base + row;
#pragma unroll
        for (int e = 0; e < SO::PE; ++e) {
            const int p = SO::index(wave, sub, e);
            pv[e] = pp[(size_t)min(p, a.np_in - 1) * a.pstride];
            if (p >= a.np_in) pv[e] = 0.f;
        }
        for (int p = ch + 128; p < a.np_in; p += NCH) psum += pp[(size_t)p * a.pstride];      // hidden > 2048 only (no chain there)
    }
    // EPI_RESID: the old residual values this thread will add into (one epilogue item per thread), fetched up front
    uint2 r_old = make_uint2(0, 0);
    if (EPI == OMNI_EPI_RESID && a.resid && threadIdx.x < MT * 64) {
        const int ml = (threadIdx.x >> 6) * 16 + (lane & 15);
        if (ml < Mloc)
            r_old = *reinterpret_cast<const uint2*>(a.resid + frag_off(m_base + ml, blockIdx.x * 16 + 4 * (lane >> 4), N));
    }

    DBG_STAGE(2);                                                  // 2: + addressing, slab / old-residual loads issued
    float rstd[MT];
    auto xnorm_rstd = [&]() {
        // part 2 (after the W ring is in flight): fixed-order reduction -> rstd of this workgroup's rows.  row = lane % XROWS:
        // fold the channels that share a wave by shuffles, the 8 waves through a private LDS area past the epilogue's
        // (ONE barrier; every wave then holds all row sums and picks its MFMA rows by shuffle)
        float s_ = SO::reduce(pv);
        if (a.np_in > 128) {                                   // the slabs past the canonical 128: folded over the row's channels as before
            if (XROWS <= 32) psum = xor32_sum(psum);
            if (XROWS <= 16) psum = xor16_sum(psum);
            s_ += psum;
        }
        float* red = lds + GEMM_WAVES * NT * MT * 4 * 64;      // past the combine slots
        red[wave * 64 + lane] = s_;
        if (DEFER) return;                                     // read behind the combine barrier, by the epilogue
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < GEMM_WAVES; ++w) t += red[w * 64 + lane];
        const float rl = 1.0f / sqrtf(t / (float)K + a.eps);
#pragma unroll
        for (int i = 0; i < MT; ++i) rstd[i] = __shfl(rl, i * 16 + r, 64);
    };

    const int ntw = (nsteps - wave + GEMM_WAVES - 1) / GEMM_WAVES; // k-steps of this wave (may be 0)

    {
        {
        // ---------------- plain x operand: deep W prefetch ring, x one step ahead -----------------
        // Every wave keeps DEPTH k-steps (DEPTH KB) of W outstanding (Little's law at ~3 us loaded HBM latency);
        // x is L2-resident and fetched one k-step ahead.  (Measured alternatives, scripts/bench_ops.py: issuing a
        // whole chunk of x fragments up front -- 218 VGPRs -- was 10-40 % slower; see DESIGN.md.)
        const uint16_t* xrow[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int m = i * 16 + r;
            m = m_base + (m < Mloc ? m : Mloc - 1);   // rows past M: valid address, result discarded
            // (fragment-major x holds ceil(M / 16) row tiles: a workgroup tile that reaches past them -- 3 row tiles run as MT = 4 -- re-reads the
            //  last one, results discarded; until round 6 it read 16 rows past the end of the caller's buffer)
            const int xt = min((m_base >> 4) + i, ((a.M + 15) >> 4) - 1);
            xrow[i] = a.xshuf ? a.x + ((size_t)xt * nsteps) * 512 + lane * 8
                              : a.x + (size_t)m * a.ldx + 8 * q;
        }
        // k-steps of W AND x in flight per wave on the counted path.  Same-box A/B of the whole step (scripts/ab.sh):
        // 1 step 4.42 ms, 2 steps 4.20, 4 steps (small tiles) 4.22, 8 / 12 steps (o / down) 4.28 -- two is enough to cover
        // the round trip; every further load issued up front only delays the first MFMA group
        // -- except the residual-epilogue GEMMs (one 16 x 16 tile, long K, all weights cold): 4 steps, 4.135 -> 4.085 ms
        constexpr int P = (EPI == OMNI_EPI_RESID) ? 4 : 2;
        constexpr bool COUNTED = true;
        if (COUNTED && a.counted && ntw > 0 && nsteps % (GEMM_WAVES * P) == 0) {
            // ---- counted schedule (K % (256 P) == 0: every wave owns a multiple of P k-steps).  No load is predicated, so
            // hipcc can count: each MFMA group waits for exactly the loads issued P steps earlier (vmcnt((P-1) * loads
            // per step)) while the younger P-1 steps stay in flight -- with a predicated refill it must wait vmcnt(0)
            // before every group, i.e. one exposed L2 round trip for x per k-step.
            u32x4 Wq[P][NT], Xq[P][MT], NWq[P];
#pragma unroll
            for (int d = 0; d < P; ++d) {
                const int ks = wave + d * GEMM_WAVES;
#pragma unroll
                for (int j = 0; j < NT; ++j) Wq[d][j] = NTL ? ld16_nt(wrow[j] + ks * wstep) : ld16(wrow[j] + ks * wstep);
#pragma unroll
                for (int i = 0; i < MT; ++i) Xq[d][i] = ld16(xrow[i] + ks * xstep);
                if (NORM) NWq[d] = ld16(a.norm_w + ks * 32 + 8 * q);
            }
            if (PRO == 2) xnorm_rstd();
            const int G = ntw / P;
            for (int g = 0; g + 1 < G; ++g) {
#pragma unroll
                for (int d = 0; d < P; ++d) {
                    if (NORM) {
#pragma unroll
                        for (int i = 0; i < MT; ++i) Xq[d][i] = DEFER ? xw_frag(Xq[d][i], NWq[d]) : xnorm_frag(Xq[d][i], NWq[d], rstd[i]);
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[j][i] = mfma16(Wq[d][j], Xq[d][i], acc[j][i]);
                    const int ks = wave + ((g + 1) * P + d) * GEMM_WAVES;
#pragma unroll
                    for (int j = 0; j < NT; ++j) Wq[d][j] = NTL ? ld16_nt(wrow[j] + ks * wstep) : ld16(wrow[j] + ks * wstep);
#pragma unroll
                    for (int i = 0; i < MT; ++i) Xq[d][i] = ld16(xrow[i] + ks * xstep);
                    if (NORM) NWq[d] = ld16(a.norm_w + ks * 32 + 8 * q);
                }
            }
#pragma unroll
            for (int d = 0; d < P; ++d) {
                if (NORM) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) Xq[d][i] = DEFER ? xw_frag(Xq[d][i], NWq[d]) : xnorm_frag(Xq[d][i], NWq[d], rstd[i]);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = mfma16(Wq[d][j], Xq[d][i], acc[j][i]);
            }
        } else {
        constexpr int DEPTH = (NT == 1) ? 8 : (NT == 2 ? 4 : 2);   // KB of W in flight per wave; deeper rings (16/8/4) measured 1 % slower in the step
        u32x4 Wr[DEPTH][NT];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d < ntw) {
                const int k0 = (wave + d * GEMM_WAVES) * wstep;
#pragma unroll
                for (int j = 0; j < NT; ++j) Wr[d][j] = NTL ? ld16_nt(wrow[j] + k0) : ld16(wrow[j] + k0);
            }
        u32x4 X[2][MT], NW[2];
        if (ntw > 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i) X[0][i] = ld16(xrow[i] + wave * xstep);
            if (NORM) NW[0] = ld16(a.norm_w + wave * 32 + 8 * q);
        }
        if (PRO == 2) xnorm_rstd();
        for (int t0 = 0; t0 < ntw; t0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int t = t0 + d;
                if (t < ntw) {
                    if (t + 1 < ntw) {
                        const int k1 = (wave + (t + 1) * GEMM_WAVES) * xstep;
#pragma unroll
                        for (int i = 0; i < MT; ++i) X[(d + 1) & 1][i] = ld16(xrow[i] + k1);
                        if (NORM) NW[(d + 1) & 1] = ld16(a.norm_w + (wave + (t + 1) * GEMM_WAVES) * 32 + 8 * q);
                    }
                    if (NORM) {
#pragma unroll
                        for (int i = 0; i < MT; ++i) X[d & 1][i] = DEFER ? xw_frag(X[d & 1][i], NW[d & 1]) : xnorm_frag(X[d & 1][i], NW[d & 1], rstd[i]);
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
        }
        }
    }

    if (DEFER) xnorm_rstd();      // PRO 3: the slab reduction behind the MFMA loop (its loads returned ahead of the operands'); read by the epilogue
    DBG_STAGE(3);                                                  // 3: + main loop (operand loads, normalisation, MFMA)
    // the normalised rows themselves (h[t+1] of the final norm) leave row-major, spread over the n groups: workgroup column x
    // re-reads the (L2-hot) fragments of k-steps x, x + gridDim.x, ... of its rows -- a second pass kept out of the main loop so
    // that its waits stay counted, and short (one k-step per wave at most) so that no workgroup column carries a tail
    if (PRO == 2 && a.normed_out != nullptr) {
        const int nlive = a.num_live ? min(*a.num_live, a.M) - m_base : Mloc;    // h[t+1] of padding rows stays untouched
        for (int ks = blockIdx.x + wave * gridDim.x; ks < nsteps; ks += GEMM_WAVES * gridDim.x) {
            const u32x4 nw = ld16(a.norm_w + ks * 32 + 8 * q);
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (i * 16 + r < min(Mloc, nlive)) {
                    const u32x4 v = ld16(a.x + ((size_t)((m_base >> 4) + i) * nsteps + ks) * 512 + lane * 8);
                    *reinterpret_cast<u32x4*>(a.normed_out + (size_t)(m_base + i * 16 + r) * K + ks * 32 + 8 * q) = xnorm_frag(v, nw, rstd[i]);
                }
        }
    }

    // ---- combine the 8 K-partials through LDS: one 16-byte slot per (wave, tile, lane) holding the lane's 4 accumulator
    // registers -- ds_write_b128 / ds_read_b128, a quarter of the LDS instructions of the former [wave][reg][lane] floats
    // (GEMM stage attribution, scripts/ab_knobs.py gemm_stage: the combine + epilogue were 0.42 ms of a 3.93 ms step)
    f32x4* lds4 = reinterpret_cast<f32x4*>(lds);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) lds4[(wave * (NT * MT) + j * MT + i) * 64 + lane] = acc[j][i];
    __syncthreads();
    DBG_STAGE(4);                                                  // 4: + LDS write of the K partials + barrier

    // item = (m-tile i, [n-tile j], lane l): 4 consecutive n (reg 0..3) of one row m
    // OMNI_EPI_SILU_MUL_GU8: every 16-row W tile = 8 gate rows (D lanes 0..31) + the 8 matching up rows (lanes 32..63)
    constexpr bool GU8 = EPI == OMNI_EPI_SILU_MUL_GU8;
    constexpr bool SILU = EPI == OMNI_EPI_SILU_MUL || GU8;
    constexpr int NTO = (EPI == OMNI_EPI_SILU_MUL) ? NT / 2 : NT;
    constexpr int LN = GU8 ? 32 : 64;
    constexpr int ITEMS = NTO * MT * LN;
    for (int it = threadIdx.x; it < ITEMS; it += GEMM_THREADS) {
        const int l = it % LN;
        const int t = it / LN;
        const int i = t % MT, j = t / MT;
        const int ml = i * 16 + (l & 15);
        if (ml >= Mloc) continue;
        const int m = m_base + ml;
        float v[4], v2[4];
        {
            f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f}, sum2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < GEMM_WAVES; ++w) {              // fixed order: run-to-run deterministic
                sum += lds4[(w * (NT * MT) + j * MT + i) * 64 + l];
                if (EPI == OMNI_EPI_SILU_MUL) sum2 += lds4[(w * (NT * MT) + (NT / 2 + j) * MT + i) * 64 + l];
                if (GU8) sum2 += lds4[(w * (NT * MT) + j * MT + i) * 64 + l + 32];
            }
            if (DEFER) {
                // rstd of this item's row: the eight wave partials of the slab reduction, in wave order (the combine barrier covered them)
                const float* red = lds + GEMM_WAVES * NT * MT * 4 * 64;
                float tsum = 0.f;
#pragma unroll
                for (int w = 0; w < GEMM_WAVES; ++w) tsum += red[w * 64 + ml];
                const float rl = 1.0f / sqrtf(tsum / (float)K + a.eps);
                sum *= rl;
                if (SILU) sum2 *= rl;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) { v[g] = sum[g]; v2[g] = sum2[g]; }
        }
        if (SILU) {
            const int n = GU8 ? (blockIdx.x * NT + j) * 8 + 4 * (l >> 4) : blockIdx.x * 16 * (NT / 2) + j * 16 + 4 * (l >> 4);
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
            if (EPI == OMNI_EPI_RESID) {
                // r = bf16(r + bf16(acc)) in place on the fragment-major residual stream (every element has exactly one
                // owner thread), plus this workgroup's 16-column share of sum(r^2) per row for the consumer's RMSNorm
                static_assert(EPI != OMNI_EPI_RESID || NT == 1, "EPI_RESID: one n-tile per workgroup");
                uint16_t* rp = reinterpret_cast<uint16_t*>(a.out) + frag_off(m, n, N);
                float rv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) rv[g] = bfround(v[g]);
                if (a.resid) {     // item == thread (ITEMS = MT * 64 <= 256): r_old was fetched for exactly this (m, n)
                    rv[0] = bfround(bf_lo(r_old.x) + rv[0]); rv[1] = bfround(bf_hi(r_old.x) + rv[1]);
                    rv[2] = bfround(bf_lo(r_old.y) + rv[2]); rv[3] = bfround(bf_hi(r_old.y) + rv[3]);
                }
                *reinterpret_cast<uint2*>(rp) = make_uint2(pack_bf2(rv[0], rv[1]), pack_bf2(rv[2], rv[3]));
                float ss = sq4_sum(rv[0], rv[1], rv[2], rv[3]);
                ss = xor32_sum(xor16_sum(ss));       // the 4 lanes (l & 15) + 16 * {0..3} hold the 16 columns of row m
                if (l < 16) a.part_out[blockIdx.x * a.pstride + m] = ss;   // slabs [n group][pstride rows]
            } else if (EPI == OMNI_EPI_BF16) {
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + (a.oshuf ? frag_off(m, n, N) : (size_t)m * N + n)) =
                    make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            } else {
                float4 o;
                float* po = reinterpret_cast<float*>(&o);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float y = (EPI == OMNI_EPI_F32_BF16RND) ? bfround(v[g]) : v[g];
                    if (a.mask && !a.mask[n + g]) y = a.mask_fill;
                    po[g] = y;
                }
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + (size_t)m * N + n) = o;
            }
        }
    }
}


OMNI_KNOB g_gemm_nt = 1, g_gemm_wgs = 256, g_tile_nt = 0, g_tile_mt = 0, g_gemm_counted = 1, g_gemm_stage = 0, g_gemm_defer = 1;
#ifdef OMNI_DEBUG_HOOKS
extern "C" void omni_debug_gemm_defer(int on) { g_gemm_defer = on; }          // 0: round 5's exact rstd in every norm-fused GEMM of the launch path (the chains have their own knobs)
extern "C" void omni_debug_gemm_stage(int stage) { g_gemm_stage = stage; }   // leave every GEMM kernel after stage N (timing only: results are garbage)
extern "C" void omni_debug_set(int nt, int rn, int wgs) { g_gemm_counted = rn ? 0 : 1; g_gemm_nt = nt & 1; g_gemm_wgs = wgs; }   // rn != 0: generic schedule
extern "C" void omni_debug_tile(int nt, int mt) { g_tile_nt = nt; g_tile_mt = mt; }   // 0 = policy default
#endif

template <int MT, int NT, int PRO, int EPI>
static int launch_gemm(const GemmArgs& a_in, int m_splits, hipStream_t st) {
    GemmArgs a = a_in;
    a.dbg_stage = g_gemm_stage;
    const int groups = (EPI == OMNI_EPI_SILU_MUL || EPI == OMNI_EPI_SILU_MUL_GU8) ? a.N / (8 * NT) : a.N / (16 * NT);
    size_t lds = (size_t)GEMM_WAVES * NT * MT * 4 * 64 * sizeof(float) + (PRO >= 2 ? GEMM_WAVES * 64 * sizeof(float) : 0);
    if (lds > 65536) {   // NT = 4, MT = 4: 128 KB of the CU's 160 KB (one workgroup per CU)
        static bool done_t = false, done_f = false;
        if (!done_t) { (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<MT, NT, PRO, EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); done_t = true; }
        if (!done_f) { (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<MT, NT, PRO, EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); done_f = true; }
    }
    // non-temporal W loads only when each W byte is read by exactly one workgroup (no m-split)
    if (m_splits == 1 && g_gemm_nt)
        hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, PRO, EPI, true>), dim3(groups, m_splits), dim3(GEMM_THREADS), lds, st, a);
    else
        hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, PRO, EPI, false>), dim3(groups, m_splits), dim3(GEMM_THREADS), lds, st, a);
    OMNI_CHECK_LAUNCH("omni_gemm_bf16");
    return OMNI_OK;
}

// Tile policy.  Plain GEMMs: narrow n groups (NT = 1, SiLU 16 + 16 columns), M split over grid.y until the grid has
// ~256 workgroups.  PRO_XNORM: wide n groups (NT = 4) and as many M splits as that allows -- the normalisation work
// of a workgroup is proportional to the rows it owns, and is repeated once per n group.
static void pick_tile(int pro, int epi, const GemmArgs& a, int* nt_out, int* mt_out, int* splits_out) {
    const int mt_total = (a.M + 15) / 16;
    const bool gu8 = epi == OMNI_EPI_SILU_MUL_GU8;
    const bool silu = epi == OMNI_EPI_SILU_MUL || gu8;
    const int cols = silu ? a.N / 8 : a.N / 16;               // n groups at the narrowest tile
    int nt = silu ? 2 : 1;
    if (pro == 2) {
        // measured at M = 64 (scripts/bench_tiles.py): 64+ groups of NT = 4, else NT = 2
        nt = (cols % 4 == 0 && cols / 4 >= 64) ? 4 : ((cols % 2 == 0 && cols / 2 >= 32) ? 2 : nt);
        // 4096 columns over K >= 2048 (the backbone's qkv): 128 groups x 2 m-splits beat 64 x 4 (10.4 vs 10.8 us; the code
        // predictor's K = 1024 shape does not: same-box A/B of the step)
        // (33-48 rows -- three 16-row tiles -- take the 64-row choice too since round 5: 128 groups x 2 m-splits of 32 rows, the tile set of
        //  the persistent backbone launch, which now runs from 33 rows on: bb_chain.hip)
        if (!silu && nt == 4 && cols / 4 == 64 && a.K >= 2048 && mt_total >= 3) nt = 2;
        if (silu && nt < 2) nt = 2;
        if (gu8 && cols % 3 == 0 && cols / 3 >= 128) nt = 3;  // 256 workgroups (backbone), 128 x 2 m-splits (code predictor)
    } else if (silu && cols % 4 == 0 && cols / 4 >= 160 && mt_total == 4) {
        nt = 4;                                               // backbone gate_up at full batch: 32 + 32 columns, no M split
        if (gu8 && cols % 3 == 0 && cols / 3 >= 256) nt = 3;  // 24 columns x 256 workgroups: 14.8 vs 16.1 us
    }
    if (g_tile_nt && epi != OMNI_EPI_RESID && (!silu || g_tile_nt >= 2) && (g_tile_nt != 3 || gu8) && cols % g_tile_nt == 0) nt = g_tile_nt;
    const int groups = silu ? a.N / (8 * nt) : a.N / (16 * nt);
    int splits = groups >= 160 ? 1 : (g_gemm_wgs + groups - 1) / groups;
    // the 1024-wide gate_up (code predictor, 0.6B backbone: 128 groups of 24 columns) at 33-48 rows: three 16-row tiles, not a 32-row pair
    // with a half-empty partner (round 5) -- the tile set of the persistent chains there, whose fourth row group then has no work in ANY
    // stage and leaves at once instead of idling through four stages per layer to wake up for this one (B = 40: the predictor phase cost
    // 0.25 ms MORE than at 64 rows; cp_chain.hip k_cp_chain, bb_chain.hip)
    if (pro == 2 && gu8 && nt == 3 && groups == 128 && mt_total == 3) splits = 3;
    if (splits > mt_total) splits = mt_total;
    if (splits < 1) splits = 1;
    int mt = (mt_total + splits - 1) / splits;       // 1..4
    if (g_tile_mt) mt = g_tile_mt;
    if (mt == 3) mt = 4;
    if (mt > 4) mt = 4;
    *nt_out = nt; *mt_out = mt; *splits_out = (mt_total + mt - 1) / mt;
}

template <int PRO, int EPI>
static int dispatch_tile(const GemmArgs& a, hipStream_t st) {
    int nt, mt, splits;
    pick_tile(PRO == 3 ? 2 : PRO, EPI, a, &nt, &mt, &splits);      // the deferred-rstd form takes the norm-fused tiles
#define TILE(N_, M_) if (nt == N_ && mt == M_) return launch_gemm<M_, N_, PRO, EPI>(a, splits, st);
#define TILE_M(N_) TILE(N_, 1) TILE(N_, 2) TILE(N_, 4)
    if constexpr (EPI == OMNI_EPI_SILU_MUL) { TILE_M(2) TILE_M(4) }
    else if constexpr (EPI == OMNI_EPI_SILU_MUL_GU8) { TILE_M(2) TILE_M(3) TILE_M(4) }
    else if constexpr (EPI == OMNI_EPI_RESID || EPI == OMNI_EPI_F32) { TILE_M(1) }
    else if constexpr (EPI == OMNI_EPI_F32_BF16RND && PRO == 0) { TILE_M(1) }
    else { TILE_M(1) TILE_M(2) TILE_M(4) }
#undef TILE_M
#undef TILE
    omni_set_error("omni_gemm_bf16: no kernel for tile NT=%d MT=%d (epilogue %d)", nt, mt, EPI);
    return OMNI_EINVAL;
}

static int check_common(const GemmArgs& a) {
    OMNI_CHECK_ARG(a.W && a.out, "omni_gemm_bf16: null pointer");
    OMNI_CHECK_ARG(a.M >= 1 && a.M <= a.pstride, "omni_gemm_bf16: M=%d outside 1..%d", a.M, a.pstride);
    OMNI_CHECK_ARG(a.N > 0 && a.N % 16 == 0, "omni_gemm_bf16: N=%d not a multiple of 16", a.N);
    OMNI_CHECK_ARG(a.K > 0 && a.K % 32 == 0, "omni_gemm_bf16: K=%d not a multiple of 32", a.K);
    return OMNI_OK;
}

template <int PRO>
static int dispatch_epi(const GemmArgs& a, int epilogue, hipStream_t st) {
    switch (epilogue) {
        case OMNI_EPI_BF16:
            OMNI_CHECK_ARG(a.mask == nullptr, "omni_gemm_bf16: mask needs an fp32 epilogue");
            return dispatch_tile<PRO, OMNI_EPI_BF16>(a, st);
        case OMNI_EPI_SILU_MUL:
            OMNI_CHECK_ARG(a.bias == nullptr && a.mask == nullptr, "omni_gemm_bf16: silu_mul takes no bias/mask");
            return dispatch_tile<PRO, OMNI_EPI_SILU_MUL>(a, st);
        case OMNI_EPI_SILU_MUL_GU8:
            OMNI_CHECK_ARG(a.bias == nullptr && a.mask == nullptr && a.wshuf, "omni_gemm_bf16: silu_mul_gu8 takes no bias/mask and a fragment-major W");
            return dispatch_tile<PRO, OMNI_EPI_SILU_MUL_GU8>(a, st);
        case OMNI_EPI_F32:
            if constexpr (PRO == 0) return dispatch_tile<0, OMNI_EPI_F32>(a, st);
            break;
        case OMNI_EPI_F32_BF16RND:
            if constexpr (PRO != 3) return dispatch_tile<PRO, OMNI_EPI_F32_BF16RND>(a, st);
            break;
        case OMNI_EPI_RESID:
            if constexpr (PRO == 0) return dispatch_tile<0, OMNI_EPI_RESID>(a, st);
            break;
        default:
            break;
    }
    omni_set_error("omni_gemm_bf16: unsupported epilogue %d", epilogue);
    return OMNI_EINVAL;
}

int k_gemm_bf16_ex(const void* x, int ldx, const void* w, const void* bias, void* out, int M, int N, int K, int epilogue,
                   const uint8_t* mask, int layout, void* stream, float mask_fill) {
    GemmArgs a{};
    a.mask_fill = mask_fill;
    a.pstride = 64;
    a.counted = g_gemm_counted;
    a.wshuf = (layout & OMNI_LAYOUT_W_FRAG) != 0;
    a.xshuf = (layout & OMNI_LAYOUT_X_FRAG) != 0;
    a.oshuf = (layout & OMNI_LAYOUT_OUT_FRAG) != 0;
    OMNI_CHECK_ARG(epilogue != OMNI_EPI_RESID, "omni_gemm_bf16: the residual epilogue is omni_gemm_resid");
    OMNI_CHECK_ARG(!a.oshuf || epilogue == OMNI_EPI_BF16 || epilogue == OMNI_EPI_SILU_MUL || epilogue == OMNI_EPI_SILU_MUL_GU8,
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

extern "C" int omni_gemm_bf16_ex(const void* x, int ldx, const void* w, const void* bias, void* out, int M, int N,
                                 int K, int epilogue, const uint8_t* mask, int layout, void* stream) {
    return k_gemm_bf16_ex(x, ldx, w, bias, out, M, N, K, epilogue, mask, layout, stream, -INFINITY);
}

extern "C" int omni_gemm_bf16(const void* x, int ldx, const void* w, const void* bias, void* out, int M, int N,
                              int K, int epilogue, const uint8_t* mask, void* stream) {
    return omni_gemm_bf16_ex(x, ldx, w, bias, out, M, N, K, epilogue, mask, 0, stream);
}

// ---- the norm-free residual stream (fragment-major r + per-row sum-of-squares slabs) ---------------------------------
// Producer: r = bf16(r + bf16(x . W^T + bias)) in place (accumulate) or r = bf16(x . W^T + bias), and
// partials[n group][row] = that workgroup's share of sum(r^2); *nparts_out = N / 16.
int k_gemm_resid(const void* x, int ldx, const void* w, const void* bias, void* r_io, int accumulate, float* partials,
                 int* nparts_out, int M, int N, int K, int layout, int pstride, void* stream) {
    GemmArgs a{};
    a.mask_fill = -INFINITY;
    OMNI_CHECK_ARG(pstride == 64 || pstride == 128, "omni_gemm_resid: slab stride %d", pstride);
    a.pstride = pstride;
    a.counted = g_gemm_counted;
    a.wshuf = (layout & OMNI_LAYOUT_W_FRAG) != 0;
    a.xshuf = (layout & OMNI_LAYOUT_X_FRAG) != 0;
    a.oshuf = 1;
    a.x = (const uint16_t*)x; a.ldx = ldx; a.W = (const uint16_t*)w; a.bias = (const uint16_t*)bias; a.out = r_io;
    a.resid = accumulate ? (const uint16_t*)r_io : nullptr;
    a.part_out = partials;
    a.M = M; a.N = N; a.K = K;
    int rc = check_common(a);
    if (rc != OMNI_OK) return rc;
    OMNI_CHECK_ARG(x && partials, "omni_gemm_resid: null pointer");
    OMNI_CHECK_ARG(N % 32 == 0, "omni_gemm_resid: N=%d not a multiple of 32 (fragment-major residual stream)", N);
    OMNI_CHECK_ARG(a.xshuf || (ldx >= K && ldx % 8 == 0), "omni_gemm_resid: ldx=%d (need >= K, multiple of 8)", ldx);
    if (nparts_out) *nparts_out = N / 16;
    return dispatch_epi<0>(a, OMNI_EPI_RESID, (hipStream_t)stream);
}
extern "C" int omni_gemm_resid(const void* x, int ldx, const void* w, const void* bias, void* r_io, int accumulate,
                               float* partials, int* nparts_out, int M, int N, int K, int layout, void* stream) {
    return k_gemm_resid(x, ldx, w, bias, r_io, accumulate, partials, nparts_out, M, N, K, layout, 64, stream);
}

// Consumer: out = epilogue( (norm_w * bf16(r * rstd)) . W^T ), rstd = rsqrt(sum_p partials[p][row] / K + eps);
// normed_out (row-major, optional) receives the normalised rows.  r and W fragment-major.
int k_gemm_xnorm(const void* r, const float* partials, int nparts, const void* norm_w, float eps, void* normed_out,
                 const void* w, void* out, int M, int N, int K, int epilogue, const uint8_t* mask, int out_frag, int pstride,
                 void* stream, float mask_fill, const int32_t* num_live) {
    GemmArgs a{};
    a.mask_fill = mask_fill;
    a.num_live = num_live;
    OMNI_CHECK_ARG(pstride == 64 || pstride == 128, "omni_gemm_xnorm: slab stride %d", pstride);
    a.pstride = pstride;
    a.counted = g_gemm_counted;
    a.wshuf = 1; a.xshuf = 1; a.oshuf = out_frag != 0;
    a.x = (const uint16_t*)r; a.ldx = K; a.W = (const uint16_t*)w; a.out = out;
    a.M = M; a.N = N; a.K = K; a.mask = mask;
    a.norm_w = (const uint16_t*)norm_w; a.eps = eps; a.normed_out = (uint16_t*)normed_out;
    a.part_in = partials; a.np_in = nparts;
    int rc = check_common(a);
    if (rc != OMNI_OK) return rc;
    OMNI_CHECK_ARG(r && partials && norm_w && nparts >= 1, "omni_gemm_xnorm: null pointer / nparts=%d", nparts);
    OMNI_CHECK_ARG(epilogue == OMNI_EPI_BF16 || epilogue == OMNI_EPI_SILU_MUL || epilogue == OMNI_EPI_SILU_MUL_GU8 ||
                   epilogue == OMNI_EPI_F32_BF16RND, "omni_gemm_xnorm: epilogue %d unsupported", epilogue);
    OMNI_CHECK_ARG(!a.oshuf || (epilogue != OMNI_EPI_F32_BF16RND && N % 32 == 0),
                   "omni_gemm_xnorm: fragment-major output needs a bf16 epilogue and N %% 32 == 0");
    // deferred rstd (PRO 3) where the normalised rows are nobody's output and the epilogue is a bf16 one: qkv, gate_up -- the rule of the
    // persistent chains (cp_chain.hip, bb_chain.hip, moe_chain.hip), so that both schedules produce the same bits
    if (g_gemm_defer && normed_out == nullptr && epilogue != OMNI_EPI_F32_BF16RND) return dispatch_epi<3>(a, epilogue, (hipStream_t)stream);
    return dispatch_epi<2>(a, epilogue, (hipStream_t)stream);
}
extern "C" int omni_gemm_xnorm(const void* r, const float* partials, int nparts, const void* norm_w, float eps,
                               void* normed_out, const void* w, void* out, int M, int N, int K, int epilogue,
                               const uint8_t* mask, int out_frag, void* stream) {
    return k_gemm_xnorm(r, partials, nparts, norm_w, eps, normed_out, w, out, M, N, K, epilogue, mask, out_frag, 64, stream);
}
