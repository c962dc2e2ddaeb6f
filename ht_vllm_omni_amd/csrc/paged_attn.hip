// Paged-attention decode (query_len = 1) over bf16 / fp8-e4m3fn / int8 KV blocks, head_dim 128.
//
// HBM-bound: every K and V byte of the context is read exactly once, 16 B per lane, each
// 128-B (fp8/int8) or 256-B (bf16) token row fully consumed.  Workgroup = 4 waves = one
// (row, kv-head[, KV split]); a wave walks 8-token groups (8 lanes per token, 16 elements per
// lane); the G = Hq/Hkv query heads of the kv head share every K/V load.  Dequant in registers,
// fp32 scores, online softmax per 8-lane group (running max/sum, no cross-wave traffic in the
// loop), one LDS combine at the end; optional KV splits merged by a second tiny kernel.
// Output = oracle attention_rows (fp32 softmax and PV, one rounding to bf16).
#include "common.cuh"

#define PA_THREADS 256
#define PA_WAVES 4
#define PA_U 4                 // 8-token groups in flight per wave
#define LOG2E 1.4426950408889634f

template <int KV>
struct KVLoad {  // raw bytes of one lane's 16 elements of one token row
    uint4 a;
    uint4 b;     // bf16 only (second 8 elements)
};

// element index (0..127) of the e-th value (0..15) a lane with sub = lane&7 holds
template <int KV>
__device__ __forceinline__ int elem_of(int sub, int e) {
    if (KV == OMNI_KV_BF16) return (e < 8) ? (sub * 8 + e) : (64 + sub * 8 + (e - 8));
    return sub * 16 + e;
}

template <int KV>
__device__ __forceinline__ KVLoad<KV> load_row(const void* base, size_t row, int sub) {
    KVLoad<KV> r;
    if (KV == OMNI_KV_BF16) {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(base) + row * 128;
        r.a = *reinterpret_cast<const uint4*>(p + sub * 8);
        r.b = *reinterpret_cast<const uint4*>(p + 64 + sub * 8);
    } else {
        const uint8_t* p = reinterpret_cast<const uint8_t*>(base) + row * 128;
        r.a = *reinterpret_cast<const uint4*>(p + sub * 16);
        r.b = make_uint4(0, 0, 0, 0);
    }
    return r;
}

template <int KV>
__device__ __forceinline__ void to_f32(const KVLoad<KV>& r, float* f) {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(&r.a);
    if (KV == OMNI_KV_BF16) {
        const uint32_t* b = reinterpret_cast<const uint32_t*>(&r.b);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j] = bf_lo(a[j]);
            f[2 * j + 1] = bf_hi(a[j]);
            f[8 + 2 * j] = bf_lo(b[j]);
            f[8 + 2 * j + 1] = bf_hi(b[j]);
        }
    } else if (KV == OMNI_KV_FP8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) unpack_fp8x4(a[j], f + 4 * j);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[4 * j + 0] = (float)(int8_t)(a[j] & 0xFF);
            f[4 * j + 1] = (float)(int8_t)((a[j] >> 8) & 0xFF);
            f[4 * j + 2] = (float)(int8_t)((a[j] >> 16) & 0xFF);
            f[4 * j + 3] = (float)(int8_t)(a[j] >> 24);
        }
    }
}

// partial record per (row, q-head, split): [0]=m (log2 domain) [1]=l [2..129]=acc (unnormalised)
#define PA_REC 130

template <int KV, int G>
__global__ __launch_bounds__(PA_THREADS) void paged_attn_decode_kernel(
    const uint16_t* __restrict__ q, const void* __restrict__ k_cache, const void* __restrict__ v_cache,
    const float* __restrict__ k_scales, const float* __restrict__ v_scales, const int32_t* __restrict__ block_table,
    int bt_stride, const int32_t* __restrict__ seq_lens, const int32_t* __restrict__ req_of_row, int seq_from_pos,
    uint16_t* __restrict__ out, float* __restrict__ partial, int q_heads, int kv_heads, int bs, float k_scale,
    float v_scale, float sm_scale, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [PA_WAVES*8][G][PA_REC]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 7, tg = lane >> 3;
    const int kvh = blockIdx.x, row = blockIdx.y, sp = blockIdx.z;
    const int req = req_of_row ? req_of_row[row] : row;
    const int seq_len = seq_lens[row] + (seq_from_pos ? 1 : 0);
    // token range of this split (multiple of 32 so 8-token groups never straddle splits)
    int per = (seq_len + nsplit - 1) / nsplit;
    per = (per + 31) & ~31;
    const int t_begin = sp * per;
    const int t_end = min(seq_len, t_begin + per);

    // q for the G heads of this kv head, pre-scaled into the log2 domain
    float qf[G][16];
    const float qs = sm_scale * LOG2E * (KV == OMNI_KV_FP8 ? k_scale : 1.0f);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const uint16_t* qp = q + ((size_t)row * q_heads + kvh * G + g) * 128;
#pragma unroll
        for (int e = 0; e < 16; ++e) qf[g][e] = bf2f(qp[elem_of<KV>(sub, e)]) * qs;
    }
    float m[G], l[G], acc[G][16];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
    }
    const int32_t* bt = block_table + (size_t)req * bt_stride;

    for (int t0 = t_begin + wave * 8; t0 < t_end; t0 += PA_WAVES * 8 * PA_U) {
        KVLoad<KV> kr[PA_U], vr[PA_U];
        float ksc[PA_U], vsc[PA_U];
        bool valid[PA_U];
#pragma unroll
        for (int u = 0; u < PA_U; ++u) {
            const int t = t0 + u * PA_WAVES * 8 + tg;
            valid[u] = t < t_end;
            const int tc = valid[u] ? t : (t_end - 1);        // clamp: valid address, masked below
            const size_t r = ((size_t)bt[tc / bs] * bs + tc % bs) * kv_heads + kvh;
            kr[u] = load_row<KV>(k_cache, r, sub);
            vr[u] = load_row<KV>(v_cache, r, sub);
            if (KV == OMNI_KV_INT8) {
                ksc[u] = k_scales[r];
                vsc[u] = v_scales[r];
            }
        }
        float s[PA_U][G];
        float mx[G];
#pragma unroll
        for (int g = 0; g < G; ++g) mx[g] = m[g];
#pragma unroll
        for (int u = 0; u < PA_U; ++u) {
            float kf[16];
            to_f32<KV>(kr[u], kf);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) d = fmaf(qf[g][e], kf[e], d);
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                if (KV == OMNI_KV_INT8) d *= ksc[u];
                d = valid[u] ? d : -INFINITY;
                s[u][g] = d;
                mx[g] = fmaxf(mx[g], d);
            }
        }
        // one rescale per batch of PA_U tokens (mx stays -inf only if nothing valid yet)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float corr = (mx[g] == -INFINITY) ? 1.0f : exp2f(m[g] - mx[g]);
            m[g] = mx[g];
            l[g] *= corr;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[g][e] *= corr;
        }
#pragma unroll
        for (int u = 0; u < PA_U; ++u) {
            float vf[16];
            to_f32<KV>(vr[u], vf);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float p = valid[u] ? exp2f(s[u][g] - m[g]) : 0.f;
                l[g] += p;
                if (KV == OMNI_KV_INT8) p *= vsc[u];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[g][e] = fmaf(p, vf[e], acc[g][e]);
            }
        }
    }

    // ---- combine: every (wave, token-group) publishes (m, l, acc) per head through LDS
    const int part = wave * 8 + tg;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float* rec = lds + ((size_t)part * G + g) * PA_REC;
        if (sub == 0) {
            rec[0] = m[g];
            rec[1] = l[g];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) rec[2 + elem_of<KV>(sub, e)] = acc[g][e];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < G * 128; it += PA_THREADS) {
        const int g = it >> 7, d = it & 127;
        float M = -INFINITY;
#pragma unroll 8
        for (int p = 0; p < PA_WAVES * 8; ++p) M = fmaxf(M, lds[((size_t)p * G + g) * PA_REC]);
        float L = 0.f, A = 0.f;
#pragma unroll 8
        for (int p = 0; p < PA_WAVES * 8; ++p) {
            const float* rec = lds + ((size_t)p * G + g) * PA_REC;
            const float w = (rec[0] == -INFINITY) ? 0.f : exp2f(rec[0] - M);
            L = fmaf(rec[1], w, L);
            A = fmaf(rec[2 + d], w, A);
        }
        const int qh = kvh * G + g;
        if (nsplit == 1) {
            const float vs = (KV == OMNI_KV_FP8) ? v_scale : 1.0f;
            out[((size_t)row * q_heads + qh) * 128 + d] = f2bf(L > 0.f ? (A / L) * vs : 0.f);
        } else {
            float* rec = partial + (((size_t)row * q_heads + qh) * nsplit + sp) * PA_REC;
            if (d == 0) {
                rec[0] = M;
                rec[1] = L;
            }
            rec[2 + d] = A;
        }
    }
}

// merge KV splits: one 128-thread block per (row, q-head)
__global__ __launch_bounds__(128) void paged_attn_merge_kernel(const float* __restrict__ partial,
                                                               uint16_t* __restrict__ out, int nsplit, float v_mul) {
    const size_t rh = blockIdx.x;
    const int d = threadIdx.x;
    const float* base = partial + rh * nsplit * PA_REC;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, base[s * PA_REC]);
    float L = 0.f, A = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* rec = base + s * PA_REC;
        const float w = (rec[0] == -INFINITY) ? 0.f : exp2f(rec[0] - M);
        L = fmaf(rec[1], w, L);
        A = fmaf(rec[2 + d], w, A);
    }
    out[rh * 128 + d] = f2bf(L > 0.f ? (A / L) * v_mul : 0.f);
}

static int pick_nsplit(int rows, int kv_heads, int max_seq_len) {
    int wgs = rows * kv_heads;
    int ns = (1024 + wgs - 1) / wgs;                  // aim for >= ~1024 workgroups (4/CU)
    int cap = (max_seq_len + 255) / 256;              // >= 256 tokens per split
    if (ns > cap) ns = cap;
    if (ns > 16) ns = 16;
    return ns < 1 ? 1 : ns;
}

extern "C" int64_t omni_paged_attn_workspace_bytes(int B, int q_heads, int head_dim, int max_seq_len) {
    (void)head_dim;
    (void)max_seq_len;
    return (int64_t)B * q_heads * 16 * PA_REC * sizeof(float);
}

template <int KV>
static int launch_pa(const void* q, const void* kc, const void* vc, const float* ks, const float* vs,
                     const int32_t* bt, int bt_stride, const int32_t* seq_lens, const int32_t* req_of_row,
                     int seq_from_pos, void* out, void* ws, int rows, int q_heads, int kv_heads, int bs, float k_scale,
                     float v_scale, float sm_scale, int nsplit, hipStream_t st) {
    const int G = q_heads / kv_heads;
    dim3 grid(kv_heads, rows, nsplit), block(PA_THREADS);
    const size_t lds = (size_t)PA_WAVES * 8 * G * PA_REC * sizeof(float);
#define LAUNCH(GG)                                                                                                 \
    hipLaunchKernelGGL((paged_attn_decode_kernel<KV, GG>), grid, block, lds, st, (const uint16_t*)q, kc, vc, ks, vs, \
                       bt, bt_stride, seq_lens, req_of_row, seq_from_pos, (uint16_t*)out, (float*)ws, q_heads,      \
                       kv_heads, bs, k_scale, v_scale, sm_scale, nsplit)
    switch (G) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 4: LAUNCH(4); break;
        default: omni_set_error("omni_paged_attn: Hq/Hkv=%d unsupported (1,2,4)", G); return OMNI_EINVAL;
    }
#undef LAUNCH
    OMNI_CHECK_LAUNCH("omni_paged_attn_decode");
    if (nsplit > 1) {
        hipLaunchKernelGGL(paged_attn_merge_kernel, dim3(rows * q_heads), dim3(128), 0, st, (const float*)ws,
                           (uint16_t*)out, nsplit, KV == OMNI_KV_FP8 ? v_scale : 1.0f);
        OMNI_CHECK_LAUNCH("omni_paged_attn_merge");
    }
    return OMNI_OK;
}

static int pa_dispatch(const void* q, const void* kc, const void* vc, const float* ks, const float* vs,
                       const int32_t* bt, int bt_stride, const int32_t* seq_lens, const int32_t* req_of_row,
                       int seq_from_pos, void* out, void* ws, int rows, int q_heads, int kv_heads, int head_dim,
                       int bs, int kv_dtype, float k_scale, float v_scale, float sm_scale, int nsplit, void* stream) {
    OMNI_CHECK_ARG(q && kc && vc && bt && seq_lens && out, "omni_paged_attn: null pointer");
    OMNI_CHECK_ARG(head_dim == 128, "omni_paged_attn: head_dim=%d (only 128)", head_dim);
    OMNI_CHECK_ARG(kv_heads > 0 && q_heads % kv_heads == 0, "omni_paged_attn: q_heads=%d kv_heads=%d", q_heads, kv_heads);
    OMNI_CHECK_ARG(bs > 0, "omni_paged_attn: block_size=%d", bs);
    OMNI_CHECK_ARG(kv_dtype != OMNI_KV_INT8 || (ks && vs), "omni_paged_attn: int8 needs scale arrays");
    OMNI_CHECK_ARG(nsplit == 1 || ws, "omni_paged_attn: workspace required for KV splits");
    if (rows <= 0) return OMNI_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (kv_dtype) {
        case OMNI_KV_BF16:
            return launch_pa<OMNI_KV_BF16>(q, kc, vc, ks, vs, bt, bt_stride, seq_lens, req_of_row, seq_from_pos, out, ws,
                                           rows, q_heads, kv_heads, bs, k_scale, v_scale, sm_scale, nsplit, st);
        case OMNI_KV_FP8:
            return launch_pa<OMNI_KV_FP8>(q, kc, vc, ks, vs, bt, bt_stride, seq_lens, req_of_row, seq_from_pos, out, ws,
                                          rows, q_heads, kv_heads, bs, k_scale, v_scale, sm_scale, nsplit, st);
        case OMNI_KV_INT8:
            return launch_pa<OMNI_KV_INT8>(q, kc, vc, ks, vs, bt, bt_stride, seq_lens, req_of_row, seq_from_pos, out, ws,
                                           rows, q_heads, kv_heads, bs, k_scale, v_scale, sm_scale, nsplit, st);
        default:
            omni_set_error("omni_paged_attn: kv_dtype=%d", kv_dtype);
            return OMNI_EINVAL;
    }
}

extern "C" int omni_paged_attn_decode(const void* q, const void* k_cache, const void* v_cache, const float* k_scales,
                                      const float* v_scales, const int32_t* block_table, int bt_stride,
                                      const int32_t* seq_lens, void* out, void* workspace, int B, int q_heads,
                                      int kv_heads, int head_dim, int block_size, int kv_dtype, float k_scale,
                                      float v_scale, float sm_scale, int max_seq_len, void* stream) {
    const int nsplit = (workspace && kv_heads > 0 && B > 0) ? pick_nsplit(B, kv_heads, max_seq_len) : 1;
    return pa_dispatch(q, k_cache, v_cache, k_scales, v_scales, block_table, bt_stride, seq_lens, nullptr, 0, out,
                       workspace, B, q_heads, kv_heads, head_dim, block_size, kv_dtype, k_scale, v_scale, sm_scale,
                       nsplit, stream);
}

extern "C" int omni_paged_attn_prefill(const void* q, const void* k_cache, const void* v_cache, const float* k_scales,
                                       const float* v_scales, const int32_t* block_table, int bt_stride,
                                       const int32_t* req_of_tok, const int32_t* positions, void* out, int T,
                                       int q_heads, int kv_heads, int head_dim, int block_size, int kv_dtype,
                                       float k_scale, float v_scale, float sm_scale, void* stream) {
    OMNI_CHECK_ARG(req_of_tok && positions, "omni_paged_attn_prefill: null pointer");
    // every token is a decode row whose context is positions[t] + 1 keys of request req_of_tok[t]
    return pa_dispatch(q, k_cache, v_cache, k_scales, v_scales, block_table, bt_stride, positions, req_of_tok, 1, out,
                       nullptr, T, q_heads, kv_heads, head_dim, block_size, kv_dtype, k_scale, v_scale, sm_scale, 1,
                       stream);
}
