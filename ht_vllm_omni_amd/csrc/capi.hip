// Talker engine: the decode step of SURVEY 3.3 as a fixed sequence of kernel launches on one
// HIP stream (no allocation, no host sync -> capturable into a hipGraph by the caller).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "coherent.cuh"
#include "common.cuh"
#include "kernels.h"

// ------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";
void omni_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* omni_last_error(void) { return g_err; }
extern "C" int omni_abi_version(void) { return 5; }

#define TRY(expr)                    \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != OMNI_OK) return rc__; \
    } while (0)

// ------------------------------------------------------------------ small step kernels
// x[t+1] = bf16( bf16( sum_fp32(e0, emb_1..emb_{Q-1}) ) + text_step ); invalid layer-0 id -> frame of zeros
// (qwen3_tts_talker.py:1630-1641).  One block per row, 8 elements (16 B) per thread, all gathers independent.
__global__ __launch_bounds__(256) void mtp_finalize_kernel(
    const int32_t* __restrict__ input_ids, const int32_t* __restrict__ codes /*[B,Q]*/, const uint16_t* __restrict__ embed,
    int vocab, const uint16_t* __restrict__ cp_embed, const uint16_t* __restrict__ text_step, uint16_t* __restrict__ x_out,
    uint16_t* __restrict__ resid_out, float* __restrict__ part_out /* != NULL: resid_out fragment-major + sum(x^2) slab 0 */,
    int64_t* __restrict__ audio_codes, int H, int Q, int codebook) {
    const int b = blockIdx.x;
    const int c0 = input_ids[b];
    const bool invalid0 = c0 < 0 || c0 >= codebook;
    __shared__ int cg[64];
    if (threadIdx.x < Q) {
        int c = threadIdx.x == 0 ? c0 : codes[(size_t)b * Q + threadIdx.x];
        if (invalid0) c = 0;
        cg[threadIdx.x] = c;
        audio_codes[(size_t)b * Q + threadIdx.x] = (int64_t)c;
    }
    __syncthreads();
    float ss = 0.f;
    for (int v = threadIdx.x; v < H / 8; v += blockDim.x) {
        float s[8];
        {
            // e0 = embed_input_ids(last sampled id): the id itself, not the zeroed code (talker.py:1636)
            const bool ok = c0 >= 0 && c0 < vocab;
            const uint4 e = ok ? *reinterpret_cast<const uint4*>(embed + (size_t)c0 * H + v * 8) : make_uint4(0, 0, 0, 0);
            const uint32_t* w = reinterpret_cast<const uint32_t*>(&e);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[2 * j] = bf_lo(w[j]); s[2 * j + 1] = bf_hi(w[j]); }
        }
#pragma unroll 4
        for (int g = 1; g < Q; ++g) {
            const uint4 e = *reinterpret_cast<const uint4*>(cp_embed + ((size_t)(g - 1) * codebook + cg[g]) * H + v * 8);
            const uint32_t* w = reinterpret_cast<const uint32_t*>(&e);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[2 * j] += bf_lo(w[j]); s[2 * j + 1] += bf_hi(w[j]); }
        }
        const uint4 tx = *reinterpret_cast<const uint4*>(text_step + (size_t)b * H + v * 8);
        const uint32_t* tw = reinterpret_cast<const uint32_t*>(&tx);
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack_bf2(bfround(s[2 * j]) + bf_lo(tw[j]), bfround(s[2 * j + 1]) + bf_hi(tw[j]));
        const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4*>(x_out + (size_t)b * H + v * 8) = ov;
        if (part_out) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ss += bf_lo(o[j]) * bf_lo(o[j]) + bf_hi(o[j]) * bf_hi(o[j]);
            *reinterpret_cast<uint4*>(resid_out + frag_off(b, v * 8, H)) = ov;
        } else {
            *reinterpret_cast<uint4*>(resid_out + (size_t)b * H + v * 8) = ov;
        }
    }
    if (part_out) {
        __shared__ float red[4];
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        if (threadIdx.x == 0) part_out[b] = red[0] + red[1] + red[2] + red[3];
    }
}

__global__ void copy_i32_to_i64_kernel(const int32_t* src, int64_t* dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ------------------------------------------------------------------ engine
struct omni_talker {
    omni_talker_desc d;
    std::vector<omni_layer_weights> layer, cp_layer;
    std::vector<void*> k_cache, v_cache;
    std::vector<float*> k_scales, v_scales;
    int Bm, cp_bs;
    // scratch carve
    uint16_t *resid, *resid_b, *normed, *qkv, *q, *attn, *attn_out, *act, *mlp_out, *hidden, *e0;
    uint16_t *normed_rm, *moe_logits, *moe_w, *moe_act, *moe_y, *moe_shared;   // sparse-MoE MLP scratch
    int32_t* moe_idx;
    float *attn_ws, *part, *cp_part;   // sum-of-squares slabs of the fused-norm residual streams
    uint16_t *cp_resid, *cp_resid_b, *cp_normed, *cp_qkv, *cp_q, *cp_attn, *cp_o, *cp_act, *cp_mlp, *cp_hidden, *cp_in, *cp_row;
    float* cp_logits;
    int32_t *codes, *cp_bt, *cp_pos, *cp_seq;
    int64_t* cp_slots;
    std::vector<uint16_t*> cp_k, cp_v;
    int32_t* pf_seq;   // prefill scratch
    float* kv_scale_dev;                // fp8 KV: device [layers][2] = {k_scale, v_scale} per layer, read by the decode step's attention launches
    std::vector<float> ksc_h, vsc_h;    // the same on the host (arguments of the eager prefill launches)
    void* bb_table;                     // device table of the backbone layers' pointers (bb_all.hip)
    uint32_t* chain_flags;              // stage flags of the persistent chains: OMNI_FLAG_REPLICAS copies of [256] (coherent.cuh) + the error word at [320]
    omni_ar_peers ar_attn, ar_mlp;      // copies of desc.ar_* (has_ar)
    bool has_ar;
    bool chain_half;                    // omni_talker_set_chains(t, 2): the backbone chain on 128 workgroups, the predictor on the launch path
    bool bb_ar_off;                     // omni_talker_set_chains(t, 3): a tensor-parallel rank's backbone stays launch per op (all-reduce launches)
    int ran;                            // persistent chains launched by the decode-step call in progress / last made (bit 0 cp, bit 1 bb)
    bool head_fused = false;            // the last backbone launch of the step in progress computed logits + h[t + 1] (run_backbone fuse_head)
    bool tail_fused = false;            // the predictor's all-pass launch of the step in progress assembled the backbone's input and computed layer 0's qkv
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct Carver {
    char* base;
    size_t off;
    template <typename T>
    T* take(size_t n) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += align_up(n * sizeof(T));
        return p;
    }
};

static int carve(omni_talker* t, char* base, size_t* total) {
    const omni_talker_desc& d = t->d;
    const size_t B = d.max_batch, H = d.hidden, Hc = d.cp_hidden, Q = d.num_code_groups;
    const size_t qkv_out = (size_t)(d.q_heads + 2 * d.kv_heads) * d.head_dim;
    const size_t cp_qkv_out = (size_t)(d.cp_q_heads + 2 * d.cp_kv_heads) * d.cp_head_dim;
    Carver c{base, 0};
    const size_t B16 = (B + 15) & ~(size_t)15;      // fragment-major activation buffers hold whole 16-row tiles
    t->resid = c.take<uint16_t>(B16 * H);          // fragment-major in the fused-norm step
    t->resid_b = c.take<uint16_t>(B * H);
    t->part = c.take<float>(H / 16 * 64);
    t->cp_part = c.take<float>(Hc / 16 * 128);       // two-position pass: 128 rows per slab
    t->normed = c.take<uint16_t>(B16 * H);
    t->qkv = c.take<uint16_t>(B * qkv_out);
    t->q = c.take<uint16_t>(B * d.q_heads * d.head_dim);
    t->attn = c.take<uint16_t>(B16 * d.q_heads * d.head_dim);
    t->attn_out = c.take<uint16_t>(B * H);
    t->act = c.take<uint16_t>(B16 * d.inter);
    t->mlp_out = c.take<uint16_t>(B * H);
    t->hidden = c.take<uint16_t>(B * H);
    t->e0 = c.take<uint16_t>(B * H);
    t->attn_ws = c.take<float>((size_t)omni_paged_attn_workspace_bytes(d.max_batch, d.q_heads, d.head_dim, d.max_model_len) / 4);
    t->cp_resid = c.take<uint16_t>(2 * B16 * Hc);     // rows [0, B) position 0 and [B16, B16 + B) position 1 in the pair pass
    t->cp_resid_b = c.take<uint16_t>(B * Hc);
    t->cp_normed = c.take<uint16_t>(B16 * Hc);
    t->cp_qkv = c.take<uint16_t>(2 * B16 * cp_qkv_out);
    t->cp_q = c.take<uint16_t>(B * d.cp_q_heads * d.cp_head_dim);
    t->cp_attn = c.take<uint16_t>(2 * B16 * d.cp_q_heads * d.cp_head_dim);
    t->cp_o = c.take<uint16_t>(B * Hc);
    t->cp_act = c.take<uint16_t>(2 * B16 * d.cp_inter);
    t->cp_mlp = c.take<uint16_t>(B * Hc);
    t->cp_hidden = c.take<uint16_t>(B * Hc);
    t->cp_in = c.take<uint16_t>(B * Hc);
    t->cp_row = c.take<uint16_t>(B * H);
    t->cp_logits = c.take<float>(B * d.codebook);
    t->codes = c.take<int32_t>(B * Q);
    t->cp_bt = c.take<int32_t>(B);
    t->cp_pos = c.take<int32_t>((Q + 1) * B);
    t->cp_seq = c.take<int32_t>((Q + 1) * B);
    t->cp_slots = c.take<int64_t>((Q + 1) * B);
    t->pf_seq = c.take<int32_t>(8);
    t->chain_flags = c.take<uint32_t>(OMNI_FLAG_WORDS);
    t->kv_scale_dev = c.take<float>(2 * (size_t)(d.layers > 0 ? d.layers : 1));
#ifdef OMNI_DEBUG_HOOKS
    t->bb_table = c.take<char>(k_bb_all_table_bytes(d.layers > 0 ? d.layers : 1));
#endif
    if (d.moe_experts > 0) {
        const size_t k = d.moe_top_k, Im = d.moe_inter, Is = d.moe_shared_inter;
        t->normed_rm = c.take<uint16_t>(B * H);
        t->moe_logits = c.take<uint16_t>(B * d.moe_experts);
        t->moe_idx = c.take<int32_t>(B * k);
        t->moe_w = c.take<uint16_t>(B * k);
        t->moe_act = c.take<uint16_t>(B * k * Im);
        t->moe_y = c.take<uint16_t>(B * k * H);
        t->moe_shared = c.take<uint16_t>(B * H);
        if (Is > d.inter) c.take<uint16_t>(B16 * (Is - d.inter));      // t->act doubles as the shared expert's activation
    }
    t->cp_k.resize(d.cp_layers);
    t->cp_v.resize(d.cp_layers);
    const size_t cpkv = B * t->cp_bs * d.cp_kv_heads * d.cp_head_dim;
    for (int l = 0; l < d.cp_layers; ++l) {
        t->cp_k[l] = c.take<uint16_t>(cpkv);
        t->cp_v[l] = c.take<uint16_t>(cpkv);
    }
    *total = c.off;
    return OMNI_OK;
}

static int check_desc(const omni_talker_desc* d) {
    OMNI_CHECK_ARG(d, "omni_talker: null descriptor");
    OMNI_CHECK_ARG(d->head_dim == 128 && d->cp_head_dim == 128, "omni_talker: head_dim must be 128");
    OMNI_CHECK_ARG(d->max_batch >= 1 && d->max_batch <= 64, "omni_talker: max_batch=%d outside 1..64", d->max_batch);
    OMNI_CHECK_ARG(d->block_size > 0 && (d->block_size & (d->block_size - 1)) == 0, "omni_talker: block_size=%d must be a power of two", d->block_size);
    OMNI_CHECK_ARG(d->num_code_groups >= 1 && d->num_code_groups <= 63, "omni_talker: num_code_groups=%d", d->num_code_groups);
    OMNI_CHECK_ARG(d->hidden % 32 == 0 && d->inter % 32 == 0 && d->cp_hidden % 32 == 0 && d->cp_inter % 32 == 0,
                   "omni_talker: hidden/intermediate sizes must be multiples of 32");
    OMNI_CHECK_ARG(d->vocab % 16 == 0 && d->codebook % 16 == 0, "omni_talker: vocab/codebook must be multiples of 16");
    OMNI_CHECK_ARG(d->kv_heads > 0 && d->q_heads % d->kv_heads == 0 && d->cp_kv_heads > 0 && d->cp_q_heads % d->cp_kv_heads == 0,
                   "omni_talker: bad head counts");
    return OMNI_OK;
}

extern "C" int64_t omni_talker_scratch_bytes(const omni_talker_desc* desc) {
    if (check_desc(desc) != OMNI_OK) return -1;
    omni_talker tmp;
    tmp.d = *desc;
    tmp.cp_bs = desc->num_code_groups + 1;
    size_t total = 0;
    carve(&tmp, nullptr, &total);
    return (int64_t)total;
}

extern "C" omni_talker* omni_talker_create(const omni_talker_desc* desc) {
    if (check_desc(desc) != OMNI_OK) return nullptr;
    if (!desc->layer || !desc->cp_layer || !desc->k_cache || !desc->v_cache || !desc->scratch) {
        omni_set_error("omni_talker_create: null weight/cache/scratch pointer");
        return nullptr;
    }
    if (desc->has_cp_projection && (!desc->cp_proj_w || !desc->cp_proj_b)) {
        omni_set_error("omni_talker_create: projection weights missing");
        return nullptr;
    }
    if (!desc->has_cp_projection && desc->cp_hidden != desc->hidden) {
        omni_set_error("omni_talker_create: cp_hidden != hidden needs the projection");
        return nullptr;
    }
    if ((desc->fused_norm || desc->cp_fused_norm) && !desc->frag_layout) {
        omni_set_error("omni_talker_create: fused_norm / cp_fused_norm need frag_layout");
        return nullptr;
    }
    if (desc->moe_experts > 0 && (!desc->frag_layout || desc->moe_top_k < 1 || desc->moe_top_k > 8 ||
                                  desc->moe_experts % 16 || desc->moe_experts > 256 || desc->moe_inter % 32 ||
                                  desc->moe_shared_inter % 32 || desc->hidden % 64 || desc->moe_shared_inter > desc->inter)) {
        omni_set_error("omni_talker_create: MoE backbone needs frag_layout, top_k <= 8, experts %% 16 == 0 (<= 256), "
                       "moe / shared intermediate %% 32 == 0, hidden %% 64 == 0, shared intermediate <= inter (scratch)");
        return nullptr;
    }
    omni_talker* t = new omni_talker();
    t->d = *desc;
    t->Bm = desc->max_batch;
    t->cp_bs = desc->num_code_groups + 1;
    t->layer.assign(desc->layer, desc->layer + desc->layers);
    t->cp_layer.assign(desc->cp_layer, desc->cp_layer + desc->cp_layers);
    t->k_cache.assign(desc->k_cache, desc->k_cache + desc->layers);
    t->v_cache.assign(desc->v_cache, desc->v_cache + desc->layers);
    if (desc->kv_dtype == OMNI_KV_INT8) {
        if (!desc->k_scales || !desc->v_scales) {
            omni_set_error("omni_talker_create: int8 KV needs scale arrays");
            delete t;
            return nullptr;
        }
        t->k_scales.assign(desc->k_scales, desc->k_scales + desc->layers);
        t->v_scales.assign(desc->v_scales, desc->v_scales + desc->layers);
    } else {
        t->k_scales.assign(desc->layers, nullptr);
        t->v_scales.assign(desc->layers, nullptr);
    }
    t->has_ar = desc->ar_attn != nullptr && desc->ar_mlp != nullptr;
    if (t->has_ar) {
        t->ar_attn = *desc->ar_attn;
        t->ar_mlp = *desc->ar_mlp;
        if (!desc->fused_norm || t->ar_attn.world != t->ar_mlp.world || t->ar_attn.rank != t->ar_mlp.rank ||
            t->ar_attn.world < 1 || t->ar_attn.world > 8 || !t->ar_attn.data[t->ar_attn.rank] || !t->ar_mlp.data[t->ar_mlp.rank]) {
            omni_set_error("omni_talker_create: peer-mapped all-reduce needs fused_norm and consistent peer tables");
            delete t;
            return nullptr;
        }
    }
    size_t total = 0;
    carve(t, reinterpret_cast<char*>(desc->scratch), &total);
    if ((int64_t)total > desc->scratch_bytes) {
        omni_set_error("omni_talker_create: scratch too small (%lld < %zu)", (long long)desc->scratch_bytes, total);
        delete t;
        return nullptr;
    }
    // constant code-predictor metadata: request b owns "block" b of cp_bs positions
    const int B = t->Bm, Q = desc->num_code_groups;
    std::vector<int32_t> bt(B), pos((Q + 1) * B), seq((Q + 1) * B);
    std::vector<int64_t> slots((Q + 1) * B);
    for (int b = 0; b < B; ++b) bt[b] = b;
    for (int p = 0; p <= Q; ++p)
        for (int b = 0; b < B; ++b) {
            pos[p * B + b] = p;
            seq[p * B + b] = p + 1;
            slots[p * B + b] = (int64_t)b * t->cp_bs + p;
        }
#ifdef OMNI_DEBUG_HOOKS      // round 4's one-launch backbone (bb_all.hip): an A/B arm of the debug library (it lost: DESIGN 6)
    if (desc->moe_experts == 0 &&
        k_bb_all_table(t->bb_table, t->d, t->layer.data(), t->k_cache.data(), t->v_cache.data(),
                       desc->kv_dtype == OMNI_KV_INT8 ? t->k_scales.data() : nullptr, desc->kv_dtype == OMNI_KV_INT8 ? t->v_scales.data() : nullptr) != OMNI_OK) {
        delete t;
        return nullptr;
    }
#endif
    hipError_t e = hipMemcpy(t->cp_bt, bt.data(), bt.size() * 4, hipMemcpyHostToDevice);
    t->ksc_h.assign(desc->layers, desc->k_scale);
    t->vsc_h.assign(desc->layers, desc->v_scale);
    {
        std::vector<float> sc(2 * (size_t)desc->layers);
        for (int l = 0; l < desc->layers; ++l) { sc[2 * l] = desc->k_scale; sc[2 * l + 1] = desc->v_scale; }
        if (e == hipSuccess && desc->layers > 0) e = hipMemcpy(t->kv_scale_dev, sc.data(), sc.size() * 4, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemset(t->chain_flags, 0, OMNI_FLAG_WORDS * 4);
    if (e == hipSuccess) e = hipMemcpy(t->cp_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->cp_seq, seq.data(), seq.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->cp_slots, slots.data(), slots.size() * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        omni_set_error("omni_talker_create: hipMemcpy failed: %s", hipGetErrorString(e));
        delete t;
        return nullptr;
    }
    return t;
}

extern "C" void omni_talker_destroy(omni_talker* t) { delete t; }

// the sticky error word of the persistent code-predictor chain: 0 = every flag wait so far was satisfied; otherwise the
// stage code (16 * layer + stage + 1) whose spin ran out first.  Synchronises the device; reset = 1 clears word and flags.
extern "C" int omni_talker_chain_error(omni_talker* t, int reset) {
    if (!t) return OMNI_EINVAL;
    int32_t v = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, t->chain_flags + 320, 4, hipMemcpyDeviceToHost) != hipSuccess) {
        omni_set_error("omni_talker_chain_error: device read failed");
        return OMNI_EHIP;
    }
    if (reset == 1 && hipMemset(t->chain_flags, 0, OMNI_FLAG_WORDS * 4) != hipSuccess) return OMNI_EHIP;
    if (reset == 2) {                   // fault injection (tests of the host's fall-back): as if a flag wait had just timed out
        const int32_t code = 0x7ffe;
        if (hipMemcpy(t->chain_flags + 320, &code, 4, hipMemcpyHostToDevice) != hipSuccess) return OMNI_EHIP;
    }
    return v;
}
extern "C" int omni_talker_set_chains(omni_talker* t, int on) {
    if (!t) return OMNI_EINVAL;
    t->d.cp_chain = on != 0;            // k_cp_chain_supported / k_bb_chain_supported read it per call
    t->chain_half = on == 2;
    t->bb_ar_off = on == 3;
    return OMNI_OK;
}
extern "C" int omni_talker_chains_ran(const omni_talker* t) { return t ? t->ran : 0; }
// per-layer fp8 KV scales (host arrays [layers]) after a calibration pass: the device copy the captured decode steps read, and the
// host copy the eager prefill launches pass as arguments
extern "C" int omni_talker_set_kv_scales(omni_talker* t, const float* k_scale, const float* v_scale, void* stream) {
    OMNI_CHECK_ARG(t && k_scale && v_scale, "omni_talker_set_kv_scales: null pointer");
    const int L = t->d.layers;
    std::vector<float> sc(2 * (size_t)L);
    for (int l = 0; l < L; ++l) {
        OMNI_CHECK_ARG(k_scale[l] > 0.f && v_scale[l] > 0.f, "omni_talker_set_kv_scales: layer %d: scales must be > 0", l);
        t->ksc_h[l] = k_scale[l]; t->vsc_h[l] = v_scale[l];
        sc[2 * l] = k_scale[l]; sc[2 * l + 1] = v_scale[l];
    }
    // synchronous on purpose: a set-up call (once per engine), and `sc` is a stack buffer
    if (L > 0 && (hipStreamSynchronize((hipStream_t)stream) != hipSuccess ||
                  hipMemcpy(t->kv_scale_dev, sc.data(), sc.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) {
        omni_set_error("omni_talker_set_kv_scales: upload failed");
        return OMNI_EHIP;
    }
#ifdef OMNI_DEBUG_HOOKS
    if (t->d.moe_experts == 0) TRY(k_bb_all_set_scales(t->bb_table, L, k_scale, v_scale, stream));
#endif
    return OMNI_OK;
}
extern "C" void* omni_talker_attn_out(omni_talker* t) { return t ? t->attn_out : nullptr; }
extern "C" void* omni_talker_mlp_out(omni_talker* t) { return t ? t->mlp_out : nullptr; }

static int check_io(const omni_talker* t, const omni_step_io* io) {
    OMNI_CHECK_ARG(t && io, "omni_talker: null engine/io");
    OMNI_CHECK_ARG(io->B >= 1 && io->B <= t->Bm, "omni_talker: B=%d outside 1..%d", io->B, t->Bm);
    OMNI_CHECK_ARG(io->input_ids && io->positions && io->seq_lens && io->block_table && io->slot_mapping &&
                       io->last_hidden && io->text_step && io->inputs_embeds && io->audio_codes && io->logits,
                   "omni_talker: null io buffer");
    return OMNI_OK;
}

// value written for masked-out logits (desc.masked_logit: 0 = -inf, the Omni talker's -1e9)
static inline float mask_fill(const omni_talker* t) { return t->d.masked_logit != 0.0f ? t->d.masked_logit : -INFINITY; }

// dense gate_up weights (layer.wgu): fragment-major engines hold them gate / up interleaved by 8 rows (OMNI_EPI_SILU_MUL_GU8)
static inline int silu_epi(const omni_talker* t) { return t->d.frag_layout ? OMNI_EPI_SILU_MUL_GU8 : OMNI_EPI_SILU_MUL; }

// ---- fused-or-fallback building blocks --------------------------------------------------------
// separate-norm path (tensor-parallel ranks, prefill rows): out = epilogue( rmsnorm(resid_in (+delta)) . W^T );
// r = resid_in + delta -> resid_out
static int norm_gemm(omni_talker* t, const uint16_t* resid_in, const uint16_t* delta, uint16_t* resid_out, const void* norm_w,
                     uint16_t* normed_scratch, void* normed_out, const void* w, void* out, int rows, int N, int K, int epi,
                     const uint8_t* mask, int out_frag, void* st, const int32_t* out_live = nullptr) {
    const float eps = t->d.eps;
    const int F = t->d.frag_layout;
    OMNI_CHECK_ARG(resid_out || delta == nullptr, "norm_gemm: delta without resid_out");
    if (F) {
        // normalised rows go out fragment-major for the GEMM (and row-major too when the caller wants them; rows past
        // *out_live keep what normed_out held)
        TRY(k_rmsnorm(nullptr, delta, resid_in, resid_out, norm_w, normed_out, normed_scratch, rows, K, eps, st, out_live));
        return k_gemm_bf16_ex(normed_scratch, K, w, nullptr, out, rows, N, K, epi, mask,
                              OMNI_LAYOUT_W_FRAG | OMNI_LAYOUT_X_FRAG | (out_frag ? OMNI_LAYOUT_OUT_FRAG : 0), st, mask_fill(t));
    }
    if (normed_out && out_live) {
        // row-major engine with a gated copy: the GEMM reads the scratch rows, the caller's buffer gets the live rows only
        TRY(k_rmsnorm(nullptr, delta, resid_in, resid_out, norm_w, normed_scratch, nullptr, rows, K, eps, st));
        TRY(k_rmsnorm(delta ? resid_out : resid_in, nullptr, nullptr, nullptr, norm_w, normed_out, nullptr, rows, K, eps, st, out_live));
        return k_gemm_bf16_ex(normed_scratch, K, w, nullptr, out, rows, N, K, epi, mask, 0, st, mask_fill(t));
    }
    uint16_t* nx = normed_out ? reinterpret_cast<uint16_t*>(normed_out) : normed_scratch;
    TRY(k_rmsnorm(nullptr, delta, resid_in, resid_out, norm_w, nx, nullptr, rows, K, eps, st));
    return k_gemm_bf16_ex(nx, K, w, nullptr, out, rows, N, K, epi, mask, 0, st, mask_fill(t));
}

// plain GEMM on an activation buffer produced by one of our kernels (fragment-major when the engine runs that layout)
static int act_gemm(omni_talker* t, const void* x, const void* w, const void* bias, void* out, int rows, int N, int K, void* st) {
    const int F = t->d.frag_layout;
    return omni_gemm_bf16_ex(x, K, w, bias, out, rows, N, K, OMNI_EPI_BF16, nullptr,
                             F ? (OMNI_LAYOUT_W_FRAG | OMNI_LAYOUT_X_FRAG) : 0, st);
}

// ---- fused-norm building blocks (desc.fused_norm): the residual stream r stays fragment-major, every RMSNorm is folded
// into the GEMM that consumes it (omni_gemm_xnorm) and every residual add into the GEMM that produces it (omni_gemm_resid)
static int xnorm_gemm(omni_talker* t, const uint16_t* r, const float* part, int np, const void* norm_w, void* normed_out,
                      const void* w, void* out, int rows, int N, int K, int epi, const uint8_t* mask, int out_frag, void* st,
                      const int32_t* num_live = nullptr) {
    return k_gemm_xnorm(r, part, np, norm_w, t->d.eps, normed_out, w, out, rows, N, K, epi, mask, out_frag, 64, st, mask_fill(t),
                        num_live);
}
static int resid_gemm(const void* x_frag, const void* w, uint16_t* r, float* part, int rows, int N, int K, void* st) {
    return omni_gemm_resid(x_frag, K, w, nullptr, r, 1, part, nullptr, rows, N, K, OMNI_LAYOUT_W_FRAG | OMNI_LAYOUT_X_FRAG, st);
}

// code-predictor pass at position p on the fused stream (r = t->cp_resid, slabs = t->cp_part); returns the slab count
static int cp_forward_fused(omni_talker* t, int B, int p, int* np, void* st) {
    const omni_talker_desc& d = t->d;
    const int Hc = d.cp_hidden, hq = d.cp_q_heads, hkv = d.cp_kv_heads, D = d.cp_head_dim;
    const int Bm = t->Bm;
    const float sm = 1.0f / sqrtf((float)D);
    for (int l = 0; l < d.cp_layers; ++l) {
        const omni_layer_weights& w = t->cp_layer[l];
        TRY(xnorm_gemm(t, t->cp_resid, t->cp_part, *np, w.ln1, nullptr, w.wqkv, t->cp_qkv, B, (hq + 2 * hkv) * D, Hc,
                       OMNI_EPI_BF16, nullptr, 0, st));
        TRY(k_attn_decode_fused(t->cp_qkv, w.qnorm, w.knorm, t->cp_pos + (size_t)p * Bm, d.cp_cos_sin, d.eps, t->cp_k[l],
                                t->cp_v[l], nullptr, nullptr, t->cp_bt, 1, t->cp_seq + (size_t)p * Bm, nullptr,
                                t->cp_attn, nullptr, B, hq, hkv, D, t->cp_bs, OMNI_KV_BF16, 1.f, 1.f, sm, t->cp_bs, 1,
                                (p < 16 && t->cp_bs >= 16 && (hq / hkv == 1 || hq / hkv == 2 || hq / hkv == 4)) ? p : -1, st));
        if (p == 0 && l == d.cp_layers - 1) break;
        TRY(resid_gemm(t->cp_attn, w.wo, t->cp_resid, t->cp_part, B, Hc, hq * D, st));
        *np = Hc / 16;
        TRY(xnorm_gemm(t, t->cp_resid, t->cp_part, *np, w.ln2, nullptr, w.wgu, t->cp_act, B, d.cp_inter, Hc, silu_epi(t),
                       nullptr, 1, st));
        TRY(resid_gemm(t->cp_act, w.wdown, t->cp_resid, t->cp_part, B, Hc, d.cp_inter, st));
    }
    return OMNI_OK;
}

// positions 0 and 1 of the code predictor as ONE pass over a two-block stream: rows [0, B) = position 0 (the talker's last
// hidden state), rows [Bp, Bp + B) = position 1 (the layer-0 code embedding), Bp = B rounded up to 16; slabs 128 rows wide.
// Both inputs exist when the predictor starts, and a row's GEMM result does not depend on the rows beside it: 25 launches
// instead of 22 + 25.  Position 0 only feeds its K / V (written by the pair attention); its rows ride along.
OMNI_KNOB g_cp_pair01 = 1;
#ifdef OMNI_DEBUG_HOOKS
extern "C" void omni_debug_cp_pair01(int on) { g_cp_pair01 = on; }
#endif
static bool cp_pair01_ok(const omni_talker* t, int B) {
    const omni_talker_desc& d = t->d;
    const int Bp = (B + 15) & ~15;
    return g_cp_pair01 && d.cp_fused_norm && d.num_code_groups >= 2 && Bp + B <= 128 && d.cp_head_dim == 128 && t->cp_bs >= 2;
}
static int cp_forward_pair01(omni_talker* t, int B, int* np, void* st) {
    const omni_talker_desc& d = t->d;
    const int Hc = d.cp_hidden, hq = d.cp_q_heads, hkv = d.cp_kv_heads, D = d.cp_head_dim;
    const int Bp = (B + 15) & ~15, M2 = Bp + B;
    const int lay = OMNI_LAYOUT_W_FRAG | OMNI_LAYOUT_X_FRAG;
    for (int l = 0; l < d.cp_layers; ++l) {
        const omni_layer_weights& w = t->cp_layer[l];
        TRY(k_gemm_xnorm(t->cp_resid, t->cp_part, *np, w.ln1, d.eps, nullptr, w.wqkv, t->cp_qkv, M2, (hq + 2 * hkv) * D, Hc,
                         OMNI_EPI_BF16, nullptr, 0, 128, st));
        TRY(k_attn_pair01(t->cp_qkv, Bp, w.qnorm, w.knorm, d.cp_cos_sin, d.eps, t->cp_k[l], t->cp_v[l], t->cp_attn, B, hq, hkv,
                          t->cp_bs, 1.0f / sqrtf((float)D), st));
        TRY(k_gemm_resid(t->cp_attn, hq * D, w.wo, nullptr, t->cp_resid, 1, t->cp_part, nullptr, M2, Hc, hq * D, lay, 128, st));
        *np = Hc / 16;
        TRY(k_gemm_xnorm(t->cp_resid, t->cp_part, *np, w.ln2, d.eps, nullptr, w.wgu, t->cp_act, M2, d.cp_inter, Hc, silu_epi(t),
                         nullptr, 1, 128, st));
        TRY(k_gemm_resid(t->cp_act, d.cp_inter, w.wdown, nullptr, t->cp_resid, 1, t->cp_part, nullptr, M2, Hc, d.cp_inter, lay, 128, st));
    }
    return OMNI_OK;
}

// small_to_mtp_projection of row-major rows into the fused stream
static int cp_project_fused(omni_talker* t, const void* rows, int B, int* np, void* st) {
    const omni_talker_desc& d = t->d;
    if (d.has_cp_projection) {
        *np = d.cp_hidden / 16;
        return omni_gemm_resid(rows, d.hidden, d.cp_proj_w, d.cp_proj_b, t->cp_resid, 0, t->cp_part, nullptr, B, d.cp_hidden,
                               d.hidden, OMNI_LAYOUT_W_FRAG, st);
    }
    *np = 1;
    return k_gather_frag(nullptr, 0, rows, t->cp_resid, t->cp_part, B, d.hidden, 0, st);
}

// ---- one code-predictor forward pass at buffer position p (input rows = residual stream in t->cp_resid)
// leaves the last layer's MLP output in t->cp_mlp (the final norm is fused into the lm_head GEMM)
static int cp_forward(omni_talker* t, int B, int p, void* st) {
    const omni_talker_desc& d = t->d;
    const int Hc = d.cp_hidden, hq = d.cp_q_heads, hkv = d.cp_kv_heads, D = d.cp_head_dim;
    const int Bm = t->Bm;
    const float sm = 1.0f / sqrtf((float)D);
    for (int l = 0; l < d.cp_layers; ++l) {
        const omni_layer_weights& w = t->cp_layer[l];
        // residual stream ping-pongs cp_resid -> cp_resid_b (attention half) -> cp_resid (MLP half)
        TRY(norm_gemm(t, t->cp_resid, l == 0 ? nullptr : t->cp_mlp, t->cp_resid_b, w.ln1, t->cp_normed, nullptr, w.wqkv,
                      t->cp_qkv, B, (hq + 2 * hkv) * D, Hc, OMNI_EPI_BF16, nullptr, 0, st));
        TRY(k_attn_decode_fused(t->cp_qkv, w.qnorm, w.knorm, t->cp_pos + (size_t)p * Bm, d.cp_cos_sin, d.eps, t->cp_k[l],
                                t->cp_v[l], nullptr, nullptr, t->cp_bt, 1, t->cp_seq + (size_t)p * Bm, nullptr,
                                t->cp_attn, nullptr, B, hq, hkv, D, t->cp_bs, OMNI_KV_BF16, 1.f, 1.f, sm, t->cp_bs,
                                d.frag_layout, -1, st));
        // position 0 only feeds later positions through its K/V: nothing after the last layer's KV write is used
        if (p == 0 && l == d.cp_layers - 1) break;
        TRY(act_gemm(t, t->cp_attn, w.wo, nullptr, t->cp_o, B, Hc, hq * D, st));
        TRY(norm_gemm(t, t->cp_resid_b, t->cp_o, t->cp_resid, w.ln2, t->cp_normed, nullptr, w.wgu, t->cp_act, B, d.cp_inter,
                      Hc, silu_epi(t), nullptr, d.frag_layout, st));
        TRY(act_gemm(t, t->cp_act, w.wdown, nullptr, t->cp_mlp, B, Hc, d.cp_inter, st));
    }
    return OMNI_OK;
}

// small_to_mtp_projection of arbitrary rows into the residual stream (code_predictor_vllm.py:528-529)
static int cp_project(omni_talker* t, const void* rows /*bf16 [B,H]*/, int B, void* st) {
    const omni_talker_desc& d = t->d;
    if (d.has_cp_projection)
        return omni_gemm_bf16_ex(rows, d.hidden, d.cp_proj_w, d.cp_proj_b, t->cp_resid, B, d.cp_hidden, d.hidden, OMNI_EPI_BF16,
                                 nullptr, d.frag_layout ? OMNI_LAYOUT_W_FRAG : 0, st);
    hipError_t e = hipMemcpyAsync(t->cp_resid, rows, (size_t)B * d.hidden * 2, hipMemcpyDeviceToDevice, (hipStream_t)st);
    if (e != hipSuccess) { omni_set_error("cp_project: memcpy: %s", hipGetErrorString(e)); return OMNI_EHIP; }
    return OMNI_OK;
}

// codes int32 [B,Q] in t->codes (column 0 untouched); cp_logits fp32 [B,Q-1,codebook] optional.
// layer0_ids != NULL and d.cp_e0_table: position-1 input is gathered from the folded table instead of projected.
static int run_code_predictor(omni_talker* t, const int32_t* layer0_ids, const void* layer0_embed, const void* last_hidden,
                              int B, int greedy, float temperature, int top_k, float top_p, uint32_t seed, int32_t* steps,
                              float* cp_logits_out, void* st, const uint32_t* row_seed = nullptr,
                              const omni_chain_tail* tail = nullptr /* the step's input assembly + first qkv ride with the all-pass launch; t->tail_fused says whether they did */) {
    t->tail_fused = false;
    const omni_talker_desc& d = t->d;
    const int Q = d.num_code_groups, Hc = d.cp_hidden;
    if (Q <= 1) return OMNI_OK;
    // the sub-step sampling parameters are the model's (talker_mtp hard-codes them, qwen3_tts_talker.py:1620-1627); only the
    // noise stream is per request
    omni_row_sampling cp_rows{};
    cp_rows.seed = row_seed;
    const omni_row_sampling* cpr = row_seed ? &cp_rows : nullptr;
    if (d.cp_fused_norm) {
        int np = 1;
        bool pair_chain = false;
        const bool pair = cp_pair01_ok(t, B);
        const int Bp = (B + 15) & ~15;
        // rows of the stream / slabs that position 1 (and the head GEMM after it) uses: behind the position-0 block in the pair pass
        uint16_t* r1 = pair ? t->cp_resid + (size_t)Bp * Hc : t->cp_resid;
        float* part1 = pair ? t->cp_part + Bp : t->cp_part;
        const int ps = pair ? 128 : 64;
        if (pair) {
            if (d.has_cp_projection) {
                np = Hc / 16;
                TRY(k_gemm_resid(last_hidden, d.hidden, d.cp_proj_w, d.cp_proj_b, t->cp_resid, 0, t->cp_part, nullptr, B, Hc, d.hidden,
                                 OMNI_LAYOUT_W_FRAG, 128, st));
                if (layer0_ids && d.cp_e0_table)
                    TRY(k_gather_frag(layer0_ids, 1, d.cp_e0_table, r1, part1, B, Hc, d.vocab, st, np, 128));
                else
                    TRY(k_gemm_resid(layer0_embed, d.hidden, d.cp_proj_w, d.cp_proj_b, r1, 0, part1, nullptr, B, Hc, d.hidden,
                                     OMNI_LAYOUT_W_FRAG, 128, st));
            } else {
                np = 1;
                TRY(k_gather_frag(nullptr, 0, last_hidden, t->cp_resid, t->cp_part, B, d.hidden, 0, st, 1, 128));
                if (layer0_ids && d.cp_e0_table)
                    TRY(k_gather_frag(layer0_ids, 1, d.cp_e0_table, r1, part1, B, Hc, d.vocab, st, 1, 128));
                else
                    TRY(k_gather_frag(nullptr, 0, layer0_embed, r1, part1, B, d.hidden, 0, st, 1, 128));
            }
            if (d.cp_chain && !t->chain_half && k_cp_pair_supported(d, B, greedy, top_k, top_p)) {
                // positions 0 / 1 (25 stages) + group 1's head GEMM and sampler as ONE persistent launch (cp_chain.hip cp_pair_kernel)
                omni_chain_head hd{};
                hd.logits = cp_logits_out ? cp_logits_out : t->cp_logits;
                hd.logits_ld = cp_logits_out ? (Q - 1) * d.codebook : d.codebook;
                hd.logits_pass = cp_logits_out ? d.codebook : 0;
                hd.greedy = greedy; hd.top_k = top_k; hd.temperature = temperature; hd.top_p = top_p; hd.seed = seed;
                hd.steps = steps; hd.row_seed = row_seed; hd.codes = t->codes;
                TRY(k_cp_pair(d, t->cp_layer.data(), t->cp_k.data(), t->cp_v.data(), B, np, t->cp_resid, t->cp_part, t->cp_qkv, t->cp_attn, t->cp_act,
                              t->chain_flags, reinterpret_cast<int32_t*>(t->chain_flags + 320), &hd, st));
                t->ran |= 1;
                pair_chain = true;
                np = 1;
            } else
            TRY(cp_forward_pair01(t, B, &np, st));
        } else {
            TRY(cp_project_fused(t, last_hidden, B, &np, st));
            TRY(cp_forward_fused(t, B, 0, &np, st));
            if (layer0_ids && d.cp_e0_table) {
                TRY(k_gather_frag(layer0_ids, 1, d.cp_e0_table, t->cp_resid, t->cp_part, B, Hc, d.vocab, st));
                np = 1;
            } else {
                TRY(cp_project_fused(t, layer0_embed, B, &np, st));
            }
        }
        uint32_t* cflags = t->chain_flags;
        int32_t* cerr = reinterpret_cast<int32_t*>(t->chain_flags + 320);
        for (int g = pair_chain ? 2 : 1; g < Q; ++g) {
            const bool in_pair = pair && g == 1;          // position 1 was computed by the pair pass
            const uint16_t* head = reinterpret_cast<const uint16_t*>(d.cp_lm_head) + (size_t)(g - 1) * d.codebook * Hc;
            if (!in_pair && d.cp_chain && !t->chain_half && k_cp_chain_all_supported(d, g, greedy, top_k, top_p)) {
                // passes g .. Q - 1 -- layer stacks, head GEMMs, samplers, input gathers -- as ONE persistent launch
                omni_chain_head hd{};
                hd.logits = cp_logits_out ? cp_logits_out : t->cp_logits;
                hd.logits_ld = cp_logits_out ? (Q - 1) * d.codebook : d.codebook;
                hd.logits_pass = cp_logits_out ? d.codebook : 0;
                hd.greedy = greedy; hd.top_k = top_k; hd.temperature = temperature; hd.top_p = top_p; hd.seed = seed;
                hd.steps = steps; hd.row_seed = row_seed; hd.codes = t->codes;
                hd.tail = (tail && k_cp_chain_tail_supported(d, B)) ? tail : nullptr;
                TRY(k_cp_chain(d, t->cp_layer.data(), t->cp_k.data(), t->cp_v.data(), B, g, Q, np, t->cp_resid, t->cp_part, t->cp_qkv,
                               t->cp_attn, t->cp_act, cflags, cerr, &hd, st));
                t->ran |= 1;
                t->tail_fused = hd.tail != nullptr;
                break;
            }
            if (!in_pair) {
                if (d.cp_chain && !t->chain_half && k_cp_chain_supported(d, g)) {
                    // the layer stack of this pass as one persistent launch: 25 stages behind flag hand-offs
                    TRY(k_cp_chain(d, t->cp_layer.data(), t->cp_k.data(), t->cp_v.data(), B, g, g + 1, np, t->cp_resid, t->cp_part, t->cp_qkv,
                                   t->cp_attn, t->cp_act, cflags, cerr, nullptr, st));
                    t->ran |= 1;
                    np = Hc / 16;
                } else {
                    TRY(cp_forward_fused(t, B, g, &np, st));
                }
            }
            TRY(k_gemm_xnorm(in_pair ? r1 : t->cp_resid, in_pair ? part1 : t->cp_part, np, d.cp_norm, d.eps, nullptr, head, t->cp_logits,
                             B, d.codebook, Hc, OMNI_EPI_F32_BF16RND, nullptr, 0, in_pair ? ps : 64, st));
            if (cp_logits_out) {
                hipError_t e = hipMemcpy2DAsync(cp_logits_out + (size_t)(g - 1) * d.codebook, (size_t)(Q - 1) * d.codebook * 4,
                                                t->cp_logits, (size_t)d.codebook * 4, (size_t)d.codebook * 4, B,
                                                hipMemcpyDeviceToDevice, (hipStream_t)st);
                if (e != hipSuccess) { omni_set_error("code_predictor: memcpy2D: %s", hipGetErrorString(e)); return OMNI_EHIP; }
            }
            const bool more = g < Q - 1;
            const uint16_t* ptab = (more && d.cp_proj_table)
                                       ? reinterpret_cast<const uint16_t*>(d.cp_proj_table) + (size_t)(g - 1) * d.codebook * Hc : nullptr;
            TRY(k_sample_gather(t->cp_logits, d.codebook, B, d.codebook, greedy, temperature, top_k, top_p, 1.0f, nullptr, seed, steps, Q,
                                g, 0, t->codes + g, Q, ptab, t->cp_resid, Hc, ptab ? t->cp_part : nullptr, st, nullptr, nullptr, cpr));
            if (ptab) np = 1;
            if (more && !ptab) {
                const uint16_t* tab = reinterpret_cast<const uint16_t*>(d.cp_embed) + (size_t)(g - 1) * d.codebook * d.hidden;
                TRY(k_embed(t->codes + g, Q, tab, t->cp_row, B, d.hidden, d.codebook, st));
                TRY(cp_project_fused(t, t->cp_row, B, &np, st));
            }
        }
        return OMNI_OK;
    }
    TRY(cp_project(t, last_hidden, B, st));
    TRY(cp_forward(t, B, 0, st));
    if (layer0_ids && d.cp_e0_table)
        TRY(k_embed(layer0_ids, 1, d.cp_e0_table, t->cp_resid, B, Hc, d.vocab, st));
    else
        TRY(cp_project(t, layer0_embed, B, st));
    for (int g = 1; g < Q; ++g) {
        TRY(cp_forward(t, B, g, st));
        const uint16_t* head = reinterpret_cast<const uint16_t*>(d.cp_lm_head) + (size_t)(g - 1) * d.codebook * Hc;
        TRY(norm_gemm(t, t->cp_resid, t->cp_mlp, t->cp_resid_b, d.cp_norm, t->cp_normed, nullptr, head, t->cp_logits, B,
                      d.codebook, Hc, OMNI_EPI_F32_BF16RND, nullptr, 0, st));
        if (cp_logits_out) {
            hipError_t e = hipMemcpy2DAsync(cp_logits_out + (size_t)(g - 1) * d.codebook, (size_t)(Q - 1) * d.codebook * 4,
                                            t->cp_logits, (size_t)d.codebook * 4, (size_t)d.codebook * 4, B,
                                            hipMemcpyDeviceToDevice, (hipStream_t)st);
            if (e != hipSuccess) { omni_set_error("code_predictor: memcpy2D: %s", hipGetErrorString(e)); return OMNI_EHIP; }
        }
        // RNG key = steps[b] * Q + g (oracle: step * Q + g); the sampled code's projected embedding row is
        // gathered straight into the residual stream when the folded table exists
        const bool more = g < Q - 1;
        const uint16_t* ptab = (more && d.cp_proj_table)
                                   ? reinterpret_cast<const uint16_t*>(d.cp_proj_table) + (size_t)(g - 1) * d.codebook * Hc : nullptr;
        TRY(k_sample_gather(t->cp_logits, d.codebook, B, d.codebook, greedy, temperature, top_k, top_p, 1.0f, nullptr, seed, steps, Q,
                            g, 0, t->codes + g, Q, ptab, t->cp_resid, Hc, nullptr, st, nullptr, nullptr, cpr));
        if (more && !ptab) {
            const uint16_t* tab = reinterpret_cast<const uint16_t*>(d.cp_embed) + (size_t)(g - 1) * d.codebook * d.hidden;
            TRY(k_embed(t->codes + g, Q, tab, t->cp_row, B, d.hidden, d.codebook, st));
            TRY(cp_project(t, t->cp_row, B, st));
        }
    }
    return OMNI_OK;
}

extern "C" int omni_talker_code_predictor(omni_talker* t, const int32_t* layer0_ids, const void* layer0_embed,
                                          const void* last_hidden, int64_t* codes, float* cp_logits, int B, int greedy,
                                          float temperature, int top_k, float top_p, uint32_t seed, const int32_t* steps,
                                          void* stream) {
    OMNI_CHECK_ARG(t && layer0_ids && layer0_embed && last_hidden && codes, "omni_talker_code_predictor: null pointer");
    OMNI_CHECK_ARG(B >= 1 && B <= t->Bm, "omni_talker_code_predictor: B=%d", B);
    const int Q = t->d.num_code_groups;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpy2DAsync(t->codes, (size_t)Q * 4, layer0_ids, 4, 4, B, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { omni_set_error("code_predictor: memcpy2D: %s", hipGetErrorString(e)); return OMNI_EHIP; }
    // explicit layer0_embed given: project it (parity entry point; the folded e0 table is the step path)
    TRY(run_code_predictor(t, nullptr, layer0_embed, last_hidden, B, greedy, temperature, top_k, top_p, seed,
                           const_cast<int32_t*>(steps), cp_logits, stream));
    hipLaunchKernelGGL(copy_i32_to_i64_kernel, dim3((B * Q + 255) / 256), dim3(256), 0, st, t->codes, codes, B * Q);
    OMNI_CHECK_LAUNCH("copy_i32_to_i64");
    return OMNI_OK;
}

// fuse_tail (the full-step callers): the input assembly below and layer 0's qkv ride as the tail of the predictor's all-pass launch where that
// launch runs and the backbone takes the chain's launch structure (t->tail_fused; run_backbone then skips its first qkv launch)
static int mtp_phase(omni_talker* t, const omni_step_io* io, void* stream, bool fuse_tail) {
    TRY(check_io(t, io));
    t->ran = 0;        // the first phase of every step, whoever drives the phases (decode_step, the tensor-parallel phase calls): ADVICE r4
    const omni_talker_desc& d = t->d;
    const int B = io->B, Q = d.num_code_groups;
    hipStream_t st = (hipStream_t)stream;
    // e0 = codec_embedding(last sampled id)  (qwen3_tts_talker.py:637-640): gathered only when no folded table
    if (!d.cp_e0_table) TRY(k_embed(io->input_ids, 1, d.embed, t->e0, B, d.hidden, d.vocab, stream));
    const omni_chain_tail tl{io->input_ids, d.embed, d.vocab, d.cp_embed, io->text_step, io->inputs_embeds, t->resid, t->part, io->audio_codes, d.hidden,
                             d.layers > 0 ? t->layer[0].wqkv : nullptr, d.layers > 0 ? t->layer[0].ln1 : nullptr, t->qkv, (d.q_heads + 2 * d.kv_heads) * d.head_dim};
    // (a tensor-parallel rank too: layer 0's qkv is column-parallel -- this rank's shard, no exchange)
    const bool want_tail = fuse_tail && d.layers > 0 && !t->chain_half && !(t->has_ar && t->bb_ar_off) &&
                           k_bb_chain_supported(d, B, t->has_ar ? &t->ar_attn : nullptr, false) && !k_bb_chain_small(d);
    TRY(run_code_predictor(t, io->input_ids, t->e0, io->last_hidden, B, io->cp_greedy, io->cp_temperature, io->cp_top_k, io->cp_top_p,
                           io->seed, io->steps, nullptr, stream, io->rows.seed, want_tail ? &tl : nullptr));
    if (t->tail_fused) return OMNI_OK;
    hipLaunchKernelGGL(mtp_finalize_kernel, dim3(B), dim3(256), 0, st, io->input_ids, t->codes, (const uint16_t*)d.embed,
                       d.vocab, (const uint16_t*)d.cp_embed, (const uint16_t*)io->text_step, (uint16_t*)io->inputs_embeds,
                       t->resid, d.fused_norm ? t->part : nullptr, io->audio_codes, d.hidden, Q, d.codebook);
    OMNI_CHECK_LAUNCH("mtp_finalize");
    return OMNI_OK;
}
extern "C" int omni_talker_mtp(omni_talker* t, const omni_step_io* io, void* stream) { return mtp_phase(t, io, stream, false); }

// decode rows: fused norm+qkv, fused rope/kv-write/attention, o_proj
static int layer_attn_decode(omni_talker* t, int l, const omni_step_io* io, void* st, bool with_o = true, bool with_qkv = true) {
    const omni_talker_desc& d = t->d;
    const omni_layer_weights& w = t->layer[l];
    const int H = d.hidden, hq = d.q_heads, hkv = d.kv_heads, D = d.head_dim, B = io->B;
    if (!with_qkv) {}       // the qkv rows were left by the previous layer's persistent tail (moe_chain.hip moe_tail_kernel)
    else if (d.fused_norm)  // slab count: 1 after mtp_finalize, H / 16 after any residual GEMM
        TRY(xnorm_gemm(t, t->resid, t->part, l == 0 ? 1 : H / 16, w.ln1, nullptr, w.wqkv, t->qkv, B, (hq + 2 * hkv) * D, H,
                       OMNI_EPI_BF16, nullptr, 0, st));
    else
    TRY(norm_gemm(t, t->resid, l == 0 ? nullptr : t->mlp_out, t->resid_b, w.ln1, t->normed, nullptr, w.wqkv, t->qkv, B,
                  (hq + 2 * hkv) * D, H, OMNI_EPI_BF16, nullptr, 0, st));
    TRY(k_attn_decode_fused(t->qkv, w.qnorm, w.knorm, io->positions, d.cos_sin, d.eps, t->k_cache[l], t->v_cache[l],
                            t->k_scales[l], t->v_scales[l], io->block_table, d.bt_stride, io->seq_lens,
                            l == 0 ? io->slot_mapping : nullptr, t->attn, t->attn_ws, B, hq, hkv, D, d.block_size,
                            d.kv_dtype, d.k_scale, d.v_scale, 1.0f / sqrtf((float)D), d.max_model_len, d.frag_layout, -1, st,
                            io->num_live, io->rope_delta, d.rope_rows > 0 ? d.rope_rows : d.max_model_len, t->kv_scale_dev + 2 * l));
    if (!with_o) return OMNI_OK;          // the o_proj is the first stage of the layer's persistent launch (moe_chain.hip)
    if (d.fused_norm && t->has_ar) {
        // tensor-parallel rank on the norm-free stream: partial o_proj -> this rank's peer-mapped buffer (fragment-major),
        // then ONE launch sums the ranks' partials, adds into r and writes the sum(r^2) slabs
        void* mine = const_cast<void*>(t->ar_attn.data[t->ar_attn.rank]);
        TRY(omni_gemm_bf16_ex(t->attn, hq * D, w.wo, nullptr, mine, B, H, hq * D, OMNI_EPI_BF16, nullptr,
                              OMNI_LAYOUT_W_FRAG | OMNI_LAYOUT_X_FRAG | OMNI_LAYOUT_OUT_FRAG, st));
        return omni_allreduce_resid(&t->ar_attn, t->resid, 1, t->part, 64, nullptr, B, H, st);
    }
    if (d.fused_norm) return resid_gemm(t->attn, w.wo, t->resid, t->part, B, H, hq * D, st);
    TRY(act_gemm(t, t->attn, w.wo, nullptr, t->attn_out, B, H, hq * D, st));
    return OMNI_OK;
}

// prefill / mixed rows (correctness path): separate rope + KV write, causal attention through the cache
static int layer_attn_prefill(omni_talker* t, int l, int rows, const int32_t* positions, const int64_t* slots,
                              const int32_t* block_table, const int32_t* req_of_tok, void* st) {
    const omni_talker_desc& d = t->d;
    const omni_layer_weights& w = t->layer[l];
    const int H = d.hidden, hq = d.q_heads, hkv = d.kv_heads, D = d.head_dim;
    TRY(norm_gemm(t, t->resid, l == 0 ? nullptr : t->mlp_out, t->resid_b, w.ln1, t->normed, nullptr, w.wqkv, t->qkv, rows,
                  (hq + 2 * hkv) * D, H, OMNI_EPI_BF16, nullptr, 0, st));
    TRY(omni_qknorm_rope_kvwrite(t->qkv, w.qnorm, w.knorm, positions, d.cos_sin, slots, t->q, t->k_cache[l], t->v_cache[l],
                                 t->k_scales[l], t->v_scales[l], rows, hq, hkv, D, d.eps, d.kv_dtype, t->ksc_h[l], t->vsc_h[l], st));
    TRY(k_paged_attn_prefill(t->q, t->k_cache[l], t->v_cache[l], t->k_scales[l], t->v_scales[l], block_table, d.bt_stride,
                             req_of_tok, positions, t->attn, rows, hq, hkv, D, d.block_size, d.kv_dtype, t->ksc_h[l],
                             t->vsc_h[l], 1.0f / sqrtf((float)D), d.frag_layout, st));
    TRY(act_gemm(t, t->attn, w.wo, nullptr, t->attn_out, rows, H, hq * D, st));
    return OMNI_OK;
}

static int layer_mlp_rows(omni_talker* t, int l, int rows, void* st) {
    const omni_talker_desc& d = t->d;
    const omni_layer_weights& w = t->layer[l];
    if (d.moe_experts > 0) {
        // sparse-MoE MLP (Omni talker): norm -> router -> shared expert -> routed experts + combine -> mlp_out
        const int H = d.hidden, E = d.moe_experts, Is = d.moe_shared_inter;
        const int lay = OMNI_LAYOUT_W_FRAG | OMNI_LAYOUT_X_FRAG;
        TRY(k_rmsnorm(nullptr, t->attn_out, t->resid_b, t->resid, w.ln2, t->normed_rm, t->normed, rows, H, d.eps, st));
        // (the router matrix is fragment-major on fused_norm engines: their decode steps give it the fused-norm prologue)
        TRY(omni_gemm_bf16_ex(t->normed, H, w.moe_router, nullptr, t->moe_logits, rows, E, H, OMNI_EPI_BF16, nullptr,
                              OMNI_LAYOUT_X_FRAG | (d.fused_norm ? OMNI_LAYOUT_W_FRAG : 0), st));
        TRY(omni_moe_route(t->moe_logits, rows, E, d.moe_top_k, d.moe_norm_topk, t->moe_idx, t->moe_w, st));
        const void* shared = nullptr;
        if (Is > 0) {
            TRY(omni_gemm_bf16_ex(t->normed, H, w.moe_shared_gate_up, nullptr, t->act, rows, Is, H, OMNI_EPI_SILU_MUL, nullptr,
                                  lay | OMNI_LAYOUT_OUT_FRAG, st));
            TRY(omni_gemm_bf16_ex(t->act, Is, w.moe_shared_down, nullptr, t->moe_shared, rows, H, Is, OMNI_EPI_BF16, nullptr, lay, st));
            shared = t->moe_shared;
        }
        const int El = d.moe_experts_local > 0 ? d.moe_experts_local : E;
        return omni_moe_experts_ex(t->normed_rm, t->moe_idx, t->moe_w, w.moe_gate_up, d.moe_w8 ? w.moe_gate_up_scale : nullptr, w.moe_down,
                                   d.moe_w8 ? w.moe_down_scale : nullptr, shared, w.moe_shared_gate, t->moe_act, t->moe_y, t->mlp_out,
                                   rows, H, d.moe_inter, El, d.moe_e0, d.moe_top_k, st);
    }
    TRY(norm_gemm(t, t->resid_b, t->attn_out, t->resid, w.ln2, t->normed, nullptr, w.wgu, t->act, rows, d.inter, d.hidden,
                  silu_epi(t), nullptr, d.frag_layout, st));
    TRY(act_gemm(t, t->act, w.wdown, nullptr, t->mlp_out, rows, d.hidden, d.inter, st));
    return OMNI_OK;
}

// thinker -> talker projection MLP (HF Qwen3OmniMoeTalkerResizeMLP: linear_fc2(silu(linear_fc1(x))), both with bias) over T
// rows, 64 at a time through the skinny GEMM (a row's result does not depend on its launch mates); act_ws bf16 [min(T,64), I]
extern "C" int omni_resize_mlp(const void* x, const void* fc1_w, const void* fc1_b, const void* fc2_w, const void* fc2_b,
                               void* act_ws, void* out, int T, int H_in, int I, int H_out, void* stream) {
    OMNI_CHECK_ARG(x && fc1_w && fc2_w && act_ws && out, "omni_resize_mlp: null pointer");
    OMNI_CHECK_ARG(T >= 0 && H_in % 32 == 0 && I % 32 == 0 && H_out % 16 == 0, "omni_resize_mlp: T=%d H_in=%d I=%d H_out=%d", T, H_in, I, H_out);
    for (int r0 = 0; r0 < T; r0 += 64) {
        const int rows = T - r0 < 64 ? T - r0 : 64;
        const uint16_t* xr = reinterpret_cast<const uint16_t*>(x) + (size_t)r0 * H_in;
        TRY(omni_gemm_bf16(xr, H_in, fc1_w, fc1_b, act_ws, rows, I, H_in, OMNI_EPI_BF16, nullptr, stream));
        TRY(omni_silu(act_ws, act_ws, (long long)rows * I, stream));
        TRY(omni_gemm_bf16(act_ws, I, fc2_w, fc2_b, reinterpret_cast<uint16_t*>(out) + (size_t)r0 * H_out, rows, H_out, I, OMNI_EPI_BF16,
                           nullptr, stream));
    }
    return OMNI_OK;
}

extern "C" int omni_talker_layer_attn(omni_talker* t, const omni_step_io* io, int layer, void* stream) {
    TRY(check_io(t, io));
    OMNI_CHECK_ARG(layer >= 0 && layer < t->d.layers, "omni_talker_layer_attn: layer=%d", layer);
    return layer_attn_decode(t, layer, io, stream);
}

extern "C" int omni_talker_layer_mlp(omni_talker* t, const omni_step_io* io, int layer, void* stream) {
    TRY(check_io(t, io));
    OMNI_CHECK_ARG(layer >= 0 && layer < t->d.layers, "omni_talker_layer_mlp: layer=%d", layer);
    const omni_talker_desc& d = t->d;
    if (d.fused_norm && d.moe_experts > 0) {
        // sparse-MoE MLP on the norm-free stream: the router GEMM takes the fused-norm prologue and leaves the normalised rows
        // (row-major) for the expert kernels; the combine adds into r and writes the slabs -- or, on a tensor- / expert-parallel
        // rank, leaves the partial in the peer-mapped buffer for the one-shot all-reduce
        const omni_layer_weights& w = t->layer[layer];
        const int H = d.hidden, E = d.moe_experts, Is = d.moe_shared_inter, B = io->B;
        const int lay = OMNI_LAYOUT_W_FRAG | OMNI_LAYOUT_X_FRAG;
        TRY(xnorm_gemm(t, t->resid, t->part, H / 16, w.ln2, t->normed_rm, w.moe_router, t->moe_logits, B, E, H, OMNI_EPI_BF16, nullptr, 0, stream));
        TRY(omni_moe_route(t->moe_logits, B, E, d.moe_top_k, d.moe_norm_topk, t->moe_idx, t->moe_w, stream));
        const void* shared = nullptr;
        if (Is > 0) {
            TRY(xnorm_gemm(t, t->resid, t->part, H / 16, w.ln2, nullptr, w.moe_shared_gate_up, t->act, B, Is, H, OMNI_EPI_SILU_MUL, nullptr, 1, stream));
            TRY(omni_gemm_bf16_ex(t->act, Is, w.moe_shared_down, nullptr, t->moe_shared, B, H, Is, OMNI_EPI_BF16, nullptr, lay, stream));
            shared = t->moe_shared;
        }
        const int El = d.moe_experts_local > 0 ? d.moe_experts_local : E;
        void* mine = t->has_ar ? const_cast<void*>(t->ar_mlp.data[t->ar_mlp.rank]) : nullptr;
        TRY(omni_moe_experts_resid(t->normed_rm, t->moe_idx, t->moe_w, w.moe_gate_up, d.moe_w8 ? w.moe_gate_up_scale : nullptr, w.moe_down,
                                   d.moe_w8 ? w.moe_down_scale : nullptr, shared, w.moe_shared_gate, t->moe_act, t->moe_y, t->resid, t->part,
                                   mine, B, H, d.moe_inter, El, d.moe_e0, d.moe_top_k, stream));
        if (t->has_ar) return omni_allreduce_resid(&t->ar_mlp, t->resid, 1, t->part, 64, nullptr, B, H, stream);
        return OMNI_OK;
    }
    if (d.fused_norm) {
        const omni_layer_weights& w = t->layer[layer];
        TRY(xnorm_gemm(t, t->resid, t->part, d.hidden / 16, w.ln2, nullptr, w.wgu, t->act, io->B, d.inter, d.hidden,
                       silu_epi(t), nullptr, 1, stream));
        if (t->has_ar) {
            void* mine = const_cast<void*>(t->ar_mlp.data[t->ar_mlp.rank]);
            TRY(omni_gemm_bf16_ex(t->act, d.inter, w.wdown, nullptr, mine, io->B, d.hidden, d.inter, OMNI_EPI_BF16, nullptr,
                                  OMNI_LAYOUT_W_FRAG | OMNI_LAYOUT_X_FRAG | OMNI_LAYOUT_OUT_FRAG, stream));
            return omni_allreduce_resid(&t->ar_mlp, t->resid, 1, t->part, 64, nullptr, io->B, d.hidden, stream);
        }
        return resid_gemm(t->act, w.wdown, t->resid, t->part, io->B, d.hidden, d.inter, stream);
    }
    return layer_mlp_rows(t, layer, io->B, stream);
}

extern "C" int omni_talker_logits(omni_talker* t, const void* hidden, float* logits, int R, int round_bf16, void* stream) {
    OMNI_CHECK_ARG(t && hidden && logits, "omni_talker_logits: null pointer");
    const omni_talker_desc& d = t->d;
    for (int r0 = 0; r0 < R; r0 += 64) {
        const int m = R - r0 < 64 ? R - r0 : 64;
        TRY(k_gemm_bf16_ex(reinterpret_cast<const uint16_t*>(hidden) + (size_t)r0 * d.hidden, d.hidden, d.lm_head, nullptr,
                           logits + (size_t)r0 * d.vocab, m, d.vocab, d.hidden,
                           round_bf16 ? OMNI_EPI_F32_BF16RND : OMNI_EPI_F32, d.allowed_mask,
                           d.frag_layout ? OMNI_LAYOUT_W_FRAG : 0, stream, mask_fill(t)));
    }
    return OMNI_OK;
}

extern "C" int omni_talker_finish(omni_talker* t, const omni_step_io* io, void* stream) {
    TRY(check_io(t, io));
    const omni_talker_desc& d = t->d;
    const int B = io->B;
    // final norm fused into the lm_head GEMM; the normalised rows ARE h[t+1] and go straight to last_hidden
    // (postprocess, qwen3_tts_talker.py:649-655): nothing else reads last_hidden after the mtp phase of this step
    if (t->head_fused) t->head_fused = false;      // logits and h[t + 1] left by the last backbone launch (run_backbone fuse_head)
    else if (d.fused_norm)
        TRY(xnorm_gemm(t, t->resid, t->part, d.layers > 0 ? d.hidden / 16 : 1, d.final_norm, io->last_hidden, d.lm_head, io->logits,
                       B, d.vocab, d.hidden, OMNI_EPI_F32_BF16RND, d.allowed_mask, 0, stream, io->num_live));
    else
    TRY(norm_gemm(t, t->resid, t->mlp_out, t->resid_b, d.final_norm, d.frag_layout || io->num_live ? t->normed : reinterpret_cast<uint16_t*>(io->last_hidden),
                  io->last_hidden, d.lm_head, io->logits, B, d.vocab, d.hidden, OMNI_EPI_F32_BF16RND, d.allowed_mask, 0, stream, io->num_live));
    const bool any_rows = io->rows.greedy || io->rows.temperature || io->rows.top_k || io->rows.top_p || io->rows.rep_penalty || io->rows.seed;
    // the step's status words leave with its last launch (ABI v4): the chain error word, the peer all-reduce's, what ran
    const omni_step_status stt{reinterpret_cast<const int32_t*>(t->chain_flags + 320), t->has_ar ? t->ar_attn.error : nullptr, io->status, t->ran};
    TRY(k_sample(io->logits, d.vocab, B, d.vocab, io->greedy, io->temperature, io->top_k, io->top_p, io->rep_penalty, io->seen, io->seed,
                 io->steps, 1, 0, 1, io->input_ids, 1, stream, io->advance ? io->positions : nullptr,
                 io->advance ? io->seq_lens : nullptr,       // positions / seq_lens += 1 inside the sampler launch
                 any_rows ? &io->rows : nullptr, io->num_live, io->status ? &stt : nullptr));
    return OMNI_OK;
}

// the routed experts + combine of a sparse-MoE layer on the norm-free stream (the tail of omni_talker_layer_mlp's MoE branch): reads the
// normalised rows, the routing and the shared expert's output, adds into r and writes the slabs
static int moe_experts_tail(omni_talker* t, int layer, int B, void* stream) {
    const omni_talker_desc& d = t->d;
    const omni_layer_weights& w = t->layer[layer];
    const int El = d.moe_experts_local > 0 ? d.moe_experts_local : d.moe_experts;
    return omni_moe_experts_resid(t->normed_rm, t->moe_idx, t->moe_w, w.moe_gate_up, d.moe_w8 ? w.moe_gate_up_scale : nullptr, w.moe_down,
                                  d.moe_w8 ? w.moe_down_scale : nullptr, d.moe_shared_inter > 0 ? t->moe_shared : nullptr, w.moe_shared_gate, t->moe_act,
                                  t->moe_y, t->resid, t->part, nullptr, B, d.hidden, d.moe_inter, El, d.moe_e0, d.moe_top_k, stream);
}

// diagnostics: append N trivial launches after every layer phase to price a launch inside the real step
OMNI_KNOB g_extra_trivial = 0, g_bb_head = 1;      // g_bb_head: the lm_head as the last stage of the last backbone launch (0: its own launch, round 5)
__global__ void dbg_nop_kernel(int32_t* p) { if (threadIdx.x == 9999) p[0] = 0; }
#ifdef OMNI_DEBUG_HOOKS
extern "C" void omni_debug_extra_trivial(int n) { g_extra_trivial = n; }
extern "C" void omni_debug_bb_head(int on) { g_bb_head = on; }
#endif

// the backbone of one decode step (every layer; the residual stream in t->resid on entry, final residual on exit)
// parts (timing attribution, omni_talker_step_part): 2 = the attention launches, 4 = everything else of the stack; both = the stack
// fuse_head: the talker's head (final norm + lm_head) rides as the last stage of the last layer's persistent launch (bb_chain.hip, round 6) --
// omni_talker_finish then launches the sampler only (t->head_fused)
static int run_backbone(omni_talker* t, const omni_step_io* io, void* stream, int parts = 6, bool fuse_head = false) {
    const omni_talker_desc& d = t->d;
    const bool do_attn = (parts & 2) != 0, do_rest = (parts & 4) != 0;
    t->head_fused = false;
#ifdef OMNI_DEBUG_HOOKS
    if (parts == 6 && !t->chain_half && k_bb_all_supported(d, io->B, t->has_ar)) {      // A/B arm: the whole stack, attention included, as one persistent launch
        t->ran |= 2;                                                  // (a split request -- step_part 2 / 4 -- takes the two-launch structure below)
        return k_bb_all(d, t->bb_table, io, t->attn, t->resid, t->part, t->act, t->qkv, t->chain_flags,
                        reinterpret_cast<int32_t*>(t->chain_flags + 320), stream);
    }
#endif
    const omni_bb_ar bar{&t->ar_attn, &t->ar_mlp};
    if (!(t->has_ar && t->bb_ar_off) && k_bb_chain_supported(d, io->B, t->has_ar ? &t->ar_attn : nullptr, t->chain_half) && d.layers > 0) {
        t->ran |= 2;
        const bool plain = !t->has_ar && !t->chain_half;      // (a tensor-parallel rank / the half grid: the 64-row stage set only)
        // attention launches alternate with one persistent launch per layer: o_proj -> gate_up -> down_proj -> next qkv
        const int H = d.hidden, hq = d.q_heads, hkv = d.kv_heads, D = d.head_dim, B = io->B;
        if (do_rest && !t->tail_fused)      // (tail_fused: layer 0's qkv rows were left by the predictor's all-pass launch, cp_chain.hip tail)
        TRY(xnorm_gemm(t, t->resid, t->part, 1, t->layer[0].ln1, nullptr, t->layer[0].wqkv, t->qkv, B, (hq + 2 * hkv) * D, H, OMNI_EPI_BF16,
                       nullptr, 0, stream));
        t->tail_fused = false;
        for (int l = 0; l < d.layers; ++l) {
            const omni_layer_weights& w = t->layer[l];
            if (do_attn)
            TRY(k_attn_decode_fused(t->qkv, w.qnorm, w.knorm, io->positions, d.cos_sin, d.eps, t->k_cache[l], t->v_cache[l], t->k_scales[l],
                                    t->v_scales[l], io->block_table, d.bt_stride, io->seq_lens, l == 0 ? io->slot_mapping : nullptr, t->attn,
                                    t->attn_ws, B, hq, hkv, D, d.block_size, d.kv_dtype, d.k_scale, d.v_scale, 1.0f / sqrtf((float)D),
                                    d.max_model_len, d.frag_layout, -1, stream, io->num_live, io->rope_delta,
                                    d.rope_rows > 0 ? d.rope_rows : d.max_model_len, t->kv_scale_dev + 2 * l));
            if (!do_rest) continue;
#ifdef OMNI_DEBUG_HOOKS      // round-3 A/B arms (all slower): debug library only
            if (k_bb_engine_enabled())
                TRY(k_bb_engine(w, l + 1 < d.layers ? &t->layer[l + 1] : nullptr, t->attn, t->resid, t->part, t->act, t->qkv, B, d.eps,
                                t->chain_flags, reinterpret_cast<int32_t*>(t->chain_flags + 320), stream));
            else
#endif
            {
                const bool last = l + 1 == d.layers;
                const omni_bb_head hd{io->logits, io->last_hidden, io->num_live, mask_fill(t)};
                const bool with_head = last && fuse_head && !t->chain_half && g_bb_head && k_bb_chain_head_supported(d);      // (the half grid has no head stage)
                TRY(k_bb_chain(d, w, last ? nullptr : &t->layer[l + 1], t->attn, t->resid, t->part, t->act, t->qkv, B, d.eps,
                               t->chain_flags, reinterpret_cast<int32_t*>(t->chain_flags + 320), stream, plain && k_bb_chain_small(d), with_head ? &hd : nullptr,
                               t->has_ar ? &bar : nullptr, t->chain_half));
                if (with_head) t->head_fused = true;
            }
        }
        return OMNI_OK;
    }
    OMNI_CHECK_ARG(parts == 6, "omni_talker_step_part: the attention / rest split needs the backbone chain's launch structure (dense 1.7B or 0.6B shape, single rank)");
    // the Omni talker's sparse-MoE layer: o_proj -> router | shared gate_up -> shared down + routing as ONE persistent launch between the
    // attention launch and the expert GEMMs (moe_chain.hip, round 5): 10 -> 6 launches per layer, same bits
    const bool moe_chain = k_moe_chain_supported(d, io->B, t->has_ar);
    if (moe_chain) t->ran |= 2;
    for (int l = 0; l < d.layers; ++l) {
        if (moe_chain) {
            const bool tail = k_moe_tail_enabled();
            TRY(layer_attn_decode(t, l, io, stream, false, l == 0 || !tail));
            TRY(k_moe_chain(d, t->layer[l], t->attn, t->resid, t->part, t->normed_rm, t->moe_logits, t->act, t->moe_shared, t->moe_idx, t->moe_w, io->B,
                            t->chain_flags, reinterpret_cast<int32_t*>(t->chain_flags + 320), stream));
            if (tail && l + 1 < d.layers) {
                // expert GEMMs, then { combine -> the next layer's qkv } as the layer's second persistent launch
                const omni_layer_weights& w = t->layer[l];
                const int El = d.moe_experts_local > 0 ? d.moe_experts_local : d.moe_experts;
                TRY(k_moe_experts_phases(t->normed_rm, t->moe_idx, t->moe_w, w.moe_gate_up, d.moe_w8 ? w.moe_gate_up_scale : nullptr, w.moe_down,
                                         d.moe_w8 ? w.moe_down_scale : nullptr, t->moe_act, t->moe_y, io->B, d.hidden, d.moe_inter, El, d.moe_e0,
                                         d.moe_top_k, stream));
                TRY(k_moe_tail(d, w, t->layer[l + 1], t->moe_y, t->normed_rm, t->moe_shared, t->moe_idx, t->resid, t->part, t->qkv, io->B,
                               t->chain_flags, reinterpret_cast<int32_t*>(t->chain_flags + 320), stream));
            } else {
                TRY(moe_experts_tail(t, l, io->B, stream));
            }
            continue;
        }
        TRY(omni_talker_layer_attn(t, io, l, stream));
        TRY(omni_talker_layer_mlp(t, io, l, stream));
    }
    return OMNI_OK;
}

// timing attribution (bench.py roofline.families): launch only some parts of a decode step -- 1 the mtp phase (code predictor +
// input assembly), 2 the backbone's attention launches, 4 the rest of the backbone stack, 8 final norm + lm_head + sampler.
// Parts run alone read whatever the buffers hold: the TIMES are the step's (no kernel's control flow depends on activation values),
// the outputs are not.  1 | 2 | 4 | 8 = omni_talker_decode_step.
extern "C" int omni_talker_step_part(omni_talker* t, const omni_step_io* io, int parts, void* stream) {
    TRY(check_io(t, io));
    OMNI_CHECK_ARG(parts > 0 && parts < 16, "omni_talker_step_part: parts=%d", parts);
    t->ran = 0;
    if (parts & 1) TRY(mtp_phase(t, io, stream, (parts & 5) == 5));
    else t->tail_fused = false;
    if (parts & 6) TRY(run_backbone(t, io, stream, parts & 6, (parts & 12) == 12));
    else t->head_fused = false;
    if (parts & 8) TRY(omni_talker_finish(t, io, stream));
    return OMNI_OK;
}

// diagnostics / A-B timing: the backbone half of a step alone (layers + final norm + lm_head + sampler; no code predictor)
extern "C" int omni_talker_backbone_step(omni_talker* t, const omni_step_io* io, void* stream) {
    TRY(check_io(t, io));
    t->ran = 0;
    t->tail_fused = false;
    TRY(run_backbone(t, io, stream, 6, true));
    return omni_talker_finish(t, io, stream);
}

extern "C" int omni_talker_decode_step(omni_talker* t, const omni_step_io* io, void* stream) {
    TRY(mtp_phase(t, io, stream, g_extra_trivial == 0));
    if (g_extra_trivial == 0) {
        TRY(run_backbone(t, io, stream, 6, true));
        return omni_talker_finish(t, io, stream);
    }
    for (int l = 0; l < t->d.layers; ++l) {
        TRY(omni_talker_layer_attn(t, io, l, stream));
        for (int k = 0; k < g_extra_trivial; ++k)
            hipLaunchKernelGGL(dbg_nop_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, io->steps);
        TRY(omni_talker_layer_mlp(t, io, l, stream));
    }
    return omni_talker_finish(t, io, stream);
}

extern "C" int omni_talker_prefill(omni_talker* t, const void* x, const int32_t* positions, const int32_t* req_of_tok,
                                   const int64_t* slot_mapping, const int32_t* block_table, void* hidden_out, int T,
                                   void* stream) {
    OMNI_CHECK_ARG(t && x && positions && req_of_tok && slot_mapping && block_table && hidden_out,
                   "omni_talker_prefill: null pointer");
    const omni_talker_desc& d = t->d;
    hipStream_t st = (hipStream_t)stream;
    // chunks of <= max_batch tokens through the skinny-GEMM path; KV of earlier chunks is already in the
    // cache when later chunks attend (causal), so chunking does not change results.
    for (int t0 = 0; t0 < T; t0 += t->Bm) {
        const int rows = T - t0 < t->Bm ? T - t0 : t->Bm;
        hipError_t e = hipMemcpyAsync(t->resid, reinterpret_cast<const uint16_t*>(x) + (size_t)t0 * d.hidden,
                                      (size_t)rows * d.hidden * 2, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { omni_set_error("prefill: memcpy: %s", hipGetErrorString(e)); return OMNI_EHIP; }
        for (int l = 0; l < d.layers; ++l) {
            TRY(layer_attn_prefill(t, l, rows, positions + t0, slot_mapping + t0, block_table, req_of_tok + t0, stream));
            TRY(layer_mlp_rows(t, l, rows, stream));
        }
        TRY(omni_rmsnorm(nullptr, t->mlp_out, t->resid, d.final_norm,
                         reinterpret_cast<uint16_t*>(hidden_out) + (size_t)t0 * d.hidden, rows, d.hidden, d.eps, stream));
    }
    return OMNI_OK;
}

// ---- the prefill path one phase at a time (tensor-parallel hosts all-reduce between the phases)
extern "C" int omni_talker_rows_begin(omni_talker* t, const void* x, int rows, void* stream) {
    OMNI_CHECK_ARG(t && x && rows >= 1 && rows <= t->Bm, "omni_talker_rows_begin: bad arguments");
    hipError_t e = hipMemcpyAsync(t->resid, x, (size_t)rows * t->d.hidden * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { omni_set_error("rows_begin: memcpy: %s", hipGetErrorString(e)); return OMNI_EHIP; }
    return OMNI_OK;
}
extern "C" int omni_talker_rows_attn(omni_talker* t, int layer, int rows, const int32_t* positions, const int64_t* slot_mapping,
                                     const int32_t* block_table, const int32_t* req_of_tok, void* stream) {
    OMNI_CHECK_ARG(t && positions && slot_mapping && block_table && req_of_tok && rows >= 1 && rows <= t->Bm &&
                       layer >= 0 && layer < t->d.layers, "omni_talker_rows_attn: bad arguments");
    return layer_attn_prefill(t, layer, rows, positions, slot_mapping, block_table, req_of_tok, stream);
}
extern "C" int omni_talker_rows_mlp(omni_talker* t, int layer, int rows, void* stream) {
    OMNI_CHECK_ARG(t && rows >= 1 && rows <= t->Bm && layer >= 0 && layer < t->d.layers, "omni_talker_rows_mlp: bad arguments");
    return layer_mlp_rows(t, layer, rows, stream);
}
extern "C" int omni_talker_rows_end(omni_talker* t, void* hidden_out, int rows, void* stream) {
    OMNI_CHECK_ARG(t && hidden_out && rows >= 1 && rows <= t->Bm, "omni_talker_rows_end: bad arguments");
    return omni_rmsnorm(nullptr, t->mlp_out, t->resid, t->d.final_norm, hidden_out, rows, t->d.hidden, t->d.eps, stream);
}
