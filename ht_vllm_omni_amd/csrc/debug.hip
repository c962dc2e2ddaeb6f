// Micro-benchmarks of launch / dependency floors (diagnostics only; not part of the talker path).
#include "common.cuh"

__global__ void dbg_empty_kernel() {}
__global__ void dbg_touch_kernel(float* p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    p[i] = p[i] + 1.0f;
}
// `depth` dependent loads per thread (pointer chase through idx[])
__global__ void dbg_chase_kernel(const int* idx, int* out, int depth) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    for (int d = 0; d < depth; ++d) j = idx[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = j;
}

__global__ __launch_bounds__(512) void dbg_lds_kernel(float* p) {
    extern __shared__ float sm[];
    if (p && threadIdx.x == 9999) p[0] = sm[0];
}
// many live VGPRs, no memory traffic
template <int NV>
__global__ __launch_bounds__(512) void dbg_vgpr_kernel(float* p, int n) {
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0001f + v[(i + 1) % NV];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    if (s == 12345.678f) p[0] = s;
}

// distinct trivial kernels (different code objects) to separate code-fetch cost from data coldness
template <int ID>
__global__ void dbg_distinct_kernel(float* p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = p[i + ID * 4096];
#pragma unroll
    for (int k = 0; k < 8 + ID; ++k) v = v * 1.0001f + (float)(ID + k);
    p[i + ID * 4096] = v;
}
// streams `n` floats (L2 thrash between the small kernels)
__global__ void dbg_stream_kernel(const float4* __restrict__ src, float* dst, size_t n4) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) dst[0] = acc;
}
extern "C" int omni_debug_mix(int pattern, float* small, const void* big, size_t big_bytes, int reps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    for (int r = 0; r < reps; ++r) {
        if (pattern & 2) hipLaunchKernelGGL(dbg_stream_kernel, dim3(2048), dim3(256), 0, st, (const float4*)big, small, big_bytes / 16);
        if (pattern & 1) {
            hipLaunchKernelGGL(dbg_distinct_kernel<0>, dim3(16), dim3(256), 0, st, small);
            hipLaunchKernelGGL(dbg_distinct_kernel<1>, dim3(16), dim3(256), 0, st, small);
            hipLaunchKernelGGL(dbg_distinct_kernel<2>, dim3(16), dim3(256), 0, st, small);
            hipLaunchKernelGGL(dbg_distinct_kernel<3>, dim3(16), dim3(256), 0, st, small);
        } else if (!(pattern & 4)) {
            for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(dbg_distinct_kernel<0>, dim3(16), dim3(256), 0, st, small);
        }
    }
    OMNI_CHECK_LAUNCH("omni_debug_mix");
    return OMNI_OK;
}

__global__ __launch_bounds__(512) void dbg_cfg_a_kernel(float* p) {   // 512 threads, dynamic LDS
    extern __shared__ float sm[];
    sm[threadIdx.x] = p[threadIdx.x];
    __syncthreads();
    p[blockIdx.x * 512 + threadIdx.x] = sm[(threadIdx.x + 1) & 511];
}
__global__ __launch_bounds__(256) void dbg_cfg_b_kernel(float* p) {   // 256 threads, no LDS
    p[blockIdx.x * 256 + threadIdx.x] += 1.0f;
}
__global__ __launch_bounds__(64) void dbg_cfg_c_kernel(float* p) {    // 1 wave
    p[threadIdx.x] += 1.0f;
}
// alternate kernels with different launch configurations (threads / LDS size / grid)
extern "C" int omni_debug_cfgmix(int mode, float* p, int reps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    for (int r = 0; r < reps; ++r) {
        if (mode == 0) { for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(dbg_cfg_b_kernel, dim3(64), dim3(256), 0, st, p); }
        else if (mode == 1) { for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(dbg_cfg_a_kernel, dim3(256), dim3(512), 65536, st, p); }
        else if (mode == 2) {
            hipLaunchKernelGGL(dbg_cfg_a_kernel, dim3(256), dim3(512), 65536, st, p);
            hipLaunchKernelGGL(dbg_cfg_b_kernel, dim3(64), dim3(256), 0, st, p);
            hipLaunchKernelGGL(dbg_cfg_a_kernel, dim3(256), dim3(512), 32768, st, p);
            hipLaunchKernelGGL(dbg_cfg_c_kernel, dim3(1), dim3(64), 0, st, p);
        } else {
            hipLaunchKernelGGL(dbg_cfg_a_kernel, dim3(256), dim3(512), 65536, st, p);
            hipLaunchKernelGGL(dbg_cfg_a_kernel, dim3(256), dim3(512), 32768, st, p);
            hipLaunchKernelGGL(dbg_cfg_a_kernel, dim3(128), dim3(512), 16384, st, p);
            hipLaunchKernelGGL(dbg_cfg_a_kernel, dim3(384), dim3(512), 65536, st, p);
        }
    }
    OMNI_CHECK_LAUNCH("omni_debug_cfgmix");
    return OMNI_OK;
}

extern "C" int omni_debug_launch(int mode, int blocks, int threads, void* p0, void* p1, int arg, int reps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    for (int r = 0; r < reps; ++r) {
        if (mode == 0) hipLaunchKernelGGL(dbg_empty_kernel, dim3(blocks), dim3(threads), 0, st);
        else if (mode == 1) hipLaunchKernelGGL(dbg_touch_kernel, dim3(blocks), dim3(threads), 0, st, (float*)p0);
        else if (mode == 3) hipLaunchKernelGGL(dbg_lds_kernel, dim3(blocks), dim3(threads), (size_t)arg, st, (float*)nullptr);
        else if (mode == 4) hipLaunchKernelGGL(dbg_vgpr_kernel<32>, dim3(blocks), dim3(threads), 0, st, (float*)p0, arg);
        else if (mode == 5) hipLaunchKernelGGL(dbg_vgpr_kernel<120>, dim3(blocks), dim3(threads), 0, st, (float*)p0, arg);
        else hipLaunchKernelGGL(dbg_chase_kernel, dim3(blocks), dim3(threads), 0, st, (const int*)p0, (int*)p1, arg);
    }
    OMNI_CHECK_LAUNCH("omni_debug_launch");
    return OMNI_OK;
}

// ---- which XCD does block b of consecutive launches land on?  (the L2 warm-up chain of gemm.hip assumes: the same one
// for equal b mod 8 as long as every grid in between is a multiple of 8 workgroups)
__global__ __launch_bounds__(512) void dbg_xcc_kernel(int32_t* out) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x + gridDim.x * blockIdx.y] = (int)(x & 0xF);
    }
}
// reps probe launches of (gx, gy) blocks into out[r * gx * gy ...]; between them `odd` blocks of an unrelated kernel
extern "C" int omni_debug_xcc_probe(int32_t* out, float* scratch, int gx, int gy, int odd, int reps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(dbg_xcc_kernel, dim3(gx, gy), dim3(512), 0, st, out + (size_t)r * gx * gy);
        if (odd > 0) hipLaunchKernelGGL(dbg_touch_kernel, dim3(odd), dim3(256), 0, st, scratch);
    }
    OMNI_CHECK_LAUNCH("omni_debug_xcc_probe");
    return OMNI_OK;
}

// ---- dependent chain of small kernels whose first instruction needs a kernel argument: by-value struct (s_load from the
// kernarg segment) vs leading scalar arguments (this file is built with -amdgpu-kernarg-preload-count=16: they arrive in
// SGPRs with the wave)
struct DbgChainArgs { const float* in; float* out; int n; float s; int pad[24]; };
__global__ __launch_bounds__(512) void dbg_chain_struct_kernel(const DbgChainArgs a) {
    const int i = blockIdx.x * 512 + threadIdx.x;
    if (i < a.n) a.out[i] = a.in[i] * a.s + 1.0f;
}
__global__ __launch_bounds__(512) void dbg_chain_scalar_kernel(const float* in, float* out, int n, float s) {
    const int i = blockIdx.x * 512 + threadIdx.x;
    if (i < n) out[i] = in[i] * s + 1.0f;
}
extern "C" int omni_debug_chain(int mode, float* a, float* b, int blocks, int reps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    for (int r = 0; r < reps; ++r) {
        float* src = (r & 1) ? b : a;
        float* dst = (r & 1) ? a : b;
        if (mode == 0) {
            DbgChainArgs x{};
            x.in = src; x.out = dst; x.n = blocks * 512; x.s = 0.5f;
            hipLaunchKernelGGL(dbg_chain_struct_kernel, dim3(blocks), dim3(512), 0, st, x);
        } else {
            hipLaunchKernelGGL(dbg_chain_scalar_kernel, dim3(blocks), dim3(512), 0, st, (const float*)src, dst, blocks * 512, 0.5f);
        }
    }
    OMNI_CHECK_LAUNCH("omni_debug_chain");
    return OMNI_OK;
}

// ---- what does a grid barrier cost inside ONE persistent launch?  (The alternative to a kernel boundary for the code predictor's
// chain of dependent steps.)  `blocks` workgroups of 512 threads, all co-resident (blocks <= 256, one per CU), run `iters` steps of
// { out[i] = in[j] * s + 1 with j on ANOTHER workgroup's slice (a real cross-XCD hand-off), grid barrier }.
//   mode 0: one counter: release-add at agent scope, every workgroup's thread 0 polls it
//   mode 1: as 0, the pollers s_sleep between polls
//   mode 2: hierarchical: one counter per XCD (its 32 workgroups), the last arriver of an XCD adds to the global one; the pollers
//           read the global counter
//   mode 3 / 4: as 0 / 2 with the exchanged values moved by sc1 (L2-bypassing) stores and loads instead of release / acquire fences
// Bounded spins: a workgroup that never sees its peers sets err and moves on (a wrong number, not a hung GPU).
#define DBG_GBAR_BOUND (1u << 20)
__global__ __launch_bounds__(512) void dbg_gbar_kernel(float* a, float* b, unsigned* ctr /* [0] global, [8 + 16 x] per XCD */, int* err,
                                                       int iters, int mode) {
    const int nwg = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    // workgroups of this XCD: ids congruent mod 8 (round-robin dealing), so nwg / 8 each when nwg % 8 == 0
    const unsigned per_xcd = (unsigned)nwg >> 3;
    for (int it = 0; it < iters; ++it) {
        const float* src = (it & 1) ? b : a;
        float* dst = (it & 1) ? a : b;
        const int j = ((w + 37) % nwg) * 512 + tid;                       // a slice some other workgroup (other XCD) wrote last step
        if (mode >= 3) {
            // coherent-by-access: the exchanged values bypass the XCD-private L2 in both directions (sc1 store / sc1 load), no cache
            // write-back or invalidate at the barrier -- what a tuned persistent kernel would do for its activations
            const float v = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 0.5f + 1.0f;
            __hip_atomic_store(dst + w * 512 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's stores have left the CU
        } else {
            dst[w * 512 + tid] = __builtin_nontemporal_load(src + j) * 0.5f + 1.0f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");            // this workgroup's stores visible beyond its XCD's L2
        }
        __syncthreads();
        if (tid == 0) {
            const unsigned target = (unsigned)(it + 1) * (unsigned)nwg;
            if (mode == 2 || mode == 4) {
                unsigned* xc = ctr + 8 + 16 * xcc;
                const unsigned prev = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (prev + 1 == (unsigned)(it + 1) * per_xcd)
                    __hip_atomic_fetch_add(ctr, per_xcd, mode == 2 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_fetch_add(ctr, 1u, mode < 3 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned spins = 0;
            const bool dead = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;     // sticky: one timeout, not one per step
            while (!dead && (int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                if (mode == 1) __builtin_amdgcn_s_sleep(2);
                if (++spins > DBG_GBAR_BOUND) { atomicExch(err, 1 + it); break; }
            }
        }
        __syncthreads();
        if (mode < 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop stale lines before reading the peers' slices
    }
}
extern "C" int omni_debug_grid_barrier_chain(int mode, float* a, float* b, unsigned* counters, int* err, int blocks, int iters, void* stream) {
    OMNI_CHECK_ARG(blocks > 0 && blocks <= 256 && blocks % 8 == 0 && mode >= 0 && mode <= 4, "omni_debug_grid_barrier_chain: blocks=%d mode=%d", blocks, mode);
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(counters, 0, 8 * 64 + 64, st);
    hipMemsetAsync(err, 0, 4, st);
    hipLaunchKernelGGL(dbg_gbar_kernel, dim3(blocks), dim3(512), 0, st, a, b, counters, err, iters, mode);
    OMNI_CHECK_LAUNCH("omni_debug_grid_barrier_chain");
    return OMNI_OK;
}

// ---------------------------------------------------------------- operand-stream probe (scripts/probe_stream_mix.py)
// What bounds the backbone chain's streaming phases?  256 workgroups x 8 waves, one per CU.  Every wave keeps DEPTH 1-KB loads in
// flight (buffer_load_dwordx4, the chains' access) and walks
//   W: a private slice of a large buffer (unique bytes: HBM), and / or
//   X: one small buffer that EVERY workgroup reads again and again (L2 hits after the first touch; plain or sc1 loads).
// mode 0: W only.  1: X only.  2: every wave alternates W and X in ONE queue (the plain chain).  3: waves 0-3 stream W, waves 4-7
// stream X (two queues per SIMD pair).  Per-wave counts are given in 1-KB loads; the result word keeps the loads alive.
template <int DEPTH>
__global__ __launch_bounds__(512) void dbg_stream_mix_kernel(const uint8_t* __restrict__ W, size_t w_wg_bytes, const uint8_t* __restrict__ X,
                                                             unsigned x_bytes, int nw, int nx, int mode, int sc1, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)blockIdx.x * w_wg_bytes), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, 0x7fffffff, 0x00020000);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4 ring[DEPTH];
    unsigned acc = 0;
    const bool w_wave = mode == 0 || mode == 2 || (mode == 3 && wave < 4);
    const bool x_wave = mode == 1 || mode == 2 || (mode == 3 && wave >= 4);
    const int per = (mode == 3) ? 4 : 8;                  // waves sharing a stream
    const int wi = (mode == 3) ? (wave & 3) : wave;
    // total loads of this wave: W loads nw, X loads nx; in mode 2 they alternate 1 : (nx / nw) as the counts dictate
    const int total = (w_wave ? nw : 0) + (x_wave ? nx : 0);
    // loads are issued in order k = 0, 1, ...: running counters instead of divisions; mixed waves issue one W then R = nx / nw X loads
    const int R = (w_wave && x_wave) ? max(nx / max(nw, 1), 1) : 0;
    // sc1 bit 1 (value 2 added): stagger -- every workgroup starts its walk over X at a different place (no chip-wide same-line rush)
    const bool stagger = (sc1 & 2) != 0;
    sc1 &= 1;
    int wn = 0, xn = stagger ? (int)((blockIdx.x * 29u) & 0xffu) : 0, phase = 0;
    auto issue = [&](int) -> u4 {
        bool is_w = w_wave;
        if (w_wave && x_wave) {
            is_w = phase == 0;
            phase = phase == R ? 0 : phase + 1;
        }
        if (is_w) {
            const unsigned off = (unsigned)((((size_t)wn * per + wi) * 1024) & (w_wg_bytes - 1)) + lane * 16;     // sizes are powers of two
            ++wn;
            return __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0);
        }
        const unsigned off = (unsigned)((((size_t)xn * per + wi) * 1024) & (x_bytes - 1)) + lane * 16;
        ++xn;
        return sc1 ? __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 1);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring[d] = d < total ? issue(d) : (u4){0, 0, 0, 0};
    for (int k0 = 0; k0 < total; k0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int k = k0 + d;
            if (k < total) {
                acc ^= ring[d][0] ^ ring[d][3];
                if (k + DEPTH < total) ring[d] = issue(k + DEPTH);
            }
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

extern "C" int omni_debug_stream_mix(const void* W, size_t w_wg_bytes, const void* X, unsigned x_bytes, int nw, int nx, int mode, int sc1,
                                     int depth, unsigned* out, int reps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    for (int r = 0; r < reps; ++r) {
        if (depth == 4) hipLaunchKernelGGL(dbg_stream_mix_kernel<4>, dim3(256), dim3(512), 0, st, (const uint8_t*)W, w_wg_bytes, (const uint8_t*)X, x_bytes, nw, nx, mode, sc1, out);
        else if (depth == 8) hipLaunchKernelGGL(dbg_stream_mix_kernel<8>, dim3(256), dim3(512), 0, st, (const uint8_t*)W, w_wg_bytes, (const uint8_t*)X, x_bytes, nw, nx, mode, sc1, out);
        else if (depth == 16) hipLaunchKernelGGL(dbg_stream_mix_kernel<16>, dim3(256), dim3(512), 0, st, (const uint8_t*)W, w_wg_bytes, (const uint8_t*)X, x_bytes, nw, nx, mode, sc1, out);
        else hipLaunchKernelGGL(dbg_stream_mix_kernel<32>, dim3(256), dim3(512), 0, st, (const uint8_t*)W, w_wg_bytes, (const uint8_t*)X, x_bytes, nw, nx, mode, sc1, out);
    }
    OMNI_CHECK_LAUNCH("omni_debug_stream_mix");
    return OMNI_OK;
}
