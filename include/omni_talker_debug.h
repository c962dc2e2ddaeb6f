/* Diagnostic hooks exported ONLY by libomni_talker_debug.so (same sources built with -DOMNI_DEBUG_HOOKS + csrc/debug.hip).  NOT part of the drop-in boundary:
 * they exist so that scripts/ (tile sweeps, same-box A/B runs, launch-cost probes) can flip a policy at run time.
 * Process-global, not thread-safe, defaults = the shipped policy. */
#pragma once
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
void omni_debug_set(int nt, int generic_schedule, int wgs); /* nt & 1: non-temporal W loads; != 0: predicated (pre-v8) GEMM
                                                               schedule; target workgroups per GEMM launch (256) */
void omni_debug_tile(int nt, int mt);                       /* force the GEMM tile (0 = policy)                       */
void omni_debug_int8_max_g(int g);                         /* int8-KV decode attention: q heads per workgroup (2 | 4) */
void omni_debug_cp_pair01(int on);                         /* code predictor: positions 0 and 1 as one two-block pass */
void omni_debug_cp_chain(int on);                          /* code predictor: the layer stack of a pass as one persistent launch */
void omni_debug_chain_mode(int dom, int gu_narrow, int nap); /* chain A/B: log2 flag domain (6 | 7 | 8); gate_up on the 32 x 24 tile; s_sleep units between polls */
void omni_debug_chain_stamps(void* buf);                   /* device uint64 [16 passes][40][8][256]: per-stage timeline stamps of every chain launch, one block per predictor pass, block 0 = the pair kernel (NULL: off) */
void omni_debug_bb_chain(int on);                          /* backbone: o_proj -> gate_up -> down_proj -> next qkv as one persistent launch per layer */
void omni_debug_bb_all(int on);                            /* backbone: the whole decoder stack, attention included, as ONE persistent launch (bb_all.hip) */
void omni_debug_bb_all_stamps(void* buf, int layer);       /* uint64 [8][8][256] timeline stamps of one layer's stages of that launch (NULL: off) */
void omni_debug_chain_defer(int on);                       /* code-predictor chain A/B arm: RMSNorm rstd applied in the qkv / gate_up epilogues (another rounding point than the reference's; timing + accuracy experiments) */
void omni_debug_chain_pair(int on);                        /* code predictor: positions 0 / 1 + group 1's head and sampler as one persistent launch (cp_pair_kernel) */
void omni_debug_chain_tail(int on);                         /* 1 (default): the step's input assembly + layer 0's qkv as the tail of the predictor's all-pass launch; 0: launches of their own (round 5) */
void omni_debug_chain_skip(int mode);                      /* code-predictor chain timing experiment: 1 = fetch half of every weight slice, 2 = half of the activations, 4 = the polling wave fetches no weights, 5 = no wave does (garbage results) */
void omni_debug_bb_min_rows(int rows);                     /* backbone chain: smallest batch the 64-row stage set takes (33 since round 5; 49: round 4's policy, 33-48 rows launch per op) */
void omni_debug_moe_chain(int on);                         /* sparse-MoE layer: o_proj -> router | shared gate_up -> shared down + routing as one persistent launch (moe_chain.hip) */
void omni_debug_pa_tail(int on);                           /* decode attention: the last, partial 128-token round as contiguous 32-token chunks per wave (0: interleaved groups) */
void omni_debug_pa_merge(int inkernel);                    /* 1: KV splits merged by their last arriver inside the attention launch (round-6 A/B arm; default 0 = the merge launch) */
void omni_debug_bb_deep(int mode);                         /* backbone chain arms: 1-3 deeper rings, 4 two-pass gate_up combine, 5 gate_up weights ahead of the flags, 6 rstd in the epilogue, 7-9 nt weight loads (gate_up / o+down+qkv / all) */
void omni_debug_bb_ar(int on);                             /* 0: tensor-parallel ranks keep the launch-per-op backbone with all-reduce launches (round 5) */
void omni_debug_sample_wave(int on);                       /* row sampler: 1 = one wave per row (smp_pick_wave, round 6) where eligible, 0 = always the 4-wave sample_kernel */
void omni_debug_bb_engine(int on);                         /* backbone segment as the loader / consumer engine (bb_engine.hip) instead of the plain chain */
void omni_debug_eng_stamps(void* buf);                     /* stage stamps of the engine's compute wave 0 */
void omni_debug_bb_stamps(void* buf);                      /* as omni_debug_chain_stamps for the backbone segment launches */
void omni_debug_gemm_defer(int on);                         /* launch path: 1 = rstd of the qkv / gate_up GEMMs applied in the epilogue (round 6 default, the chains' arithmetic), 0 = round 5's exact form */
void omni_debug_gemm_stage(int stage);                      /* leave every GEMM kernel after stage 1..4 (timing attribution only) */
void omni_debug_small_splitq(int on);                       /* small attention: one wave per (row, q head)            */
void omni_debug_small_tiny(int on);                         /* code-predictor attention: the (token, quarter) / readlane kernel */
void omni_debug_prefill_mfma(int on);                       /* prefill attention on MFMA (off: per-token VALU path)   */
void omni_debug_bb_head(int on);                            /* 1 (default): the talker's lm_head rides as the last stage of the last backbone launch; 0: its own launch (round 5) */
void omni_debug_extra_trivial(int n);                       /* append n no-op launches per layer phase                */
int omni_debug_launch(int mode, int blocks, int threads, void* p0, void* p1, int arg, int reps, void* stream);
int omni_debug_mix(int pattern, float* small, const void* big, size_t big_bytes, int reps, void* stream);
int omni_debug_cfgmix(int mode, float* p, int reps, void* stream);
int omni_debug_xcc_probe(int32_t* out, float* scratch, int gx, int gy, int odd, int reps, void* stream); /* XCC id of every block */
int omni_debug_chain(int mode, float* a, float* b, int blocks, int reps, void* stream);   /* 0: struct kernarg, 1: preloaded scalars */
/* One persistent launch of `blocks` (<= 256, multiple of 8) co-resident workgroups running `iters` steps of { cross-workgroup hand-off;
 * grid barrier } -- the in-kernel alternative to a kernel boundary (mode 0: one counter; 1: + s_sleep between polls; 2: per-XCD
 * counters feeding the global one).  counters: >= 576 zeroable bytes; err: int32 set when a spin ran out (bounded: no hang). */
int omni_debug_stream_mix(const void* W, size_t w_wg_bytes, const void* X, unsigned x_bytes, int nw, int nx, int mode, int sc1, int depth,
                          unsigned* out, int reps, void* stream);   /* operand-stream probe (scripts/probe_stream_mix.py) */
int omni_debug_grid_barrier_chain(int mode, float* a, float* b, unsigned* counters, int* err, int blocks, int iters, void* stream);

#ifdef __cplusplus
}
#endif
