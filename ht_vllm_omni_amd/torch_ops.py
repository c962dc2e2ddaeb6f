"""`torch.ops.mi355x_omni.*`: the per-op entry points of the C-ABI registered as PyTorch custom ops (SURVEY 8b last row,
north_star "Python host code calling HIP through PyTorch-ROCm custom ops").

Same C symbols as `ops.py` / `_lib.py` (include/omni_talker.h), reached through `torch.library`: tensors in, caller-allocated
outputs for the in-place / out-variant ops (graph-safe: no allocation, no host sync inside), torch's current HIP stream,
`RuntimeError` (OmniError) on a bad argument -- never an abort (V/worker error convention).  Only the CUDA (= HIP) dispatch key
has kernels: a CPU tensor fails loudly with PyTorch's "no kernel for backend" error, there is no fallback.

  rmsnorm_residual_(x?, delta?, residual?, w, out, eps)                 vLLM fused_add_rms_norm   (qwen3_tts_talker.py:341)
  skinny_gemm(x, w, bias?, mask?, epilogue, layout) -> Tensor          QKV / Row / MergedColumn / LMHead linear apply()
  silu_mul(gate_up) -> Tensor                                          SiluAndMul
  lmhead_mask(hidden, w, mask, round_bf16) -> Tensor                   compute_logits            (qwen3_tts_talker.py:424-443)
  qknorm_rope_kvwrite_(qkv, qnorm_w, knorm_w, positions, cos_sin, slots, q_out, k_cache, v_cache, ...)   q/k-norm + RoPE + cache write
  paged_attn_decode(q, k_cache, v_cache, block_table, seq_lens, ...) -> Tensor      vLLM paged attention, query_len 1
  paged_attn_prefill(q, k_cache, v_cache, block_table, req_of_tok, positions, ...) -> Tensor
  topk_sample(logits, seen?, steps?, greedy, temperature, top_k, top_p, rep_penalty, seed, ...) -> Tensor   vLLM Sampler
  allreduce_oneshot_(peers, r_io?, partials?, out?, which, M)          RowParallelLinear's all-reduce (tp_comm.PeerAllReduce)
  decode_step_(engine, out_record, positions, seq_lens, seen, steps, kv_caches, text_step, block_table, num_live, B, advance)
        the WHOLE talker decode step -- talker_mtp + code predictor, 28 layers over the paged KV, lm_head, sampler -- as ONE op over
        `omni_talker_decode_step` (gpu_ar_model_runner.py:93-400 `_model_forward` + :454 `_sample`): the call the product makes
        every step (engine.TalkerEngine.decode_step routes through it; hipGraph capture records its launches)
"""
from __future__ import annotations

import math

import torch

from . import _lib as L
from . import ops

NS = "mi355x_omni"
_lib = torch.library.Library(NS, "DEF")
_PEERS: dict[int, object] = {}          # allreduce_oneshot_: handle -> PeerAllReduce (a Python object cannot ride in an op schema)

_lib.define("rmsnorm_residual_(Tensor? x, Tensor? delta, Tensor(a!)? residual, Tensor w, Tensor(b!) out, float eps) -> ()")
_lib.define("skinny_gemm(Tensor x, Tensor w, Tensor? bias, Tensor? mask, int epilogue, int layout) -> Tensor")
_lib.define("silu_mul(Tensor gate_up) -> Tensor")
_lib.define("lmhead_mask(Tensor hidden, Tensor w, Tensor? mask, bool round_bf16) -> Tensor")
_lib.define("qknorm_rope_kvwrite_(Tensor qkv, Tensor qnorm_w, Tensor knorm_w, Tensor positions, Tensor cos_sin, Tensor slots, "
            "Tensor(a!) q_out, Tensor(b!) k_cache, Tensor(c!) v_cache, Tensor(d!)? k_scales, Tensor(e!)? v_scales, int q_heads, int kv_heads, "
            "int head_dim, float eps, int kv_dtype, float k_scale, float v_scale) -> ()")
_lib.define("paged_attn_decode(Tensor q, Tensor k_cache, Tensor v_cache, Tensor block_table, Tensor seq_lens, Tensor? k_scales, "
            "Tensor? v_scales, int q_heads, int kv_heads, int head_dim, int block_size, int kv_dtype, float k_scale, float v_scale, "
            "int max_seq_len) -> Tensor")
_lib.define("paged_attn_prefill(Tensor q, Tensor k_cache, Tensor v_cache, Tensor block_table, Tensor req_of_tok, Tensor positions, "
            "Tensor? k_scales, Tensor? v_scales, int q_heads, int kv_heads, int head_dim, int block_size, int kv_dtype, float k_scale, "
            "float v_scale) -> Tensor")
_lib.define("topk_sample(Tensor logits, Tensor(a!)? seen, Tensor(b!)? steps, bool greedy, float temperature, int top_k, float top_p, "
            "float rep_penalty, int seed, int step_mul, int step_add, bool inc_steps) -> Tensor")
_lib.define("allreduce_oneshot_(int peers, Tensor(a!)? r_io, Tensor(b!)? partials, Tensor(c!)? out, int which, int M, bool accumulate) -> ()")
_lib.define("decode_step_(int engine, Tensor(a!) out_record, Tensor(b!) positions, Tensor(c!) seq_lens, Tensor(d!) seen, Tensor(e!) steps, "
            "Tensor(f!)[] kv_caches, Tensor text_step, Tensor block_table, Tensor num_live, int B, bool advance) -> ()")
_ENGINES: dict[int, object] = {}        # decode_step_: handle -> TalkerEngine (weak: the engine unregisters itself)


def _impl(name):
    def deco(fn):
        _lib.impl(name, fn, "CUDA")
        return fn
    return deco


@_impl("rmsnorm_residual_")
def _rmsnorm_residual_(x, delta, residual, w, out, eps):
    src = residual if residual is not None else x
    ops._chk_dev(x, delta, residual, w, out)
    rows, hidden = src.shape
    L.check(L.load().omni_rmsnorm(L.ptr(x), L.ptr(delta), L.ptr(residual), L.ptr(w), L.ptr(out), rows, hidden, float(eps),
                                  L.current_stream()), "omni_rmsnorm")


@_impl("skinny_gemm")
def _skinny_gemm(x, w, bias, mask, epilogue, layout):
    return ops.gemm(x, w, bias=bias, epilogue=int(epilogue), mask=mask, layout=int(layout))


@_impl("silu_mul")
def _silu_mul(gate_up):
    return ops.silu_mul(gate_up)


@_impl("lmhead_mask")
def _lmhead_mask(hidden, w, mask, round_bf16):
    return ops.gemm(hidden, w, epilogue=L.EPI_F32_BF16RND if round_bf16 else L.EPI_F32, mask=mask)


@_impl("qknorm_rope_kvwrite_")
def _qknorm_rope_kvwrite_(qkv, qnorm_w, knorm_w, positions, cos_sin, slots, q_out, k_cache, v_cache, k_scales, v_scales, q_heads,
                          kv_heads, head_dim, eps, kv_dtype, k_scale, v_scale):
    ops._chk_dev(qkv, qnorm_w, knorm_w, positions, cos_sin, slots, q_out, k_cache, v_cache, k_scales, v_scales)
    L.check(L.load().omni_qknorm_rope_kvwrite(
        L.ptr(qkv), L.ptr(qnorm_w), L.ptr(knorm_w), L.ptr(positions), L.ptr(cos_sin), L.ptr(slots), L.ptr(q_out), L.ptr(k_cache),
        L.ptr(v_cache), L.ptr(k_scales), L.ptr(v_scales), qkv.shape[0], int(q_heads), int(kv_heads), int(head_dim), float(eps),
        int(kv_dtype), float(k_scale), float(v_scale), L.current_stream()), "omni_qknorm_rope_kvwrite")


@_impl("paged_attn_decode")
def _paged_attn_decode(q, k_cache, v_cache, block_table, seq_lens, k_scales, v_scales, q_heads, kv_heads, head_dim, block_size, kv_dtype,
                       k_scale, v_scale, max_seq_len):
    return ops.paged_attn_decode(q, k_cache, v_cache, block_table, seq_lens, q_heads=int(q_heads), kv_heads=int(kv_heads),
                                 head_dim=int(head_dim), block_size=int(block_size), kv_dtype=int(kv_dtype), k_scale=k_scale,
                                 v_scale=v_scale, k_scales=k_scales, v_scales=v_scales, max_seq_len=int(max_seq_len))


@_impl("paged_attn_prefill")
def _paged_attn_prefill(q, k_cache, v_cache, block_table, req_of_tok, positions, k_scales, v_scales, q_heads, kv_heads, head_dim,
                        block_size, kv_dtype, k_scale, v_scale):
    return ops.paged_attn_prefill(q, k_cache, v_cache, block_table, req_of_tok, positions, q_heads=int(q_heads), kv_heads=int(kv_heads),
                                  head_dim=int(head_dim), block_size=int(block_size), kv_dtype=int(kv_dtype), k_scale=k_scale,
                                  v_scale=v_scale, k_scales=k_scales, v_scales=v_scales)


@_impl("topk_sample")
def _topk_sample(logits, seen, steps, greedy, temperature, top_k, top_p, rep_penalty, seed, step_mul, step_add, inc_steps):
    return ops.sample(logits, greedy=bool(greedy), temperature=temperature, top_k=int(top_k), top_p=top_p, rep_penalty=rep_penalty,
                      seen=seen, seed=int(seed), steps=steps, step_mul=int(step_mul), step_add=int(step_add), inc_steps=bool(inc_steps))


@_impl("allreduce_oneshot_")
def _allreduce_oneshot_(peers, r_io, partials, out, which, M, accumulate):
    ar = _PEERS.get(int(peers))
    if ar is None:
        raise L.OmniError(f"allreduce_oneshot_: unknown peer-group handle {peers} (register_peers first)")
    ar.all_reduce(int(which), r_io=r_io, accumulate=bool(accumulate), partials=partials, out=out, M=int(M))


@_impl("decode_step_")
def _decode_step_(engine, out_record, positions, seq_lens, seen, steps, kv_caches, text_step, block_table, num_live, B, advance):
    eng = _ENGINES.get(int(engine))
    eng = eng() if eng is not None else None
    if eng is None:
        raise L.OmniError(f"decode_step_: unknown engine handle {engine} (register_engine first)")
    # the native step reads and writes the engine's graph-stable buffers: the tensors named in the schema (what the step mutates,
    # for the dispatcher's aliasing rules) must BE those buffers
    for name, t in (("out_record", out_record), ("positions", positions), ("seq_lens", seq_lens), ("seen", seen), ("steps", steps),
                    ("text_step", text_step), ("block_table", block_table), ("num_live", num_live)):
        if t.data_ptr() != getattr(eng, name).data_ptr():
            raise L.OmniError(f"decode_step_: `{name}` is not the engine's own buffer")
    if len(kv_caches) != len(eng.kv_caches) or any(a.data_ptr() != b.data_ptr() for a, b in zip(kv_caches, eng.kv_caches)):
        raise L.OmniError("decode_step_: `kv_caches` are not the engine's own cache tensors")
    eng._decode_step_native(int(B), bool(advance))


def register_engine(eng) -> int:
    """Handle of a TalkerEngine for `decode_step_` (weakly held: an engine that is gone makes the op raise)."""
    import weakref
    h = id(eng)
    _ENGINES[h] = weakref.ref(eng, lambda _r, h=h: _ENGINES.pop(h, None))
    return h


def register_peers(ar) -> int:
    """Handle of a connected tp_comm.PeerAllReduce for `allreduce_oneshot_` (the op schema carries an int, not the object)."""
    h = id(ar)
    _PEERS[h] = ar
    return h


OPS = ("rmsnorm_residual_", "skinny_gemm", "silu_mul", "lmhead_mask", "qknorm_rope_kvwrite_", "paged_attn_decode", "paged_attn_prefill",
       "topk_sample", "allreduce_oneshot_", "decode_step_")
