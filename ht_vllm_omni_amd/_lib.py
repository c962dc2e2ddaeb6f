"""ctypes binding of libomni_talker.so (the C-ABI of include/omni_talker.h).

The product path has NO fallback: if the HIP library is missing or an entry point fails,
this module raises.  `import torch` happens first so the library resolves libamdhip64.so.7
to the HIP runtime PyTorch-ROCm already loaded (one runtime, shared streams).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the dlopen below: shares torch's HIP runtime)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libomni_talker.so")

KV_BF16, KV_FP8, KV_INT8, KV_FP16 = 0, 1, 2, 3
KV_CODES = {"bf16": KV_BF16, "auto": KV_BF16, "fp8": KV_FP8, "fp8_e4m3": KV_FP8, "int8": KV_INT8, "fp16": KV_FP16, "float16": KV_FP16,
            "half": KV_FP16}
EPI_BF16, EPI_SILU_MUL, EPI_F32, EPI_F32_BF16RND = 0, 1, 2, 3
EPI_RESID, EPI_SILU_MUL_GU8 = 4, 5
SILU_EPIS = (EPI_SILU_MUL, EPI_SILU_MUL_GU8)
LAYOUT_W_FRAG, LAYOUT_X_FRAG, LAYOUT_OUT_FRAG = 1, 2, 4

vp, i32, i64, f32, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32


class OmniError(RuntimeError):
    pass


class LayerWeights(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1", "wqkv", "qnorm", "knorm", "wo", "ln2", "wgu", "wdown", "moe_router", "moe_gate_up",
                                  "moe_down", "moe_shared_gate_up", "moe_shared_down", "moe_shared_gate", "moe_gate_up_scale",
                                  "moe_down_scale")]


class ArPeers(C.Structure):
    """omni_ar_peers: peer-mapped partial buffers + flag words of a tensor-parallel group."""
    _fields_ = [("world", i32), ("rank", i32), ("data", vp * 8), ("flags", vp * 8), ("epoch", vp), ("error", vp), ("tile_flags", vp * 8)]


class TileGemm(C.Structure):
    """omni_tile_gemm: one large-M MFMA GEMM / conv-as-GEMM launch (include/omni_talker.h)."""
    _fields_ = [("x", vp), ("x_rows", i64), ("ldx", i32), ("seg_len", i32), ("seg_rows", i32), ("row_off", i32),
                ("w", vp), ("bias", vp), ("scale", vp), ("act", i32), ("resid", vp), ("ldr", i32), ("out_f32", vp), ("ldf", i32), ("out", vp), ("ldo", i32),
                ("out2", vp), ("ldo2", i32), ("snake_alpha", vp), ("snake_inv_beta", vp), ("M", i32), ("N", i32), ("K", i32), ("tile_hint", i32),
                ("groups", i32), ("x_group_rows", i64), ("w_group_elems", i64), ("out_group_rows", i64), ("group_rows", vp)]


TILE_ACT_NONE, TILE_ACT_GELU, TILE_ACT_SILU_MUL_GU8 = 0, 1, 2


class ResUnit(C.Structure):
    """omni_res_unit (include/omni_codec.h): one fused residual unit of the Code2Wav decoder."""
    _fields_ = [("s", vp), ("h", vp), ("s_next", vp), ("w1", vp), ("b1", vp), ("snake2_alpha", vp), ("snake2_inv_beta", vp),
                ("w2", vp), ("b2", vp), ("next_alpha", vp), ("next_inv_beta", vp), ("T", i32), ("C", i32), ("dilation", i32)]


class TalkerDesc(C.Structure):
    _fields_ = [
        ("hidden", i32), ("layers", i32), ("q_heads", i32), ("kv_heads", i32), ("head_dim", i32), ("inter", i32),
        ("vocab", i32), ("codebook", i32), ("num_code_groups", i32), ("eps", f32),
        ("cp_hidden", i32), ("cp_layers", i32), ("cp_q_heads", i32), ("cp_kv_heads", i32), ("cp_head_dim", i32),
        ("cp_inter", i32), ("has_cp_projection", i32), ("frag_layout", i32),
        ("moe_experts", i32), ("moe_top_k", i32), ("moe_inter", i32), ("moe_shared_inter", i32), ("moe_norm_topk", i32),
        ("fused_norm", i32), ("cp_fused_norm", i32),
        ("max_batch", i32), ("block_size", i32), ("kv_dtype", i32), ("max_model_len", i32), ("bt_stride", i32),
        ("k_scale", f32), ("v_scale", f32), ("masked_logit", f32),
        ("embed", vp), ("layer", C.POINTER(LayerWeights)), ("final_norm", vp), ("lm_head", vp), ("allowed_mask", vp),
        ("cos_sin", vp), ("cp_proj_w", vp), ("cp_proj_b", vp), ("cp_layer", C.POINTER(LayerWeights)), ("cp_norm", vp),
        ("cp_lm_head", vp), ("cp_embed", vp), ("cp_cos_sin", vp), ("cp_proj_table", vp), ("cp_e0_table", vp),
        ("k_cache", C.POINTER(vp)), ("v_cache", C.POINTER(vp)), ("k_scales", C.POINTER(vp)), ("v_scales", C.POINTER(vp)),
        ("scratch", vp), ("scratch_bytes", i64),
        ("ar_attn", C.POINTER(ArPeers)), ("ar_mlp", C.POINTER(ArPeers)),
        ("moe_e0", i32), ("moe_experts_local", i32), ("moe_w8", i32),
        ("cp_chain", i32), ("rope_rows", i32),
    ]


class RowSampling(C.Structure):
    """omni_row_sampling: per-row device arrays [B] (NULL member = the scalar of StepIO)."""
    _fields_ = [("greedy", vp), ("temperature", vp), ("top_k", vp), ("top_p", vp), ("rep_penalty", vp), ("seed", vp)]


class StepIO(C.Structure):
    _fields_ = [
        ("B", i32), ("input_ids", vp), ("positions", vp), ("seq_lens", vp), ("block_table", vp), ("slot_mapping", vp),
        ("last_hidden", vp), ("text_step", vp), ("inputs_embeds", vp), ("audio_codes", vp), ("logits", vp),
        ("seen", vp), ("steps", vp),
        ("greedy", i32), ("temperature", f32), ("top_k", i32), ("rep_penalty", f32), ("seed", u32),
        ("cp_greedy", i32), ("cp_temperature", f32), ("cp_top_k", i32),
        ("advance", i32), ("top_p", f32), ("cp_top_p", f32),
        ("num_live", vp), ("rows", RowSampling), ("rope_delta", vp), ("status", vp),
    ]


# every symbol include/omni_talker.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "omni_last_error": (C.c_char_p, []),
    "omni_abi_version": (i32, []),
    "omni_rmsnorm": (i32, [vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    "omni_gemm_bf16": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "omni_gemm_bf16_ex": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]),
    "omni_silu_mul": (i32, [vp, vp, i32, i32, vp]),
    "omni_silu": (i32, [vp, vp, C.c_longlong, vp]),
    "omni_resize_mlp": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "omni_moe_route": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "omni_moe_experts": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "omni_moe_experts_ex": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "omni_moe_experts_resid": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "omni_snake_beta": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "omni_gemm_tile": (i32, [C.POINTER(TileGemm), vp]),
    # include/omni_codec.h
    "omni_codec_rvq_embed": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "omni_codec_rmsnorm": (i32, [vp, i32, vp, f32, vp, i32, i32, vp]),
    "omni_codec_rope": (i32, [vp, i32, i32, i32, i32, i32, f32, vp]),
    "omni_codec_window_attn": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
    "omni_codec_dwconv_ln": (i32, [vp, i32, vp, vp, vp, vp, f32, vp, i32, i32, i32, vp]),
    "omni_codec_out_conv": (i32, [vp, vp, f32, vp, i32, i32, i32, vp]),
    "omni_codec_res_unit_supported": (i32, [i32, i32, i32]),
    "omni_codec_res_unit": (i32, [C.POINTER(ResUnit), vp]),
    "omni_gemm_resid": (i32, [vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "omni_gemm_xnorm": (i32, [vp, vp, i32, vp, f32, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]),
    "omni_attn_decode_fused": (i32, [vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32,
                                     f32, f32, f32, i32, vp]),
    "omni_qknorm_rope_kvwrite": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, f32, f32, vp]),
    "omni_qknorm_mrope_kvwrite": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, f32, f32, vp]),
    "omni_slot_mapping": (i32, [vp, i32, vp, vp, i32, i32, i32, vp]),
    "omni_paged_attn_decode": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, f32, f32, i32, vp]),
    "omni_paged_attn_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "omni_paged_attn_prefill": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, f32, f32, vp]),
    "omni_embed": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "omni_sample": (i32, [vp, i32, i32, i32, i32, f32, i32, f32, f32, vp, u32, vp, i32, i32, i32, vp, vp]),
    "omni_sample_rows": (i32, [vp, i32, i32, i32, C.POINTER(RowSampling), vp, vp, i32, i32, i32, vp, vp]),
    "omni_allreduce_resid": (i32, [C.POINTER(ArPeers), vp, i32, vp, i32, vp, i32, i32, vp]),
    "omni_ar_alloc": (i32, [i64, C.POINTER(vp), vp]),
    "omni_ar_open": (i32, [vp, C.POINTER(vp)]),
    "omni_ar_close": (i32, [vp]),
    "omni_ar_free": (i32, [vp]),
    "omni_talker_scratch_bytes": (i64, [C.POINTER(TalkerDesc)]),
    "omni_talker_create": (vp, [C.POINTER(TalkerDesc)]),
    "omni_talker_destroy": (None, [vp]),
    "omni_talker_chain_error": (i32, [vp, i32]),
    "omni_talker_set_chains": (i32, [vp, i32]),
    "omni_talker_chains_ran": (i32, [vp]),
    "omni_talker_set_kv_scales": (i32, [vp, vp, vp, vp]),
    "omni_talker_mtp": (i32, [vp, C.POINTER(StepIO), vp]),
    "omni_talker_layer_attn": (i32, [vp, C.POINTER(StepIO), i32, vp]),
    "omni_talker_layer_mlp": (i32, [vp, C.POINTER(StepIO), i32, vp]),
    "omni_talker_finish": (i32, [vp, C.POINTER(StepIO), vp]),
    "omni_talker_decode_step": (i32, [vp, C.POINTER(StepIO), vp]),
    "omni_talker_backbone_step": (i32, [vp, C.POINTER(StepIO), vp]),
    "omni_talker_step_part": (i32, [vp, C.POINTER(StepIO), i32, vp]),
    "omni_talker_attn_out": (vp, [vp]),
    "omni_talker_mlp_out": (vp, [vp]),
    "omni_talker_prefill": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "omni_talker_rows_begin": (i32, [vp, vp, i32, vp]),
    "omni_talker_rows_attn": (i32, [vp, i32, i32, vp, vp, vp, vp, vp]),
    "omni_talker_rows_mlp": (i32, [vp, i32, i32, vp]),
    "omni_talker_rows_end": (i32, [vp, vp, i32, vp]),
    "omni_talker_logits": (i32, [vp, vp, vp, i32, i32, vp]),
    "omni_talker_code_predictor": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, f32, u32, vp, vp]),
}

_lib = None
_product = None
DEBUG_LIB_PATH = os.path.join(HERE, "libomni_talker_debug.so")


def _open(path: str, debug: bool) -> C.CDLL:
    if not os.path.exists(path):
        raise OmniError(
            f"{path} is missing: build it with `python -m ht_vllm_omni_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    # a library older than csrc/ would silently run yesterday's kernels: compare the digest stamped at link time with the
    # sources that are here (skipped when the sources did not travel, e.g. an installed wheel)
    stamp, csrc = path + ".digest", os.path.join(HERE, "csrc")
    if os.path.isdir(csrc) and os.environ.get("OMNI_SKIP_STALE_CHECK") != "1":
        from .build import sources_digest
        have = open(stamp).read().strip() if os.path.exists(stamp) else "<no stamp>"
        if have != sources_digest(debug):
            raise OmniError(f"{path} is stale: built from other sources than {csrc} (digest {have[:12]}); "
                            "rebuild with `python -m ht_vllm_omni_amd.build`")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise OmniError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


def load() -> C.CDLL:
    """dlopen the product library and bind every declared symbol; raises OmniError when absent or stale.
    OMNI_TALKER_DEBUG=1 (scripts/ only) selects libomni_talker_debug.so, the same ABI plus the omni_debug_* hooks."""
    global _lib
    if _lib is None:
        dbg = os.environ.get("OMNI_TALKER_DEBUG") == "1"
        _lib = _open(DEBUG_LIB_PATH if dbg else LIB_PATH, dbg)
    return _lib


class debug_library:
    """Context manager for the tile-sweep / A-B tests: engines and ops created inside run on libomni_talker_debug.so
    (same sources, run-time policy knobs); the product library is restored on exit."""

    def __enter__(self) -> C.CDLL:
        global _lib, _product
        _product = _lib
        _lib = _open(DEBUG_LIB_PATH, True)
        return _lib

    def __exit__(self, *exc) -> None:
        global _lib
        _lib = _product


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().omni_last_error().decode(errors="replace")
        raise OmniError(f"{what or 'omni call'} failed (rc={rc}): {msg}")


def ptr(t) -> int | None:
    """Device/host pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    return torch.cuda.current_stream().cuda_stream
