"""Model dimensions of the talker stage (parameters, never hard-coded in kernels).

The reference reads them from the HF checkpoint (``Qwen3TTSTalkerConfig`` /
``Qwen3TTSTalkerCodePredictorConfig``,
R/vllm_omni/model_executor/models/qwen3_tts/configuration_qwen3_tts.py:192-216,376-409);
the presets below are the shapes SURVEY.md section 8 names for the BASELINE.json configs.
"""
from __future__ import annotations

from dataclasses import dataclass, replace


@dataclass(frozen=True)
class TalkerDims:
    name: str
    # backbone (vLLM Qwen3Model in the reference, qwen3_tts_talker.py:341)
    hidden: int
    layers: int
    q_heads: int
    kv_heads: int
    head_dim: int
    inter: int
    vocab: int                 # talker vocab (codec_head rows)
    codebook: int              # real code ids are [0, codebook) (qwen3_tts_talker.py:321)
    eos_id: int                # codec_eos_token_id
    codec_pad_id: int
    num_code_groups: int       # Q
    rope_theta: float
    eps: float
    # code predictor (qwen3_tts_code_predictor_vllm.py:234-343)
    cp_hidden: int
    cp_layers: int
    cp_q_heads: int
    cp_kv_heads: int
    cp_head_dim: int
    cp_inter: int
    cp_rope_theta: float
    max_model_len: int = 4096  # qwen3_tts.yaml:22
    # sparse-MoE backbone MLP (Qwen3-Omni talker: HF Qwen3OmniMoeTalkerTextConfig); 0 experts = dense MLP
    moe_experts: int = 0
    moe_top_k: int = 0
    moe_inter: int = 0
    moe_shared_inter: int = 0
    moe_norm_topk: bool = False
    # M-RoPE (HF rope_scaling {"mrope_section", "interleaved"}): None = plain RoPE.  The talker's prompt ids normally have three
    # identical rows (positions.py); differing rows and a non-zero mrope_position_delta are honoured when a request carries them
    mrope_section: tuple | None = None
    mrope_interleaved: bool = False

    @property
    def qkv_out(self) -> int:
        return (self.q_heads + 2 * self.kv_heads) * self.head_dim

    @property
    def cp_qkv_out(self) -> int:
        return (self.cp_q_heads + 2 * self.cp_kv_heads) * self.cp_head_dim

    @property
    def has_cp_projection(self) -> bool:
        # code_predictor_vllm.py:340-343
        return self.cp_hidden != self.hidden

    def with_(self, **kw) -> "TalkerDims":
        return replace(self, **kw)


def _tts(name: str, hidden: int, inter: int) -> TalkerDims:
    return TalkerDims(
        name=name, hidden=hidden, layers=28, q_heads=16, kv_heads=8, head_dim=128, inter=inter,
        vocab=3072, codebook=2048, eos_id=2150, codec_pad_id=2148, num_code_groups=16,
        rope_theta=1_000_000.0, eps=1e-6,
        cp_hidden=1024, cp_layers=5, cp_q_heads=16, cp_kv_heads=8, cp_head_dim=128, cp_inter=3072,
        cp_rope_theta=10_000.0,
    )


PRESETS: dict[str, TalkerDims] = {
    # small shapes for parity tests (head_dim stays 128: the only value the talkers use)
    "tiny": TalkerDims(
        name="tiny", hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512,
        vocab=192, codebook=128, eos_id=150, codec_pad_id=148, num_code_groups=4,
        rope_theta=1_000_000.0, eps=1e-6,
        cp_hidden=128, cp_layers=2, cp_q_heads=2, cp_kv_heads=1, cp_head_dim=128, cp_inter=256,
        cp_rope_theta=10_000.0, max_model_len=512,
    ),
    # MoE talker in miniature (Omni: no code-predictor projection, top-k experts + gated shared expert)
    "omni-moe-tiny": TalkerDims(
        name="omni-moe-tiny", hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=128,
        vocab=192, codebook=128, eos_id=150, codec_pad_id=148, num_code_groups=4,
        rope_theta=1_000_000.0, eps=1e-6,
        cp_hidden=256, cp_layers=2, cp_q_heads=2, cp_kv_heads=1, cp_head_dim=128, cp_inter=256,
        cp_rope_theta=10_000.0, max_model_len=512,
        moe_experts=16, moe_top_k=4, moe_inter=64, moe_shared_inter=64,
        mrope_section=(24, 20, 20), mrope_interleaved=True,
    ),
    # Qwen3-Omni talker (BASELINE configs #4 / #5): HF Qwen3OmniMoeTalkerTextConfig defaults (hidden 1024, 20 layers, 16 q /
    # 2 kv heads, 128 experts top-8 of width 384) + shared expert 768 and head_dim 128 as in the released checkpoint
    # (not stated in R/: parameters); code predictor as the TTS one but without projection (width == talker width)
    "omni-talker": TalkerDims(
        name="omni-talker", hidden=1024, layers=20, q_heads=16, kv_heads=2, head_dim=128, inter=768,
        vocab=3072, codebook=2048, eos_id=2150, codec_pad_id=2148, num_code_groups=16,
        rope_theta=1_000_000.0, eps=1e-6,
        cp_hidden=1024, cp_layers=5, cp_q_heads=16, cp_kv_heads=8, cp_head_dim=128, cp_inter=3072,
        cp_rope_theta=10_000.0,
        moe_experts=128, moe_top_k=8, moe_inter=384, moe_shared_inter=768,
        mrope_section=(24, 20, 20), mrope_interleaved=True,
    ),
    "tts-0.6b": _tts("tts-0.6b", 1024, 3072),   # BASELINE config #2
    "tts-1.7b": _tts("tts-1.7b", 2048, 6144),   # BASELINE config #3 (headline)
}


def get_dims(name: str) -> TalkerDims:
    return PRESETS[name]
