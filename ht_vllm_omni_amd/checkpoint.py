"""Checkpoint loading for the talker stage: HF Qwen3-TTS safetensors -> the fused layout the engine takes.

The reference maps HF names with ``WeightsMapper`` prefixes (qwen3_tts_talker.py:297-311: ``talker.model.layers.`` ->
``model.layers.``, ``talker.model.codec_embedding.`` -> ``model.embed_tokens.``, ``talker.codec_head.`` -> ``lm_head.``,
``talker.code_predictor.`` -> ``code_predictor.``) and lets vLLM's ``Qwen3Model.load_weights`` stack ``q_proj / k_proj /
v_proj`` into ``qkv_proj`` and ``gate_proj / up_proj`` into ``gate_up_proj`` (load_weights: 1569-1590; the code
predictor keeps plain per-projection ``nn.Linear`` modules: qwen3_tts_code_predictor_vllm.py:110-193,275-287,363-384).
This module does the same renaming + stacking straight from the safetensors shards into the key set of
``weights.make_weights`` (``l{i}.wqkv`` = [q rows | k rows | v rows], ``l{i}.wgu`` = [gate rows | up rows], ...), so
``MI355XARWorker.load_model`` can serve a real checkpoint directory:

    weights, extras = load_talker_checkpoint("/models/Qwen3-TTS-12Hz-1.7B-Base")
    dims = dims_from_hf_config("/models/Qwen3-TTS-12Hz-1.7B-Base/config.json")

``extras`` holds the tensors of the stage that are not on the decode path (text embedding, text projection MLP, speaker
encoder): the prompt builder's inputs.  ``export_hf_checkpoint`` writes the inverse mapping (tests: a synthetic
checkpoint round-trips bit-exactly; no real checkpoint exists in the build environment).
"""
from __future__ import annotations

import json
import os
import re
from typing import Iterable, Iterator

import torch

from .config import TalkerDims

BF16 = torch.bfloat16

# HF leaf name -> (our key, slot inside a fused tensor or None)
_ATTN = {"q_proj": ("wqkv", 0), "k_proj": ("wqkv", 1), "v_proj": ("wqkv", 2), "o_proj": ("wo", None)}
_MLP = {"gate_proj": ("wgu", 0), "up_proj": ("wgu", 1), "down_proj": ("wdown", None)}
_LAYER_RE = re.compile(r"^(talker\.model|talker\.code_predictor\.model)\.layers\.(\d+)\.(.+)$")


def iter_safetensors(path: str) -> Iterator[tuple[str, torch.Tensor]]:
    """(name, tensor) over a .safetensors file, a sharded directory (model.safetensors.index.json) or a directory of shards."""
    from safetensors import safe_open
    files: list[str]
    if os.path.isdir(path):
        idx = os.path.join(path, "model.safetensors.index.json")
        if os.path.exists(idx):
            files = sorted({os.path.join(path, f) for f in json.load(open(idx))["weight_map"].values()})
        else:
            files = sorted(os.path.join(path, f) for f in os.listdir(path) if f.endswith(".safetensors"))
    else:
        files = [path]
    if not files:
        raise FileNotFoundError(f"no .safetensors under {path}")
    for f in files:
        with safe_open(f, framework="pt", device="cpu") as sf:
            for k in sf.keys():
                yield k, sf.get_tensor(k)


def map_hf_talker_weights(named: Iterable[tuple[str, torch.Tensor]], dtype: torch.dtype = BF16) -> tuple[dict, dict]:
    """HF names -> (engine weight dict, extras).  Unknown ``talker.*`` names raise: a silently dropped tensor is how a
    checkpoint ends up half loaded."""
    w: dict[str, torch.Tensor] = {}
    extras: dict[str, torch.Tensor] = {}
    parts: dict[str, dict[int, torch.Tensor]] = {}          # fused tensors being assembled
    unmapped: list[str] = []
    cp_embed: dict[int, torch.Tensor] = {}
    cp_head: dict[int, torch.Tensor] = {}
    for name, t in named:
        if "rotary_emb.inv_freq" in name:                    # code_predictor_vllm.py:279-280
            continue
        if not name.startswith("talker."):
            extras[name] = t                                  # speaker_encoder.*, anything outside the talker
            continue
        t = t.to(dtype)
        m = _LAYER_RE.match(name)
        if m:
            pre = ("" if m.group(1) == "talker.model" else "cp.") + f"l{int(m.group(2))}."
            leaf = m.group(3)
            if leaf == "input_layernorm.weight":
                w[pre + "ln1"] = t
            elif leaf == "post_attention_layernorm.weight":
                w[pre + "ln2"] = t
            elif leaf == "self_attn.q_norm.weight":
                w[pre + "qnorm"] = t
            elif leaf == "self_attn.k_norm.weight":
                w[pre + "knorm"] = t
            else:
                mm = re.match(r"^(self_attn|mlp)\.(\w+)\.weight$", leaf)
                table = _ATTN if mm and mm.group(1) == "self_attn" else _MLP
                if not mm or mm.group(2) not in table:
                    unmapped.append(name)
                    continue
                key, slot = table[mm.group(2)]
                if slot is None:
                    w[pre + key] = t
                else:
                    parts.setdefault(pre + key, {})[slot] = t
            continue
        if name == "talker.model.codec_embedding.weight":
            w["embed"] = t
        elif name == "talker.model.norm.weight":
            w["norm"] = t
        elif name == "talker.codec_head.weight":
            w["lm_head"] = t
        elif name == "talker.code_predictor.model.norm.weight":
            w["cp.norm"] = t
        elif name == "talker.code_predictor.small_to_mtp_projection.weight":
            w["cp.proj_w"] = t
        elif name == "talker.code_predictor.small_to_mtp_projection.bias":
            w["cp.proj_b"] = t
        elif (m2 := re.match(r"^talker\.code_predictor\.model\.codec_embedding\.(\d+)\.weight$", name)):
            cp_embed[int(m2.group(1))] = t
        elif (m2 := re.match(r"^talker\.code_predictor\.lm_head\.(\d+)\.weight$", name)):
            cp_head[int(m2.group(1))] = t
        elif name.startswith(("talker.model.text_embedding.", "talker.text_projection.")):
            extras[name[len("talker."):].replace("model.text_embedding", "text_embedding")] = t      # talker.py:303-304
        else:
            unmapped.append(name)
    if unmapped:       # every name at once: a checkpoint of another variant (biases, extra heads) shows its whole difference
        raise KeyError(f"{len(unmapped)} unmapped talker weight(s): " + ", ".join(sorted(unmapped)))
    for key, sl in parts.items():
        need = 3 if key.endswith("wqkv") else 2
        if sorted(sl) != list(range(need)):
            raise KeyError(f"{key}: projections {sorted(sl)} present, {need} needed")
        w[key] = torch.cat([sl[i] for i in range(need)], 0).contiguous()
    for nm, d in (("cp.embed", cp_embed), ("cp.lm_head", cp_head)):
        if d:
            if sorted(d) != list(range(len(d))):
                raise KeyError(f"{nm}: groups {sorted(d)} are not contiguous")
            w[nm] = torch.stack([d[i] for i in range(len(d))]).contiguous()
    return w, extras


def load_talker_checkpoint(path: str, dtype: torch.dtype = BF16) -> tuple[dict, dict]:
    return map_hf_talker_weights(iter_safetensors(path), dtype)


def check_against_dims(w: dict, d: TalkerDims) -> None:
    """Shapes of a loaded checkpoint vs the dimensions the engine will be built with (raises ValueError)."""
    want = {"embed": (d.vocab, d.hidden), "norm": (d.hidden,), "lm_head": (d.vocab, d.hidden), "cp.norm": (d.cp_hidden,),
            "cp.lm_head": (d.num_code_groups - 1, d.codebook, d.cp_hidden), "cp.embed": (d.num_code_groups - 1, d.codebook, d.hidden)}
    if d.has_cp_projection:
        want.update({"cp.proj_w": (d.cp_hidden, d.hidden), "cp.proj_b": (d.cp_hidden,)})
    for pre, n, H, qo, od, I, D in (("", d.layers, d.hidden, d.qkv_out, d.q_heads * d.head_dim, d.inter, d.head_dim),
                                    ("cp.", d.cp_layers, d.cp_hidden, d.cp_qkv_out, d.cp_q_heads * d.cp_head_dim, d.cp_inter, d.cp_head_dim)):
        for i in range(n):
            p = f"{pre}l{i}."
            want.update({p + "ln1": (H,), p + "ln2": (H,), p + "qnorm": (D,), p + "knorm": (D,), p + "wqkv": (qo, H),
                         p + "wo": (H, od), p + "wgu": (2 * I, H), p + "wdown": (H, I)})
    for k, shp in want.items():
        if k not in w:
            raise ValueError(f"checkpoint lacks {k}")
        if tuple(w[k].shape) != shp:
            raise ValueError(f"{k}: checkpoint shape {tuple(w[k].shape)} != {shp} from the config")


def dims_from_hf_config(cfg: str | dict, name: str = "checkpoint", max_model_len: int = 4096) -> TalkerDims:
    """TalkerDims from a Qwen3-TTS config.json (``talker_config`` + its ``code_predictor_config``;
    configuration_qwen3_tts.py:192-216,376-409 hold the defaults the reference falls back to)."""
    if isinstance(cfg, str):
        cfg = json.load(open(cfg))
    t = cfg.get("talker_config", cfg)
    c = t.get("code_predictor_config") or {}
    heads = int(t.get("num_attention_heads", 16))
    # the talker config class has no head_dim field (configuration_qwen3_tts.py:376-409): the HF / vLLM Qwen3 convention is
    # hidden_size // num_attention_heads when the key is absent; the code-predictor config class defaults it to 128 (:200)
    hd = int(t.get("head_dim") or int(t.get("hidden_size", 1024)) // heads)
    return TalkerDims(
        name=name, hidden=int(t.get("hidden_size", 1024)), layers=int(t.get("num_hidden_layers", 20)), q_heads=heads,
        kv_heads=int(t.get("num_key_value_heads", 2)), head_dim=hd, inter=int(t.get("intermediate_size", 2048)),
        vocab=int(t.get("vocab_size", 3072)), codebook=int(c.get("vocab_size", 2048)), eos_id=int(t.get("codec_eos_token_id", 4198)),
        codec_pad_id=int(t.get("codec_pad_id", 4196)), num_code_groups=int(t.get("num_code_groups", 32)),
        rope_theta=float(t.get("rope_theta", 10000)), eps=float(t.get("rms_norm_eps", 1e-6)),
        cp_hidden=int(c.get("hidden_size", 1024)), cp_layers=int(c.get("num_hidden_layers", 5)),
        cp_q_heads=int(c.get("num_attention_heads", 16)), cp_kv_heads=int(c.get("num_key_value_heads", 8)),
        cp_head_dim=int(c.get("head_dim", 128)), cp_inter=int(c.get("intermediate_size", 3072)),
        cp_rope_theta=float(c.get("rope_theta", 10000)), max_model_len=max_model_len)


def hf_config_from_dims(d: TalkerDims) -> dict:
    return {"talker_config": {
        "hidden_size": d.hidden, "num_hidden_layers": d.layers, "num_attention_heads": d.q_heads, "num_key_value_heads": d.kv_heads,
        "head_dim": d.head_dim, "intermediate_size": d.inter, "vocab_size": d.vocab, "codec_eos_token_id": d.eos_id,
        "codec_pad_id": d.codec_pad_id, "num_code_groups": d.num_code_groups, "rope_theta": d.rope_theta, "rms_norm_eps": d.eps,
        "code_predictor_config": {"vocab_size": d.codebook, "hidden_size": d.cp_hidden, "num_hidden_layers": d.cp_layers,
                                  "num_attention_heads": d.cp_q_heads, "num_key_value_heads": d.cp_kv_heads, "head_dim": d.cp_head_dim,
                                  "intermediate_size": d.cp_inter, "rope_theta": d.cp_rope_theta, "num_code_groups": d.num_code_groups}}}


def export_hf_checkpoint(w: dict, d: TalkerDims, out_dir: str, shards: int = 2) -> None:
    """The inverse mapping: engine weight dict -> HF-named safetensors shards + index + config.json."""
    from safetensors.torch import save_file
    hf: dict[str, torch.Tensor] = {"talker.model.codec_embedding.weight": w["embed"], "talker.model.norm.weight": w["norm"],
                                   "talker.codec_head.weight": w["lm_head"], "talker.code_predictor.model.norm.weight": w["cp.norm"]}
    if "cp.proj_w" in w:
        hf["talker.code_predictor.small_to_mtp_projection.weight"] = w["cp.proj_w"]
        hf["talker.code_predictor.small_to_mtp_projection.bias"] = w["cp.proj_b"]
    for g in range(w["cp.embed"].shape[0]):
        hf[f"talker.code_predictor.model.codec_embedding.{g}.weight"] = w["cp.embed"][g]
        hf[f"talker.code_predictor.lm_head.{g}.weight"] = w["cp.lm_head"][g]
    for pre, root, n, qh, kvh, D, I in (("", "talker.model", d.layers, d.q_heads, d.kv_heads, d.head_dim, d.inter),
                                        ("cp.", "talker.code_predictor.model", d.cp_layers, d.cp_q_heads, d.cp_kv_heads, d.cp_head_dim, d.cp_inter)):
        for i in range(n):
            p, r = f"{pre}l{i}.", f"{root}.layers.{i}."
            q, k, v = torch.split(w[p + "wqkv"], [qh * D, kvh * D, kvh * D], 0)
            gate, up = torch.split(w[p + "wgu"], [I, I], 0)
            hf.update({r + "input_layernorm.weight": w[p + "ln1"], r + "post_attention_layernorm.weight": w[p + "ln2"],
                       r + "self_attn.q_norm.weight": w[p + "qnorm"], r + "self_attn.k_norm.weight": w[p + "knorm"],
                       r + "self_attn.q_proj.weight": q, r + "self_attn.k_proj.weight": k, r + "self_attn.v_proj.weight": v,
                       r + "self_attn.o_proj.weight": w[p + "wo"], r + "mlp.gate_proj.weight": gate, r + "mlp.up_proj.weight": up,
                       r + "mlp.down_proj.weight": w[p + "wdown"]})
    os.makedirs(out_dir, exist_ok=True)
    names = sorted(hf)
    index = {"metadata": {}, "weight_map": {}}
    for s in range(shards):
        fn = f"model-{s + 1:05d}-of-{shards:05d}.safetensors"
        part = {k: hf[k].contiguous() for k in names[s::shards]}
        save_file(part, os.path.join(out_dir, fn))
        index["weight_map"].update({k: fn for k in part})
    json.dump(index, open(os.path.join(out_dir, "model.safetensors.index.json"), "w"))
    json.dump(hf_config_from_dims(d), open(os.path.join(out_dir, "config.json"), "w"))


def load_code2wav_checkpoint(path: str) -> tuple[dict, dict]:
    """The speech tokenizer's DECODER out of a Qwen3-TTS checkpoint: (decoder_config dict, state dict under the decoder's own
    parameter names) for ``code2wav.Code2WavDecoder``.  ``path`` is the model directory or its ``speech_tokenizer/`` sub-directory
    (qwen3_tts_code2wav.py:59-75 resolves ``speech_tokenizer/config.json`` and loads the whole tokenizer in fp32); the tokenizer
    model's tensors are named ``decoder.…`` / ``encoder.…`` (Qwen3TTSTokenizerV2Model, modeling_qwen3_tts_tokenizer_v2.py:1073-1100):
    only the ``decoder.`` ones are read, the prefix is stripped, dtype fp32 as stored."""
    d = os.path.join(path, "speech_tokenizer") if os.path.isdir(os.path.join(path, "speech_tokenizer")) else path
    with open(os.path.join(d, "config.json")) as f:
        cfg = json.load(f)
    dec_cfg = dict(cfg.get("decoder_config") or {})
    if "num_quantizers" not in dec_cfg:
        raise ValueError("speech_tokenizer decoder_config.num_quantizers not found")
    dec_cfg.setdefault("output_sample_rate", cfg.get("output_sample_rate", 24000))
    state = {}
    for name, t in iter_safetensors(d):
        if name.startswith("decoder."):
            state[name[len("decoder."):]] = t.to(torch.float32)
    if not state:
        raise ValueError(f"no decoder.* tensors under {d}")
    return dec_cfg, state
