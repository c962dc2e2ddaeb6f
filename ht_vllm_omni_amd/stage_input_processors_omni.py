"""Stage hand-offs of the Qwen3-Omni pipeline on either side of the talker:
thinker -> talker and talker -> Code2Wav.

Mirror of V/model_executor/stage_input_processors/qwen3_omni.py:22-420 (same function names, arguments, payload keys,
errors); pinned by known answers minted from that file (tests/golden/omni_stage_processors.pt, make_fixtures.py `os`).
The thinker side hands over, per request: the layer-"0" embeddings and layer-"24" hidden states of the whole sequence,
the token ids, and the thinker's tts bos / eos / pad embeddings; the talker prompt is a zero-filled placeholder whose
LENGTH is what matters (`_compute_talker_prompt_ids_length`).  In streaming mode the first chunk may arrive in two
pieces (chunked thinker prefill): the first piece is parked in `transfer_manager.request_payload` and concatenated.
"""
from __future__ import annotations

from typing import Any

import numpy as np
import torch

from .stage_input_processors import (FrameBuffer, OmniTokensPrompt, _validate_stage_inputs, extract_language_from_prompt,
                                     extract_language_from_request, extract_speaker_from_prompt,
                                     extract_speaker_from_request)

# chat-template token ids of the Qwen3-Omni thinker tokenizer (qwen3_omni.py:23-26)
_IM_START, _SYSTEM, _USER, _ASSISTANT = 151644, 8948, 872, 77091


def _compute_talker_prompt_ids_length(info: dict, device: torch.device | str = "cpu") -> int:
    """Length of the talker's placeholder prompt: every user turn in full, system turns dropped, and 9 positions
    (3 + 4 + 1 + 1) for the trailing assistant turn (qwen3_omni.py:22-57)."""
    seq_len = len(info["thinker_sequences"])
    ids = np.asarray(info["thinker_input_ids"], dtype=np.int64)
    starts = np.flatnonzero(ids == _IM_START).tolist() + [seq_len]
    total = 0
    for i in range(len(starts) - 1):
        s, e = starts[i], starts[i + 1]
        role = int(ids[s + 1])
        if role == _USER:
            total += e - s
        elif role == _ASSISTANT and i == len(starts) - 2:
            total += 9
    return total


def _ensure_list(x):
    """vLLM ConstantList (wraps `_x`) -> list; lists are copied; anything else is passed through."""
    if hasattr(x, "_x"):
        return list(x._x)
    if not isinstance(x, list):
        return x
    return list(x)


def _tag(info: dict, request: Any) -> None:
    speaker = extract_speaker_from_request(request)
    if speaker is not None:
        info["speaker"] = speaker
    language = extract_language_from_request(request)
    if language is not None:
        info["language"] = language


def thinker2talker_async_chunk(transfer_manager: Any, pooling_output: dict[str, Any], request: Any,
                               is_finished: bool = False):
    """One thinker step -> the talker's next additional-information chunk, or None while the first chunk is parked."""
    request_id = request.external_req_id
    chunk_id = transfer_manager.put_req_chunk[request_id]
    if chunk_id == 0:
        info = {
            "thinker_prefill_embeddings": pooling_output.get("0").detach().cpu(),
            "thinker_hidden_states": pooling_output.get("24").detach().cpu(),
            "thinker_sequences": _ensure_list(request.all_token_ids),
            "thinker_input_ids": _ensure_list(request.prompt_token_ids),
            "tts_bos_embed": pooling_output.get("tts_bos_embed").detach().cpu(),
            "tts_eos_embed": pooling_output.get("tts_eos_embed").detach().cpu(),
            "tts_pad_embed": pooling_output.get("tts_pad_embed").detach().cpu(),
            "finished": torch.tensor(is_finished, dtype=torch.bool),
        }
        _tag(info, request)
        if transfer_manager.request_payload.get(request_id) is None:
            if not is_finished:
                transfer_manager.request_payload[request_id] = info
                return None
        else:
            saved = transfer_manager.request_payload.pop(request_id)
            for key in ("thinker_prefill_embeddings", "thinker_hidden_states"):
                info[key] = torch.cat((saved.get(key), info.get(key)), dim=0)
        return info
    info = {"finished": torch.tensor(is_finished, dtype=torch.bool)}
    _tag(info, request)
    output_token_ids = _ensure_list(request.output_token_ids)
    if output_token_ids:
        info["override_keys"] = ["thinker_decode_embeddings", "thinker_output_token_ids"]
        info["thinker_decode_embeddings"] = pooling_output.get("0").detach().cpu()
        info["thinker_output_token_ids"] = output_token_ids
    else:       # a later piece of a chunked thinker prefill
        info["thinker_prefill_embeddings"] = pooling_output.get("0").detach().cpu()
        info["thinker_hidden_states"] = pooling_output.get("24").detach().cpu()
    return info


def thinker2talker(stage_list: list[Any], engine_input_source: list[int], prompt: Any = None,
                   requires_multimodal_data: bool = False, device: torch.device | str | None = None) -> list[OmniTokensPrompt]:
    """Finished thinker outputs -> talker prompts (zero placeholder ids of the computed length + fp32 tensors)."""
    outputs = _validate_stage_inputs(stage_list, engine_input_source)
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    dev = torch.device(device)
    prompts: list[OmniTokensPrompt] = []
    for i, thinker_output in enumerate(outputs):
        out = thinker_output.outputs[0]
        mm = out.multimodal_output
        info = {
            "thinker_prefill_embeddings": mm["0"].detach().to(device=dev, dtype=torch.float),
            "thinker_hidden_states": mm["24"].detach().to(device=dev, dtype=torch.float),
            "thinker_sequences": thinker_output.prompt_token_ids + out.token_ids,
            "thinker_input_ids": thinker_output.prompt_token_ids,
            "tts_bos_embed": mm["tts_bos_embed"].detach().to(device=dev, dtype=torch.float),
            "tts_eos_embed": mm["tts_eos_embed"].detach().to(device=dev, dtype=torch.float),
            "tts_pad_embed": mm["tts_pad_embed"].detach().to(device=dev, dtype=torch.float),
        }
        speaker = extract_speaker_from_prompt(prompt, index=i)
        if speaker is not None:
            info["speaker"] = speaker
        language = extract_language_from_prompt(prompt, index=i)
        if language is not None:
            info["language"] = language
        prompts.append(OmniTokensPrompt(prompt_token_ids=[0] * _compute_talker_prompt_ids_length(info, device=dev),
                                        additional_information=info))
    return prompts


def talker2code2wav_async_chunk(transfer_manager: Any, pooling_output: dict[str, Any], request: Any, is_finished: bool = False):
    """Omni flavour of the streaming window (qwen3_omni.py:238-314): fixed `codec_chunk_frames` cadence, no initial
    phase, left context clipped to what exists; frames arrive as `code_predictor_codes` [Q, 1] per step."""
    if "code_predictor_codes" not in pooling_output:
        return None
    connector = getattr(transfer_manager, "connector", None)
    raw = getattr(connector, "config", {}) or {}
    cfg = raw.get("extra", raw) if isinstance(raw, dict) else {}
    chunk, left_cfg = int(cfg.get("codec_chunk_frames", 25)), int(cfg.get("codec_left_context_frames", 25))
    codes = pooling_output["code_predictor_codes"]
    if codes is None:
        return None
    if not isinstance(codes, torch.Tensor):
        if hasattr(codes, "__len__") and len(codes) == 0:
            return None
        codes = torch.tensor(codes, dtype=torch.long)
    if codes.numel() == 0 or not codes.any():
        return None
    frame = codes.to(torch.long).transpose(0, 1).cpu().reshape(-1)
    if int(frame.sum()) == 0:
        return None
    rid = request.external_req_id
    transfer_manager.code_prompt_token_ids[rid].append(frame.tolist())
    frames = transfer_manager.code_prompt_token_ids[rid]
    length = len(frames)
    tail = length % chunk
    if tail != 0 and not is_finished:
        return None
    context = tail if tail != 0 else chunk
    left = max(0, min(length - context, left_cfg))
    end = min(length, left + context)
    window = frames.tail(end) if isinstance(frames, FrameBuffer) else np.asarray(frames[-end:], dtype=np.int64)
    return {"code_predictor_codes": np.ascontiguousarray(window.T).reshape(-1).tolist(), "left_context_size": left,
            "finished": torch.tensor(is_finished, dtype=torch.bool)}


def talker2code2wav(stage_list: list[Any], engine_input_source: list[int], prompt: Any = None,
                    requires_multimodal_data: bool = False) -> list[OmniTokensPrompt]:
    """All codes of a finished talker request, last len(token_ids) - 1 frames, codebook-major flat (qwen3_omni.py:317-364)."""
    prompts = []
    for talker_output in _validate_stage_inputs(stage_list, engine_input_source):
        out = talker_output.outputs[0]
        seq_len = len(out.token_ids) - 1
        codes = out.multimodal_output["code_predictor_codes"][-seq_len:].to(torch.long).transpose(0, 1).cpu().reshape(-1).tolist()
        prompts.append(OmniTokensPrompt(prompt_token_ids=codes))
    return prompts
