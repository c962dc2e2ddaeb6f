"""Payload types crossing the scheduler <-> worker boundary, field-for-field what the reference uses.

The reference's types subclass vLLM dataclasses (third party, not importable here); these standalone
dataclasses carry the same field names so a scheduler written against the reference can hand its
objects over unchanged (duck typing) -- see INTEGRATION.md.

  OmniNewRequestData / OmniCachedRequestData / OmniSchedulerOutput   V/core/sched/output.py:9-77
  OmniModelRunnerOutput                                              V/outputs.py:12-26
  AdditionalInformationEntry / AdditionalInformationPayload         V/engine/__init__.py:29-57 (+ serialization.py:19-113)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

import numpy as np
import torch


@dataclass
class SamplingParams:
    """Subset of vLLM SamplingParams the talker stage sets (stage_configs/qwen3_tts.yaml:27-34)."""
    temperature: float = 0.9
    top_k: int = 50
    top_p: float = 1.0
    repetition_penalty: float = 1.05
    seed: int | None = 42
    max_tokens: int = 4096
    stop_token_ids: tuple[int, ...] = (2150,)
    logprobs: int | None = None        # vLLM: the sampled token's log-probability + this many top alternatives per generated token

    @property
    def greedy(self) -> bool:
        return self.temperature == 0.0


@dataclass
class OmniNewRequestData:
    req_id: str
    prompt_token_ids: list[int] | None
    block_ids: tuple[list[int], ...]
    num_computed_tokens: int = 0
    sampling_params: SamplingParams | None = None
    prompt_embeds: torch.Tensor | None = None
    external_req_id: str | None = None
    additional_information: Any = None      # AdditionalInformationPayload (the reference's wire type) or a plain dict


@dataclass
class OmniCachedRequestData:
    req_ids: list[str] = field(default_factory=list)
    resumed_from_preemption: list[bool] = field(default_factory=list)
    new_token_ids: list[list[int]] = field(default_factory=list)
    new_block_ids: list[tuple[list[int], ...] | None] = field(default_factory=list)
    num_computed_tokens: list[int] = field(default_factory=list)


@dataclass
class OmniSchedulerOutput:
    scheduled_new_reqs: list[OmniNewRequestData] = field(default_factory=list)
    scheduled_cached_reqs: OmniCachedRequestData = field(default_factory=OmniCachedRequestData)
    num_scheduled_tokens: dict[str, int] = field(default_factory=dict)
    total_num_scheduled_tokens: int = 0
    finished_req_ids: set[str] = field(default_factory=set)
    preempted_req_ids: set[str] = field(default_factory=set)
    # omni: {req_id: {"seq_len": int, "block_ids": [...], "custom_metadata": {...}?}}  (output.py:73-77)
    finished_requests_needing_kv_transfer: dict[str, dict] = field(default_factory=dict)


@dataclass
class LogprobsLists:
    """vLLM ``v1.outputs.LogprobsLists``: one entry per OUTPUT ROW of the step (index = req_id_to_index), each covering the tokens
    that row sampled this step (0 or 1 here): ids / log-probabilities of [sampled token, top-1, top-2, ...] and the sampled token's
    rank (1 = it was the most probable)."""
    logprob_token_ids: list[list[int]]
    logprobs: list[list[float]]
    sampled_token_ranks: list[int]

    def slice_request(self, req_index: int, num_tokens: int = 1, num_logprobs: int | None = None) -> "LogprobsLists":
        n = None if num_logprobs is None else num_logprobs + 1
        return LogprobsLists([self.logprob_token_ids[req_index][:n]] if num_tokens else [], [self.logprobs[req_index][:n]] if num_tokens else [],
                             [self.sampled_token_ranks[req_index]] if num_tokens else [])


@dataclass
class OmniModelRunnerOutput:
    req_ids: list[str]
    req_id_to_index: dict[str, int]
    sampled_token_ids: list[list[int]]
    logprobs: Any = None
    prompt_logprobs_dict: dict = field(default_factory=dict)
    pooler_output: list[dict[str, Any]] | None = None
    kv_connector_output: Any = None
    num_nans_in_logits: dict | None = None
    cudagraph_stats: Any = None
    multimodal_outputs: dict[str, torch.Tensor] | None = None
    kv_extracted_req_ids: list[str] | None = None


EMPTY_MODEL_RUNNER_OUTPUT = OmniModelRunnerOutput(req_ids=[], req_id_to_index={}, sampled_token_ids=[])


# ---- additional_information wire format (V/engine/__init__.py:29-57, V/engine/serialization.py:19-113; consumer
# V/worker/gpu_model_runner.py:915-936).  The reference ships an AdditionalInformationPayload msgspec struct whose
# `entries` maps each key to an AdditionalInformationEntry in exactly one of three forms:
#   tensor  tensor_data (raw bytes, row-major) + tensor_shape + tensor_dtype (a numpy dtype name: dtype_to_name)
#   list    list_data        scalar  scalar_data
# msgspec is not a dependency of this package: the two structs are restated as dataclasses with the same field names, and
# the decoder is duck-typed on those names, so it takes the reference's own objects unchanged.
@dataclass
class AdditionalInformationEntry:
    tensor_data: bytes | None = None
    tensor_shape: list[int] | None = None
    tensor_dtype: str | None = None
    list_data: list[Any] | None = None
    scalar_data: Any | None = None


@dataclass
class AdditionalInformationPayload:
    entries: dict[str, AdditionalInformationEntry] = field(default_factory=dict)


_DTYPE_NAMES = {torch.float32: "float32", torch.float16: "float16", torch.bfloat16: "bfloat16", torch.float64: "float64",
                torch.int64: "int64", torch.int32: "int32", torch.int16: "int16", torch.int8: "int8", torch.uint8: "uint8",
                torch.bool: "bool"}


def dtype_to_name(dtype: torch.dtype) -> str:
    """serialization.py:19-39"""
    return _DTYPE_NAMES.get(dtype, str(dtype).replace("torch.", ""))


def _tensor_bytes(t: torch.Tensor) -> bytes:
    t = t.detach().to("cpu").contiguous()
    # numpy has no bfloat16 (the reference's `.numpy().tobytes()` raises on it): the bytes are taken through a uint8 view
    return t.reshape(-1).view(torch.uint8).numpy().tobytes() if t.numel() else b""


def _tensor_from_bytes(data: bytes, shape, dtype_name: str) -> torch.Tensor:
    name = str(dtype_name).replace("torch.", "")
    shape = tuple(shape) if shape is not None else ()
    if name == "bfloat16":                       # extension: the reference's decoder goes through np.dtype and cannot
        raw = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy())
        return raw.view(torch.bfloat16).reshape(shape)
    arr = np.frombuffer(data, dtype=np.dtype(name)).reshape(shape)
    return torch.from_numpy(arr.copy())


def serialize_additional_information(raw_info: dict[str, Any] | AdditionalInformationPayload | None
                                     ) -> AdditionalInformationPayload | None:
    """serialization.py:42-71: tensors -> byte entries, lists -> list_data, anything else -> scalar_data."""
    if raw_info is None:
        return None
    if hasattr(raw_info, "entries"):
        return raw_info
    entries: dict[str, AdditionalInformationEntry] = {}
    for key, value in raw_info.items():
        if isinstance(value, torch.Tensor):
            entries[key] = AdditionalInformationEntry(tensor_data=_tensor_bytes(value), tensor_shape=list(value.shape),
                                                      tensor_dtype=dtype_to_name(value.dtype))
        elif isinstance(value, list):
            entries[key] = AdditionalInformationEntry(list_data=value)
        else:
            entries[key] = AdditionalInformationEntry(scalar_data=value)
    return AdditionalInformationPayload(entries=entries) if entries else None


def deserialize_additional_information(payload: Any) -> dict[str, Any]:
    """serialization.py:73-113: None -> {}; an object with an `.entries` dict (the reference's AdditionalInformationPayload
    or the dataclass above) is decoded entry by entry; a plain dict is taken as it is, except that values in this package's
    earlier {tensor_data, shape, dtype} dict form are decoded too.  A payload that cannot be decoded is an error HERE (the
    reference logs and returns {}: a talker request without its prompt embeddings cannot run, so the runner refuses it)."""
    if payload is None:
        return {}
    if isinstance(payload, dict):
        out: dict[str, Any] = {}
        for k, v in payload.items():
            if isinstance(v, dict) and "tensor_data" in v:
                out[k] = _tensor_from_bytes(v["tensor_data"], v.get("shape", v.get("tensor_shape")), v.get("dtype", v.get("tensor_dtype", "float32")))
            elif hasattr(v, "tensor_data") or hasattr(v, "list_data") or hasattr(v, "scalar_data"):
                out[k] = _decode_entry(v)
            else:
                out[k] = v
        return out
    entries = getattr(payload, "entries", None)
    if not isinstance(entries, dict):
        raise TypeError(f"additional_information: expected a dict or an object with an `entries` dict, got {type(payload).__name__}")
    return {k: _decode_entry(e) for k, e in entries.items()}


def _decode_entry(entry: Any) -> Any:
    if getattr(entry, "tensor_data", None) is not None:
        return _tensor_from_bytes(entry.tensor_data, getattr(entry, "tensor_shape", ()), getattr(entry, "tensor_dtype", None) or "float32")
    if getattr(entry, "list_data", None) is not None:
        return entry.list_data
    if getattr(entry, "scalar_data", None) is not None:
        return entry.scalar_data
    return None


decode_additional_information = deserialize_additional_information


def encode_tensor(t: torch.Tensor) -> dict[str, Any]:
    """One tensor in the plain-dict form (tests / scripts; the wire form is serialize_additional_information)."""
    return {"tensor_data": _tensor_bytes(t), "shape": list(t.shape), "dtype": dtype_to_name(t.dtype)}
