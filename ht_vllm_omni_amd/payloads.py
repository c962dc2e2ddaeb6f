"""Payload types crossing the scheduler <-> worker boundary, field-for-field what the reference uses.

The reference's types subclass vLLM dataclasses (third party, not importable here); these standalone
dataclasses carry the same field names so a scheduler written against the reference can hand its
objects over unchanged (duck typing) -- see INTEGRATION.md.

  OmniNewRequestData / OmniCachedRequestData / OmniSchedulerOutput   V/core/sched/output.py:9-77
  OmniModelRunnerOutput                                              V/outputs.py:12-26
  AdditionalInformationPayload (tensor entries as raw bytes)         V/engine/__init__.py:16-85
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
    additional_information: dict[str, Any] | None = None


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


# ---- additional_information wire format: tensors travel as {tensor_data bytes, shape, dtype}
# (V/engine/serialization.py:73-113; decode helper V/worker/gpu_model_runner.py:883-936)
def decode_additional_information(info: dict[str, Any] | None) -> dict[str, Any]:
    out: dict[str, Any] = {}
    for k, v in (info or {}).items():
        if isinstance(v, dict) and "tensor_data" in v:
            dt = getattr(torch, str(v["dtype"]).replace("torch.", ""))
            raw = np.frombuffer(v["tensor_data"], dtype=np.uint8).copy()
            out[k] = torch.from_numpy(raw).view(dt).reshape(tuple(v["shape"]))
        else:
            out[k] = v
    return out


def encode_tensor(t: torch.Tensor) -> dict[str, Any]:
    t = t.detach().cpu().contiguous()
    return {"tensor_data": t.view(torch.uint8).numpy().tobytes(), "shape": list(t.shape), "dtype": str(t.dtype).replace("torch.", "")}
