"""Talker -> Code2Wav hand-off: codec-frame accumulation, streaming windows, payloads.

Host-side mirror of the reference's stage input processor for Qwen3-TTS
(V/model_executor/stage_input_processors/qwen3_tts.py:22-270, chunk_size_utils.py:5-33,
tts_utils.py:47-190): same function names, arguments, payload keys and error behaviour, so it can be
named in a stage YAML's `custom_process_input_func` in place of the reference's.

What changes is the data path, not the rules:
  * the reference appends `frame.cpu().tolist()` per request per step to a list of lists and builds the
    codebook-major window with a Q x F Python double loop; here frames live in one growable int32 array per
    request (`FrameBuffer`) fed from the runner's single per-step device->host copy of `audio_codes [B, Q]`
    (`CodecChunkStreamer.on_step`), and a window is a slice + transpose;
  * the emit rule is a pure integer function (`window_plan`) that the golden sweep pins against the reference.
Plain lists of lists (what the reference's transfer manager holds) are accepted everywhere a FrameBuffer is.
"""
from __future__ import annotations

import logging
from collections import defaultdict
from typing import Any

import numpy as np
import torch

logger = logging.getLogger(__name__)

_CODEBOOK_SIZE = 2048     # qwen3_tts.py:49 (stop id 2150 and friends are not codec frames)


# ---------------------------------------------------------------------------------------------------------
# initial-chunk-size policy (chunk_size_utils.py:5-33)
# ---------------------------------------------------------------------------------------------------------
def max_ic_for_chunk_size(chunk_size: int) -> int:
    """Largest power of two strictly below chunk_size (1 for chunk_size <= 2)."""
    if chunk_size <= 2:
        return 1
    return 1 << ((chunk_size - 1).bit_length() - 1)


def compute_dynamic_initial_chunk_size(active_requests: int, max_num_seqs: int, max_ic: int) -> int:
    """IC from the power-of-two ladder [2, 4, .., max_ic] by load factor: small IC (early first audio) when idle,
    large IC (fewer vocoder calls) when the batch is full."""
    steps = []
    v = 2
    while v <= max_ic:
        steps.append(v)
        v <<= 1
    if not steps:
        return max(1, max_ic)
    if max_num_seqs <= 0:
        return steps[0]
    load = min(active_requests / max_num_seqs, 1.0)
    return steps[int(round(load * (len(steps) - 1)))]


def window_plan(length: int, finished: bool, chunk_size: int, left_context: int, initial_chunk_size: int):
    """When `length` frames have accumulated: None (hold) or (end_index, left_context_size) -- the window is the last
    `end_index` frames, the first `left_context_size` of them decoder context (qwen3_tts.py:203-228).
    Initial phase (0 < IC < chunk, length < chunk): emit every IC frames; afterwards every `chunk_size` frames counted
    from the initial coverage, so that no frame is replayed at the transition; `finished` flushes the tail."""
    if length <= 0:
        return None
    ic = min(initial_chunk_size, chunk_size)
    if 0 < ic < chunk_size and length < chunk_size:
        rem = length % ic
        if not finished and rem != 0:
            return None
        context_length = rem if (finished and rem != 0) else ic
    else:
        coverage = ((chunk_size - 1) // ic) * ic if 0 < ic < chunk_size else 0
        adjusted = length - coverage
        if not finished and adjusted % chunk_size != 0:
            return None
        tail = adjusted % chunk_size
        context_length = tail if tail != 0 else chunk_size
    end_index = min(length, left_context + context_length)
    return end_index, max(0, end_index - context_length)


# ---------------------------------------------------------------------------------------------------------
# frame storage
# ---------------------------------------------------------------------------------------------------------
class FrameBuffer:
    """Growable [n, Q] int32 array of one request's codec frames; list-like where the processors need it."""

    __slots__ = ("_a", "_n")

    def __init__(self, q: int = 16, capacity: int = 64):
        self._a = np.empty((capacity, q), dtype=np.int32)
        self._n = 0

    def __len__(self) -> int:
        return self._n

    def append(self, frame) -> None:
        f = np.asarray(frame, dtype=np.int32).reshape(-1)
        if self._n == 0 and f.shape[0] != self._a.shape[1]:
            self._a = np.empty((self._a.shape[0], f.shape[0]), dtype=np.int32)
        if self._n == self._a.shape[0]:
            self._a = np.concatenate([self._a, np.empty_like(self._a)], axis=0)
        self._a[self._n] = f
        self._n += 1

    def tail(self, n: int) -> np.ndarray:
        return self._a[max(self._n - n, 0): self._n]

    def __getitem__(self, idx):
        return self._a[: self._n][idx]


def _tail_frames(frames, n: int) -> np.ndarray:
    """Last n frames of a FrameBuffer or of the reference's list of lists, as [n, Q] int array."""
    if isinstance(frames, FrameBuffer):
        return frames.tail(n)
    return np.asarray(frames[-n:], dtype=np.int64).reshape(len(frames[-n:]), -1)


def _codebook_major(window: np.ndarray) -> list[int]:
    """[F, Q] frames -> flat [Q * F] list, codebook-major (what Code2Wav expects, qwen3_tts.py:95,250-252)."""
    return np.ascontiguousarray(window.T).reshape(-1).tolist()


# ---------------------------------------------------------------------------------------------------------
# speaker / language pass-through (tts_utils.py:47-190)
# ---------------------------------------------------------------------------------------------------------
def _entry_list(request: Any, key: str):
    info = getattr(request, "additional_information", None)
    entries = getattr(info, "entries", None) if info is not None else None
    if not isinstance(entries, dict):
        return None
    entry = entries.get(key)
    data = getattr(entry, "list_data", None) if entry is not None else None
    return data if isinstance(data, list) and data else None


def extract_speaker_from_request(request: Any):
    data = _entry_list(request, "speaker")
    if data is None:
        return None
    v = data[0]
    return v.lower().strip() if isinstance(v, str) else str(v).lower().strip()


def extract_language_from_request(request: Any):
    return _entry_list(request, "language")


def _from_prompt(prompt: Any, index: int, key: str):
    if prompt is None:
        return None
    p = prompt[index] if isinstance(prompt, list) and index < len(prompt) else prompt
    if p is None:
        return None
    info = p.get("additional_information")
    if not isinstance(info, dict):
        return None
    v = info.get(key)
    return v if isinstance(v, list) and v else None


def extract_speaker_from_prompt(prompt: Any, index: int = 0):
    return _from_prompt(prompt, index, "speaker")


def extract_language_from_prompt(prompt: Any, index: int = 0):
    return _from_prompt(prompt, index, "language")


# ---------------------------------------------------------------------------------------------------------
# streaming (async chunk) processor
# ---------------------------------------------------------------------------------------------------------
def _extract_last_frame(pooling_output: dict[str, Any]):
    """The step's new frame, or None for an empty / all-zero (padding, EOS) one (qwen3_tts.py:119-131)."""
    codes = pooling_output.get("audio_codes")
    if isinstance(codes, np.ndarray):
        codes = torch.from_numpy(codes)
    if not isinstance(codes, torch.Tensor) or codes.numel() == 0:
        return None
    if codes.ndim == 2:
        frame = codes[-1]
        if frame.numel() == 0 or not bool(frame.any().item()):
            return None
        return frame.to(torch.long).reshape(-1)
    if codes.ndim == 1:
        return codes.to(torch.long).reshape(-1)
    raise ValueError(f"Invalid audio_codes shape for Qwen3-TTS async_chunk: {tuple(codes.shape)}")


def _chunk_config(transfer_manager: Any) -> tuple[int, int]:
    connector = getattr(transfer_manager, "connector", None)
    raw = getattr(connector, "config", {}) or {}
    cfg = raw.get("extra", raw) if isinstance(raw, dict) else {}
    return int(cfg.get("codec_chunk_frames", 25)), int(cfg.get("codec_left_context_frames", 25))


def _initial_chunk_size(transfer_manager: Any, request: Any, request_id, chunk_size: int) -> int:
    """Per-request override (`initial_codec_chunk_frames`) or the load-dependent IC, cached per request so that the
    emit boundaries of a running request never move (qwen3_tts.py:160-185)."""
    data = _entry_list(request, "initial_codec_chunk_frames")
    info = getattr(request, "additional_information", None)
    if data is not None and len(data) == 1 and hasattr(info, "entries"):
        return int(data[0])
    cache = getattr(transfer_manager, "_cached_ic", None)
    if cache is None:
        cache = {}
        transfer_manager._cached_ic = cache
    if request_id not in cache:
        active = sum(1 for v in transfer_manager.code_prompt_token_ids.values() if len(v) > 0)
        capacity = getattr(transfer_manager, "scheduler_max_num_seqs", 1)
        cache[request_id] = compute_dynamic_initial_chunk_size(active, capacity, max_ic_for_chunk_size(chunk_size))
    return cache[request_id]


def talker2code2wav_async_chunk(transfer_manager: Any, pooling_output: dict[str, Any] | None, request: Any,
                                is_finished: bool = False) -> dict[str, Any] | None:
    """One talker step of one request -> None (hold) or the payload of the next Code2Wav chunk
    `{code_predictor_codes (codebook-major flat), left_context_size, finished[, speaker, language]}`."""
    request_id = request.external_req_id
    finished = bool(is_finished or request.is_finished())
    payloads = getattr(transfer_manager, "request_payload", None)
    if payloads is None:
        payloads = {}
        transfer_manager.request_payload = payloads

    if isinstance(pooling_output, dict):
        frame = _extract_last_frame(pooling_output)
        if frame is not None:
            transfer_manager.code_prompt_token_ids[request_id].append(frame.cpu().tolist())
        ref_code = pooling_output.get("ref_code")
        if isinstance(ref_code, torch.Tensor) and ref_code.numel() > 0 and payloads.get(request_id) is None:
            payloads[request_id] = ref_code.to(torch.long).cpu().contiguous()
    elif not finished:
        return None

    chunk_size, left_context = _chunk_config(transfer_manager)
    ic = _initial_chunk_size(transfer_manager, request, request_id, chunk_size)
    if chunk_size <= 0 or left_context < 0 or ic < 0:
        raise ValueError(f"Invalid codec chunk config: codec_chunk_frames={chunk_size}, "
                         f"codec_left_context_frames={left_context}, initial_codec_chunk_frames={ic}")
    if ic > chunk_size:
        logger.warning("initial_codec_chunk_frames=%d > codec_chunk_frames=%d, clamping to codec_chunk_frames.", ic, chunk_size)
        ic = chunk_size

    frames = transfer_manager.code_prompt_token_ids[request_id]
    length = len(frames)
    if length <= 0:
        return {"code_predictor_codes": [], "finished": True} if finished else None
    plan = window_plan(length, finished, chunk_size, left_context, ic)
    if plan is None:
        return None
    end_index, left_context_size = plan
    window = _tail_frames(frames, end_index)

    # the voice-clone reference codes ride in front of EVERY chunk as decoder context (kept, not popped)
    ref_code = payloads.get(request_id)
    if isinstance(ref_code, torch.Tensor) and ref_code.numel() > 0:
        ref = ref_code.numpy().reshape(-1, window.shape[1]) if ref_code.ndim != 2 else ref_code.numpy()
        window = np.concatenate([ref.astype(window.dtype), window], axis=0)
        left_context_size += ref.shape[0]

    info: dict[str, Any] = {"code_predictor_codes": _codebook_major(window), "left_context_size": left_context_size,
                            "finished": finished}
    speaker = extract_speaker_from_request(request)
    if speaker is not None:
        info["speaker"] = speaker
    language = extract_language_from_request(request)
    if language is not None:
        info["language"] = language
    return info


# ---------------------------------------------------------------------------------------------------------
# non-streaming processor
# ---------------------------------------------------------------------------------------------------------
class OmniTokensPrompt(dict):
    """Field-compatible stand-in for vllm_omni.inputs.data.OmniTokensPrompt (a TypedDict there)."""

    def __init__(self, prompt_token_ids, multi_modal_data=None, mm_processor_kwargs=None, additional_information=None):
        super().__init__(prompt_token_ids=prompt_token_ids, multi_modal_data=multi_modal_data,
                         mm_processor_kwargs=mm_processor_kwargs, additional_information=additional_information)


def _validate_stage_inputs(stage_list, engine_input_source):
    """qwen3_omni.py:72-84."""
    if not engine_input_source:
        raise ValueError("engine_input_source cannot be empty")
    stage_id = engine_input_source[0]
    if stage_id >= len(stage_list):
        raise IndexError(f"Invalid stage_id: {stage_id}")
    stage = stage_list[stage_id]
    if stage.engine_outputs is None:
        raise RuntimeError(f"Stage {stage_id} has no outputs yet")
    return stage.engine_outputs


def _normalise_ref_code(ref_code, ref_code_len, num_quantizers: int):
    """-> (ref [n, Q] long tensor or None, n) following qwen3_tts.py:53-92."""
    if isinstance(ref_code_len, torch.Tensor):
        ref_code_len = int(ref_code_len.reshape(-1)[-1].item()) if ref_code_len.numel() > 0 else 0
    elif ref_code_len is None:
        ref_code_len = 0
    else:
        ref_code_len = int(ref_code_len)
    if isinstance(ref_code, list):
        ref_code = ref_code[0] if ref_code else None
    if not (isinstance(ref_code, torch.Tensor) and ref_code.numel() > 0):
        return None, 0
    ref_code = ref_code.to(torch.long).cpu().contiguous()
    if ref_code.ndim == 1:
        if ref_code.numel() % num_quantizers != 0:
            logger.warning("Ignoring malformed ref_code with %d elements not divisible by num_quantizers=%d",
                           ref_code.numel(), num_quantizers)
            return None, 0
        ref_code = ref_code.reshape(-1, num_quantizers)
    elif ref_code.ndim != 2:
        logger.warning("Ignoring malformed ref_code shape %s", tuple(ref_code.shape))
        return None, 0
    if ref_code_len > 0 and int(ref_code.shape[0]) > ref_code_len:
        logger.warning("Trimming ref_code from %d frames to ref_code_len=%d before Code2Wav.", int(ref_code.shape[0]), ref_code_len)
        ref_code = ref_code[:ref_code_len]
    return ref_code, int(ref_code.shape[0])


def talker2code2wav(stage_list: list[Any], engine_input_source: list[int], prompt: Any = None,
                    requires_multimodal_data: bool = False) -> list[Any]:
    """All of a finished request's codes at once: drop zero-padded / out-of-range frames, keep at most
    len(token_ids) - 1 trailing frames, prepend the reference codes, flatten codebook-major (qwen3_tts.py:22-116)."""
    outputs = _validate_stage_inputs(stage_list, engine_input_source)
    prompts: list[OmniTokensPrompt] = []
    for i, talker_output in enumerate(outputs):
        if not talker_output.finished:
            continue
        output = talker_output.outputs[0]
        codes = output.multimodal_output["audio_codes"].to(torch.long)
        seq_len = max(len(output.token_ids) - 1, 0)
        valid = codes.any(dim=1) & (codes.max(dim=1).values < _CODEBOOK_SIZE)
        codes = codes[valid]
        if seq_len > 0 and codes.ndim == 2 and int(codes.shape[0]) > seq_len:
            codes = codes[-seq_len:]
        nq = int(codes.shape[1]) if codes.ndim == 2 and codes.shape[1] > 0 else 16
        ref, ref_len = _normalise_ref_code(output.multimodal_output.get("ref_code"),
                                           output.multimodal_output.get("ref_code_len"), nq)
        if ref is not None:
            codes = torch.cat([ref.to(codes.device), codes], dim=0)
        info: dict[str, Any] = {}
        if ref_len > 0:
            info["left_context_size"] = [ref_len]
        speaker = extract_speaker_from_prompt(prompt, index=i)
        if speaker is not None:
            info["speaker"] = speaker
        language = extract_language_from_prompt(prompt, index=i)
        if language is not None:
            info["language"] = language
        prompts.append(OmniTokensPrompt(prompt_token_ids=_codebook_major(codes.cpu().numpy()),
                                        additional_information=info if info else None))
    return prompts


# ---------------------------------------------------------------------------------------------------------
# batch front end for the runner: one device->host copy of the step's codes feeds every request
# ---------------------------------------------------------------------------------------------------------
class CodecChunkStreamer:
    """Owns what the reference's chunk transfer adapter keeps for this hop (`code_prompt_token_ids`, `request_payload`,
    the IC cache, `put_req_chunk`) and turns one decode step of the whole batch into the chunk payloads that are due.

    on_step(req_ids, audio_codes [B, Q] host int array, finished) -> [(req_id, payload)], same payloads as calling
    `talker2code2wav_async_chunk` once per request with that request's row."""

    def __init__(self, *, codec_chunk_frames: int = 25, codec_left_context_frames: int = 25, max_num_seqs: int = 1,
                 num_quantizers: int = 16, connector: Any = None):
        self.code_prompt_token_ids = defaultdict(lambda: FrameBuffer(num_quantizers))
        self.request_payload: dict = {}
        self.put_req_chunk = defaultdict(int)
        self.scheduler_max_num_seqs = max_num_seqs
        self.connector = connector if connector is not None else _Cfg(codec_chunk_frames, codec_left_context_frames)
        if connector is not None and not isinstance(getattr(connector, "config", None), dict):
            connector.config = {"extra": {"codec_chunk_frames": codec_chunk_frames,
                                          "codec_left_context_frames": codec_left_context_frames}}

    def on_step(self, requests: list[Any], audio_codes, finished: list[bool] | None = None, ref_codes: dict | None = None):
        codes = audio_codes.cpu().numpy() if isinstance(audio_codes, torch.Tensor) else np.asarray(audio_codes)
        nonzero = codes.any(axis=1)
        out = []
        for b, req in enumerate(requests):
            rid = req.external_req_id
            fin = bool(finished[b]) if finished is not None else bool(req.is_finished())
            if nonzero[b]:
                self.code_prompt_token_ids[rid].append(codes[b])
            if ref_codes and rid in ref_codes and self.request_payload.get(rid) is None:
                self.request_payload[rid] = torch.as_tensor(ref_codes[rid]).to(torch.long).cpu().contiguous()
            payload = talker2code2wav_async_chunk(self, {}, req, is_finished=fin)
            if payload is not None:
                out.append((rid, payload))
        return out

    def send_step(self, requests: list[Any], audio_codes, finished: list[bool] | None = None, *, stage_id: int = 0,
                  ref_codes: dict | None = None) -> list[str]:
        """on_step + the SHM hop: every due payload goes to `connector.put` under the reference's chunk key
        `{external_req_id}_{stage_id}_{chunk_id}` (chunk_transfer_adapter.py:207-240); the per-request chunk counter
        advances only on a successful put; a finished request's state is dropped.  Returns the keys that were put."""
        keys = []
        fin_of = {}
        for b, req in enumerate(requests):
            fin_of[req.external_req_id] = bool(finished[b]) if finished is not None else bool(req.is_finished())
        for rid, payload in self.on_step(requests, audio_codes, finished, ref_codes):
            key = f"{rid}_{stage_id}_{self.put_req_chunk[rid]}"
            ok, _, _ = self.connector.put(from_stage=str(stage_id), to_stage=str(stage_id + 1), put_key=key, data=payload)
            if ok:
                self.put_req_chunk[rid] += 1
                keys.append(key)
        for rid, fin in fin_of.items():
            if fin:
                self.cleanup(rid)
        return keys

    def cleanup(self, request_id) -> None:
        self.code_prompt_token_ids.pop(request_id, None)
        self.request_payload.pop(request_id, None)
        self.put_req_chunk.pop(request_id, None)
        getattr(self, "_cached_ic", {}).pop(request_id, None)


class _Cfg:
    def __init__(self, chunk: int, left: int):
        self.config = {"extra": {"codec_chunk_frames": chunk, "codec_left_context_frames": left}}
