"""Prompt-embedding builder of the Qwen3-Omni talker: thinker outputs -> the talker's prefill
embeddings, its queue of per-step text embeddings, and the decode-side text-step feeders.

Mirrors `Qwen3OmniMoeForConditionalGeneration.talker_preprocess_prefill / _thinker_to_talker_prefill / _get_talker_user_parts /
_get_talker_assistant_parts / _get_tts_embed / talker_preprocess_decode / _thinker_decode_to_talker_decode`
(/root/reference/vllm_omni/model_executor/models/qwen3_omni/qwen3_omni.py:650-1060).  MI355X shape of it: the segment walk is
integer work on the host ids; all rows that need the same projection (text rows of every user segment, the assistant segment
and the three TTS marker embeddings; multimodal rows) are gathered into ONE batch per projection and go through
`omni_resize_mlp` (skinny GEMM + SiLU + skinny GEMM, 64 rows per launch group); the result rows are scattered into the prompt
on the device.  The reference projects segment by segment (one small GEMM chain each) -- a row's result does not depend on its
batch mates, so the outputs are the same.  No CPU fallback: the projections need the HIP library.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

import torch

from . import ops

BF16 = torch.bfloat16


@dataclass(frozen=True)
class OmniPromptIds:
    """Token ids the builder keys on (HF Qwen3OmniMoeConfig / talker_config / thinker_config names in comments; the values are
    the class defaults of transformers 5.x -- a checkpoint's config.json overrides them: from_hf_config)."""
    im_start: int = 151644          # config.im_start_token_id
    system: int = 8948              # config.system_token_id
    user: int = 872                 # config.user_token_id
    assistant: int = 77091          # config.assistant_token_id
    tts_pad_token: int = 151671     # config.tts_pad_token_id
    audio: int = 151646             # thinker_config.audio_token_id
    image: int = 151655             # thinker_config.image_token_id
    video: int = 151656             # thinker_config.video_token_id
    codec_nothink: int = 4203       # talker_config.codec_nothink_id
    codec_think_bos: int = 4204
    codec_think_eos: int = 4205
    codec_pad: int = 4196
    codec_bos: int = 4197
    speaker_ids: dict = field(default_factory=dict)   # talker_config.speaker_id: voice name -> codec token id
    default_speaker: str = "ethan"

    @classmethod
    def from_hf_config(cls, cfg: Any, default_speaker: str = "ethan") -> "OmniPromptIds":
        tc, th = cfg.talker_config, cfg.thinker_config
        return cls(im_start=cfg.im_start_token_id, system=cfg.system_token_id, user=cfg.user_token_id,
                   assistant=cfg.assistant_token_id, tts_pad_token=cfg.tts_pad_token_id, audio=th.audio_token_id,
                   image=th.image_token_id, video=th.video_token_id, codec_nothink=tc.codec_nothink_id,
                   codec_think_bos=tc.codec_think_bos_id, codec_think_eos=tc.codec_think_eos_id, codec_pad=tc.codec_pad_id,
                   codec_bos=tc.codec_bos_id, speaker_ids=dict(getattr(tc, "speaker_id", None) or {}), default_speaker=default_speaker)

    @classmethod
    def from_dict(cls, d: dict, **kw) -> "OmniPromptIds":
        return cls(**{k: v for k, v in d.items() if k in cls.__dataclass_fields__}, **kw)

    def speaker_token(self, voice: Any) -> int:
        """talker_preprocess_prefill's speaker parsing (qwen3_omni.py:682-691) + _get_text_spk_token_id (572-576)."""
        if voice is not None and isinstance(voice, (list, tuple)) and len(voice) > 0:
            voice = voice[0]
        if not isinstance(voice, str) or not voice.strip():
            voice = self.default_speaker
        else:
            voice = str(voice).lower().strip()
        if voice not in self.speaker_ids:
            voice = self.default_speaker
        if voice not in self.speaker_ids:
            raise ValueError(f"no speaker token id for voice {voice!r} (talker_config.speaker_id has {sorted(self.speaker_ids)})")
        return int(self.speaker_ids[voice])


@dataclass
class OmniPrompt:
    input_ids: torch.Tensor            # int64 [P] (CPU)
    embeds: torch.Tensor               # bf16 [P, H] (device)
    trailing_text_hidden: torch.Tensor  # bf16 [n, H] (device): the queue of per-decode-step text embeddings
    tts_bos: torch.Tensor              # bf16 [1, H] projected
    tts_eos: torch.Tensor
    tts_pad: torch.Tensor


def _last_row(x: Any, width: int, dev) -> torch.Tensor:
    """_get_tts_embed._ensure_1x1 + its zero fallback (qwen3_omni.py:654-670) -> [1, width] bf16 on the device."""
    if not isinstance(x, torch.Tensor) or x.numel() == 0:
        return torch.zeros(1, width, dtype=BF16, device=dev)
    if x.ndim == 3:
        x = x[0, -1:, :]
    elif x.ndim == 2:
        x = x[-1:]
    else:
        x = x.reshape(1, -1)
    return x.to(device=dev, dtype=BF16)


class OmniTalkerPromptBuilder:
    def __init__(self, weights: dict, ids: OmniPromptIds, device: str | torch.device = "cuda"):
        """weights: {"text": {fc1_w, fc1_b, fc2_w, fc2_b}, "hidden": {...}, "codec_embed": [V, H]} -- the talker's
        text_projection / hidden_projection (qwen3_omni_moe_talker.py:121-122) and codec embedding table."""
        self.dev = torch.device(device)
        up = lambda t: t.to(device=self.dev, dtype=BF16).contiguous()
        self.w_text = {k: up(v) for k, v in weights["text"].items()}
        self.w_hidden = {k: up(v) for k, v in weights["hidden"].items()}
        self.codec_embed = up(weights["codec_embed"])
        self.ids = ids
        self.thinker_hidden = int(self.w_text["fc1_w"].shape[1])
        self.hidden = int(self.codec_embed.shape[1])

    def project_text(self, x: torch.Tensor) -> torch.Tensor:
        return ops.resize_mlp(x.to(device=self.dev, dtype=BF16).contiguous(), self.w_text)

    def project_hidden(self, x: torch.Tensor) -> torch.Tensor:
        return ops.resize_mlp(x.to(device=self.dev, dtype=BF16).contiguous(), self.w_hidden)

    # ---- _thinker_to_talker_prefill ---------------------------------------------------------------------------------
    def prefill(self, thinker_embed: torch.Tensor, thinker_hidden: torch.Tensor, input_ids, result_ids, speaker_id: int,
                tts_bos=None, tts_eos=None, tts_pad=None) -> OmniPrompt:
        I, dev, H = self.ids, self.dev, self.hidden
        in_ids = torch.as_tensor(input_ids, dtype=torch.long).reshape(-1).cpu()
        res_ids = torch.as_tensor(result_ids, dtype=torch.long).reshape(-1).cpu()
        starts = torch.nonzero(in_ids == I.im_start).reshape(-1).tolist()
        if len(starts) < 2:      # the reference's torch.cat over a 0-d / empty index tensor raises here (qwen3_omni.py:851-857,903)
            raise ValueError("OmniTalkerPromptBuilder: need at least two <|im_start|> segments in thinker_input_ids")
        bounds = starts + [int(res_ids.shape[0])]
        mm = (res_ids == I.audio) | (res_ids == I.image) | (res_ids == I.video)
        # ---- plan (host, integers): which source rows go through which projection, and where each result row lands
        text_src: list[int] = []        # rows of thinker_embed -> text_projection
        text_dst: list[int] = []
        mm_src: list[int] = []          # rows of thinker_hidden -> hidden_projection
        mm_dst: list[int] = []
        out_ids: list[torch.Tensor] = []
        P = 0
        asst = None                     # (s, e, dst0) of the last assistant segment
        for i in range(len(bounds) - 1):
            s, e = bounds[i], bounds[i + 1]
            role = int(in_ids[s + 1])
            if role == I.system:
                continue
            if role == I.user:
                for t in range(s, e):
                    (mm_src if mm[t] else text_src).append(t)
                    (mm_dst if mm[t] else text_dst).append(P + t - s)
                out_ids.append(res_ids[s:e])
                P += e - s
            elif role == I.assistant and i == len(bounds) - 2:
                if e - s < 3:
                    raise ValueError("OmniTalkerPromptBuilder: assistant segment shorter than <|im_start|>assistant\\n")
                asst = (s, e, P)
                out_ids.append(torch.full((9,), I.tts_pad_token, dtype=torch.long))
                P += 9
            elif role == I.assistant:
                continue
            else:
                raise AssertionError("Expect role id after <|im_start|> (assistant, user, system)")
        if P == 0:
            raise ValueError("OmniTalkerPromptBuilder: no user / assistant segment")
        te = thinker_embed.to(device=dev, dtype=BF16)
        th = thinker_hidden.to(device=dev, dtype=BF16)
        # ---- one batch per projection: [user text rows | assistant rows | tts bos, eos, pad]
        n_user_text = len(text_src)
        a_rows = list(range(asst[0], asst[1])) if asst else []
        idx = torch.tensor(text_src + a_rows, dtype=torch.long, device=dev)
        marks = torch.cat([_last_row(t, self.thinker_hidden, dev) for t in (tts_bos, tts_eos, tts_pad)], 0)
        proj = self.project_text(torch.cat([te.index_select(0, idx), marks], 0))
        bos, eos, pad = proj[-3:-2], proj[-2:-1], proj[-1:]
        embeds = torch.zeros(P, H, dtype=BF16, device=dev)
        if n_user_text:
            embeds.index_copy_(0, torch.tensor(text_dst, dtype=torch.long, device=dev), proj[:n_user_text])
        if mm_src:
            pm = self.project_hidden(th.index_select(0, torch.tensor(mm_src, dtype=torch.long, device=dev)))
            embeds.index_copy_(0, torch.tensor(mm_dst, dtype=torch.long, device=dev), pm)
        trailing = eos
        if asst:
            ah = proj[n_user_text:n_user_text + len(a_rows)]
            first = ah[3:4] if ah.shape[0] > 3 else torch.zeros(1, H, dtype=BF16, device=dev)
            # [3 tokens] + [4 pad] + [1 BOS] + [1 first text] over [3 zero rows] + [6 codec special tokens] (qwen3_omni.py:1000-1047)
            text = torch.cat([ah[:3], pad.expand(4, -1), bos, first], 0)
            codec_ids = torch.tensor([I.codec_nothink, I.codec_think_bos, I.codec_think_eos, int(speaker_id), I.codec_pad,
                                      I.codec_bos], dtype=torch.long, device=dev)
            codec = torch.cat([torch.zeros(3, H, dtype=BF16, device=dev), self.codec_embed.index_select(0, codec_ids)], 0)
            embeds[asst[2]:asst[2] + 9] = text + codec
            if ah.shape[0] > 4:
                trailing = torch.cat([ah[4:], eos], 0)
        return OmniPrompt(torch.cat(out_ids, 0), embeds, trailing.contiguous(), bos, eos, pad)

    # ---- talker_preprocess_prefill (qwen3_omni.py:678-809): the request's additional_information -> prompt + update_dict
    def from_info(self, info: dict) -> tuple[OmniPrompt, dict]:
        if info.get("thinker_prefill_embeddings") is None or info.get("thinker_hidden_states") is None:
            raise ValueError("additional_information must include 'thinker_prefill_embeddings' and 'thinker_hidden_states' "
                             "for talker prefill.")
        te, th = info["thinker_prefill_embeddings"], info["thinker_hidden_states"]
        seqs, chat = info.get("thinker_sequences"), info.get("thinker_input_ids")
        if chat is None:      # the reference's dummy-id fallback cannot find a segment either; fail with a clear message
            raise ValueError("additional_information must include 'thinker_input_ids' (chatml prompt ids)")
        if seqs is None:
            seqs = chat
        p = self.prefill(te.reshape(-1, te.shape[-1]), th.reshape(-1, th.shape[-1]), chat, seqs,
                         self.ids.speaker_token(info.get("speaker")), info.get("tts_bos_embed"), info.get("tts_eos_embed"),
                         info.get("tts_pad_embed"))
        upd: dict[str, Any] = {"prefill_consumed_text_tokens": 1, "tts_pad_embed_projected": p.tts_pad.reshape(1, 1, -1)}
        if p.trailing_text_hidden.shape[0] > 0:
            upd["trailing_text_hidden"] = p.trailing_text_hidden
        fresh = info.get("thinker_decode_embeddings")           # _talker_cache_thinker_decode_embeds (811-832)
        if fresh is not None:
            cached = info.get("cached_thinker_decode_embeddings")
            fresh = fresh.to(device=self.dev, dtype=BF16)
            upd["cached_thinker_decode_embeddings"] = fresh if cached is None else torch.cat(
                [cached.to(device=self.dev, dtype=BF16), fresh], 0)
        upd["thinker_decode_embeddings"] = None
        return p, upd

    # ---- decode side (qwen3_omni.py:907-960) ---------------------------------------------------------------------------
    @staticmethod
    def pop_text_step(tail: torch.Tensor | None, tts_pad: torch.Tensor):
        """Non-streaming: the next queued text embedding, tts_pad once the queue has run out -> (text_step [1, H], new tail)."""
        if isinstance(tail, torch.Tensor) and tail.numel() > 0:
            return tail[0:1], (tail[1:] if tail.shape[0] > 1 else tts_pad.reshape(1, -1))
        return tts_pad.reshape(1, -1), tail

    def streaming_text_step(self, state: dict, n_thinker_output_ids: int, tts_eos: torch.Tensor, tts_pad: torch.Tensor):
        """Streaming (async_chunk): index num_processed_tokens into the cached thinker decode embeddings, project that row;
        tts_eos once when the thinker's output is exhausted, tts_pad after.  state keys: num_processed_tokens, finished_flag,
        cached [n, Ht] | None, fresh [m, Ht] | None (mutated like the reference's update_dict)."""
        start = int(state.get("num_processed_tokens", 0))
        if start >= n_thinker_output_ids - 1:
            if state.get("finished_flag"):
                return tts_pad
            state["finished_flag"] = True
            return tts_eos
        cached, fresh = state.get("cached"), state.get("fresh")
        if cached is not None and start < cached.shape[0]:
            x = cached[start].reshape(1, -1)
            squeeze = True
            if fresh is not None:
                state["cached"] = torch.cat([cached.to(self.dev), fresh.to(self.dev)], 0)
        else:
            if fresh is None:
                raise ValueError("streaming_text_step: no thinker decode embedding for this step")
            x, squeeze = fresh, False
        state["fresh"] = None
        y = self.project_text(x)
        return y.reshape(-1) if squeeze else y
