"""Integer position ids of the talker stage.

The reference asks `OmniMRotaryEmbedding.get_input_positions_tensor`
(V/model_executor/layers/rotary_embedding/mrope.py:64-109; runner hooks V/worker/gpu_model_runner.py:121-244) for
[3, T] M-RoPE position ids at prefill.  The talker's prompt carries no image / video / audio placeholder tokens
(its multimodal content arrives as embeddings from the thinker), so every branch of that function reduces to its
text-only arm: three identical aranges and `mrope_position_delta = 0`.  With identical rows the three M-RoPE sections
rotate by the same angle, i.e. M-RoPE == plain neox RoPE at that position: the kernels take ONE position per token
(`positions` int32 [T]) and this module is where the [3, T] form the vLLM side expects is produced and checked.
"""
from __future__ import annotations

import torch


class MRopeTokenIds:
    """The token ids and grid constants `_omni_get_input_positions_tensor` reads off `hf_config.thinker_config` (mrope.py:340-357)."""

    def __init__(self, thinker_config):
        g = lambda *names: next(getattr(thinker_config, n) for n in names if hasattr(thinker_config, n))
        self.audio, self.image, self.video = g("audio_token_index", "audio_token_id"), g("image_token_index", "image_token_id"), \
            g("video_token_index", "video_token_id")
        self.audio_start, self.audio_end = thinker_config.audio_start_token_id, thinker_config.audio_end_token_id
        self.vision_start, self.vision_end = thinker_config.vision_start_token_id, thinker_config.vision_end_token_id
        self.seconds_per_chunk = thinker_config.seconds_per_chunk
        self.merge = thinker_config.vision_config.spatial_merge_size
        self.tokens_per_second = getattr(thinker_config.vision_config, "tokens_per_second", 25)


def _audio_tokens(feature_len: int) -> int:
    """Placeholder tokens of an audio clip of `feature_len` mel frames (two stride-2 stages; mrope.py:390, 432)."""
    return ((int(feature_len) - 1) // 2 + 1 - 2) // 2 + 1


def _vision_block(start: int, t_index: torch.Tensor, grid_h: int, grid_w: int, merge: int) -> torch.Tensor:
    """[3, len(t_index) * gh * gw] ids of a vision item: row 0 the temporal index, rows 1 / 2 the merged grid's (h, w)
    (`_get_llm_pos_ids_for_vision`, mrope.py:480-500), all offset by `start`."""
    gh, gw = int(grid_h) // merge, int(grid_w) // merge
    tt = t_index.to(torch.int64).view(-1, 1, 1).expand(-1, gh, gw)
    hh = torch.arange(gh).view(1, -1, 1).expand(len(t_index), -1, gw)
    ww = torch.arange(gw).view(1, 1, -1).expand(len(t_index), gh, -1)
    return torch.stack([tt.reshape(-1), hh.reshape(-1), ww.reshape(-1)]) + start


def omni_input_positions(input_tokens, ids: MRopeTokenIds, image_grid_thw=(), video_grid_thw=(), second_per_grid_ts=(),
                         audio_feature_lengths=None, use_audio_in_video: bool = False) -> tuple[torch.Tensor, int]:
    """The Omni arm of the reference (`_omni_get_input_positions_tensor`, mrope.py:311-478), restated as a walk over ITEMS: a text
    token takes one id on all three rows; an audio clip a run of consecutive ids; an image / video a (t, h, w) block; with
    `use_audio_in_video` a video's frames and its audio are interleaved in chunks of `seconds_per_chunk`.  Every item starts at
    (largest id of the PREVIOUS piece) + 1 -- the previous piece, not the running maximum (mrope.py:372) -- and the delta is
    (largest id overall) + 1 - len(input_tokens).  Returns ([3, len(input_tokens)] int64, delta).  Pinned to outputs of the
    reference's function (tests/golden/mrope_positions.json)."""
    toks = list(input_tokens)
    img = torch.as_tensor(image_grid_thw, dtype=torch.int64).reshape(-1, 3)
    vid = torch.as_tensor(video_grid_thw, dtype=torch.int64).reshape(-1, 3)
    spg = list(second_per_grid_ts) if len(second_per_grid_ts) else [1] * vid.shape[0]
    pieces: list[torch.Tensor] = []
    a_i = i_i = v_i = 0
    pos = 0
    prev_max = -1                                  # largest id of the last piece
    tps, merge = ids.tokens_per_second, ids.merge
    while pos < len(toks):
        tok = toks[pos]
        start = prev_max + 1
        if tok == ids.audio:
            n = _audio_tokens(audio_feature_lengths[a_i])
            block = (torch.arange(n) + start).expand(3, -1)
            a_i += 1
        elif tok == ids.image:
            t, h, w = img[i_i].tolist()
            block = _vision_block(start, (torch.arange(t) * 1 * tps).long(), h, w, merge)
            i_i += 1
        elif tok == ids.video and not use_audio_in_video:
            t, h, w = vid[v_i].tolist()
            block = _vision_block(start, (torch.arange(t) * spg[v_i] * tps).long(), h, w, merge)
            v_i += 1
        elif tok == ids.video:
            # frames and audio of one clip, chunk by chunk: frames of [c * chunk, (c + 1) * chunk) temporal ids, then up to
            # `chunk` audio ids continuing the audio run; audio left over after the last frame chunk follows the largest id so far
            t, h, w = vid[v_i].tolist()
            chunk = int(tps * ids.seconds_per_chunk)
            t_index = (torch.arange(t) * spg[v_i] * tps).long()
            audio_left = _audio_tokens(audio_feature_lengths[a_i])
            cols: list[torch.Tensor] = []
            audio_next = start
            for c in range(int(t_index.max()) // chunk + 1):
                cols.append(_vision_block(start, t_index[(t_index // chunk) == c], h, w, merge))
                n = min(chunk, audio_left)
                if n > 0:
                    cols.append((torch.arange(n) + audio_next).expand(3, -1))
                    audio_next += n
                    audio_left -= n
                else:
                    audio_next = start             # an empty audio chunk resets the run (mrope.py:451-460)
            block = torch.cat(cols, dim=1)
            if audio_left > 0:
                block = torch.cat([block, (torch.arange(audio_left) + int(block[:, -1].max()) + 1).expand(3, -1)], dim=1)
            a_i += 1
            v_i += 1
        else:
            if use_audio_in_video and pos > 0 and ((tok == ids.vision_end and toks[pos - 1] == ids.audio_end) or
                                                   (tok == ids.audio_start and toks[pos - 1] == ids.vision_start)):
                start -= 1                         # the paired bos / eos of an audio-in-video clip share an id (mrope.py:375-381)
            block = torch.full((3, 1), start, dtype=torch.int64)
        pieces.append(block)
        # the reference keeps audio-in-video ids as single-token pieces: "previous piece" is then the clip's LAST token
        prev_max = int(block[:, -1].max()) if (tok == ids.video and use_audio_in_video) else int(block.max())
        pos += block.shape[1]
    out = torch.cat(pieces, dim=1) if pieces else torch.zeros(3, 0, dtype=torch.int64)
    delta = int(out.max()) + 1 - len(toks) if out.numel() else 0
    return out, delta


def get_input_positions_tensor(input_tokens, hf_config=None, image_grid_thw=None, video_grid_thw=None, second_per_grid_ts=None, *,
                               context_len: int = 0, seq_len: int | None = None, audio_feature_lengths=None,
                               use_audio_in_video: bool = False) -> tuple[torch.Tensor, int]:
    """`OmniMRotaryEmbedding.get_input_positions_tensor` (mrope.py:64-109): positions [3, seq_len - context_len] int64 and the
    position delta.  Without an `hf_config` (or without multimodal items) this is the text-only arm: three identical aranges,
    delta 0 (mrope.py:196-203 / 292-303).  With `hf_config.thinker_config` it is the Omni arm (`omni_input_positions`).  The
    Qwen2-VL / GLM-4V arms of the reference (mrope.py:112-309) belong to models that are not talker stages: refused."""
    if hf_config is not None and hasattr(hf_config, "thinker_config"):
        pos, delta = omni_input_positions(input_tokens, MRopeTokenIds(hf_config.thinker_config), image_grid_thw or (),
                                          video_grid_thw or (), second_per_grid_ts or (), audio_feature_lengths, use_audio_in_video)
        return pos[:, context_len:seq_len].contiguous(), delta
    if hf_config is not None or image_grid_thw or video_grid_thw or audio_feature_lengths is not None:
        raise NotImplementedError("only the Omni (thinker_config) arm and the text-only arm of get_input_positions_tensor are "
                                  "restated: the Qwen2-VL / GLM-4V arms are not reached from a talker stage")
    n = len(input_tokens)
    pos = torch.arange(n, dtype=torch.int64).view(1, -1).expand(3, -1)
    delta = int(pos.max().item() + 1 - n) if n else 0
    return pos[:, context_len:seq_len].contiguous(), delta


def get_input_positions(input_tokens, hf_config=None, image_grid_thw=None, video_grid_thw=None, second_per_grid_ts=None, *,
                        context_len: int = 0, seq_len: int | None = None, audio_feature_lengths=None,
                        use_audio_in_video: bool = False) -> tuple[list[list[int]], int]:
    pos, delta = get_input_positions_tensor(input_tokens, hf_config, image_grid_thw, video_grid_thw, second_per_grid_ts,
                                            context_len=context_len, seq_len=seq_len, audio_feature_lengths=audio_feature_lengths,
                                            use_audio_in_video=use_audio_in_video)
    return pos.tolist(), delta


def get_next_input_positions(mrope_position_delta: int, context_len: int, seq_len: int) -> list[list[int]]:
    """Decode-time positions (vLLM MRotaryEmbedding.get_next_input_positions): delta + [context_len, seq_len) x 3."""
    r = list(range(context_len + mrope_position_delta, seq_len + mrope_position_delta))
    return [r[:], r[:], r[:]]


def collapse_mrope_positions(positions: torch.Tensor) -> torch.Tensor:
    """[3, T] (or [T]) -> the single int32 [T] row the kernels consume; refuses rows that differ (a real multimodal
    M-RoPE prompt is outside the talker path and must fail loudly, not rotate by the wrong angle)."""
    if positions.ndim == 1:
        return positions.to(torch.int32)
    if positions.ndim != 2 or positions.shape[0] != 3:
        raise ValueError(f"expected [T] or [3, T] positions, got {tuple(positions.shape)}")
    if not (torch.equal(positions[0], positions[1]) and torch.equal(positions[0], positions[2])):
        raise ValueError("M-RoPE rows differ: multimodal position ids are not supported on the talker path")
    return positions[0].to(torch.int32)
