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


def get_input_positions_tensor(input_tokens, *, context_len: int = 0, seq_len: int | None = None) -> tuple[torch.Tensor, int]:
    """Text-only arm of mrope.py:64-109 (-> mrope.py:196-203 / 292-303): positions [3, seq_len - context_len] int64 and
    the position delta (max position + 1 - number of tokens = 0 for text)."""
    n = len(input_tokens)
    pos = torch.arange(n, dtype=torch.int64).view(1, -1).expand(3, -1)
    delta = int(pos.max().item() + 1 - n) if n else 0
    return pos[:, context_len:seq_len].contiguous(), delta


def get_input_positions(input_tokens, *, context_len: int = 0, seq_len: int | None = None) -> tuple[list[list[int]], int]:
    pos, delta = get_input_positions_tensor(input_tokens, context_len=context_len, seq_len=seq_len)
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
