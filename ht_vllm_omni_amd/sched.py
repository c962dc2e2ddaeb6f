"""Minimal KV block pool with the allocation order the path depends on.

The reference delegates block allocation to vLLM's ``KVCacheManager`` / ``BlockPool`` (third
party, SURVEY Appendix A "Block allocation order"): free list = queue initialised in id order;
block 0 is popped first and kept as the permanent *null block*; allocation pops from the head;
a finished request's blocks are appended to the tail in reverse order; prefix caching is off for
every talker config (V/model_executor/stage_configs/qwen3_tts.yaml:17).  The scheduler reads the
ids via ``kv_cache_manager.get_block_ids`` (V/core/sched/omni_ar_scheduler.py:565).
"""
from __future__ import annotations

from collections import deque


class BlockPool:
    def __init__(self, num_blocks: int, block_size: int):
        if num_blocks < 2:
            raise ValueError("need at least the null block and one usable block")
        self.block_size = block_size
        self.free = deque(range(num_blocks))
        self.null_block = self.free.popleft()          # block 0, never handed out
        self.owned: dict[str, list[int]] = {}

    @property
    def num_free(self) -> int:
        return len(self.free)

    def blocks_needed(self, num_tokens: int) -> int:
        return (num_tokens + self.block_size - 1) // self.block_size

    def allocate(self, req_id: str, num_tokens_total: int) -> list[int]:
        """Grow req_id's block list to cover num_tokens_total tokens; returns the NEW block ids."""
        have = self.owned.setdefault(req_id, [])
        need = self.blocks_needed(num_tokens_total) - len(have)
        if need <= 0:
            return []
        if need > len(self.free):
            raise MemoryError(f"KV pool exhausted: need {need}, free {len(self.free)}")
        new = [self.free.popleft() for _ in range(need)]
        have.extend(new)
        return new

    def free_request(self, req_id: str) -> None:
        for b in reversed(self.owned.pop(req_id, [])):
            self.free.append(b)

    def block_ids(self, req_id: str) -> list[int]:
        return list(self.owned.get(req_id, []))


def slot_of(block_ids, pos: int, block_size: int) -> int:
    """slot = block_table[r][p // bs] * bs + p % bs (vLLM BlockTable.compute_slot_mapping)."""
    return block_ids[pos // block_size] * block_size + pos % block_size


def truncate_blocks(block_ids: list[int], seq_len: int, block_size: int) -> list[int]:
    """Blocks shipped with a finished request: ceil(seq_len / block_size)
    (V/core/sched/omni_ar_scheduler.py:581-588)."""
    return block_ids[: (seq_len + block_size - 1) // block_size]
