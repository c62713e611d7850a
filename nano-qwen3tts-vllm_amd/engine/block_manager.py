"""Free-list manager for 256-token KV blocks (reference: P/engine/block_manager.py:29-115).

The reference also keeps an xxhash prefix-cache map, but sequences here carry
embeddings and an empty `token_ids`, so no block is ever full at allocation time
and that map never hits (block_manager.py:67-70).  Only the free-list order and
the append rule are observable, and those are reproduced: blocks are taken from
the head of the free deque and returned to its tail in reverse table order.  The
reference's assertion failure for prompts whose length is an exact multiple of
256 (block_manager.py:102-103) is not reproduced."""
from __future__ import annotations

from collections import deque

from .sequence import Sequence


class BlockManager:
    def __init__(self, num_blocks: int, block_size: int):
        self.block_size = block_size
        self.num_blocks = num_blocks
        self.free_block_ids: deque[int] = deque(range(num_blocks))
        self.used_block_ids: set[int] = set()

    def _take(self) -> int:
        b = self.free_block_ids.popleft()
        self.used_block_ids.add(b)
        return b

    def can_allocate(self, seq: Sequence) -> bool:
        return len(self.free_block_ids) >= seq.num_blocks

    def allocate(self, seq: Sequence):
        assert not seq.block_table
        for _ in range(seq.num_blocks):
            seq.block_table.append(self._take())

    def deallocate(self, seq: Sequence):
        for b in reversed(seq.block_table):
            self.used_block_ids.discard(b)
            self.free_block_ids.append(b)
        seq.num_cached_tokens = 0
        seq.block_table.clear()

    def can_append(self, seq: Sequence) -> bool:
        return len(self.free_block_ids) >= (len(seq) % self.block_size == 1)

    def may_append(self, seq: Sequence):
        if len(seq) % self.block_size == 1 and len(seq.block_table) < seq.num_blocks:
            seq.block_table.append(self._take())
