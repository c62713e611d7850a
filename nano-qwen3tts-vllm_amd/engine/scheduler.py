"""Continuous-batching scheduler (reference: P/engine/scheduler.py:8-79):
prefill everything that is waiting and fits, otherwise decode everything running."""
from __future__ import annotations

from collections import deque

from .block_manager import BlockManager
from .sequence import Sequence, SequenceStatus


class Scheduler:
    def __init__(self, config):
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.eos = config.eos
        self.block_manager = BlockManager(config.num_kvcache_blocks, config.kvcache_block_size)
        self.waiting: deque[Sequence] = deque()
        self.running: deque[Sequence] = deque()

    def is_finished(self):
        return not self.waiting and not self.running

    def add(self, seq: Sequence):
        self.waiting.append(seq)

    def _admit_waiting(self) -> list[Sequence]:
        batch, tokens = [], 0
        while self.waiting and len(batch) < self.max_num_seqs:
            seq = self.waiting[0]
            if tokens + len(seq) > self.max_num_batched_tokens or not self.block_manager.can_allocate(seq):
                break
            self.block_manager.allocate(seq)
            tokens += len(seq) - seq.num_cached_tokens
            seq.status = SequenceStatus.RUNNING
            self.waiting.popleft()
            self.running.append(seq)
            batch.append(seq)
        return batch

    def _reserve_slot(self, seq: Sequence) -> bool:
        """Make room for one more token of `seq`, preempting from the tail if needed."""
        while not self.block_manager.can_append(seq):
            if self.running:
                self.preempt(self.running.pop())
            else:
                self.preempt(seq)
                return False
        self.block_manager.may_append(seq)
        return True

    def schedule(self) -> tuple[list[Sequence], bool]:
        batch = self._admit_waiting()
        if batch:
            return batch, True
        while self.running and len(batch) < self.max_num_seqs:
            seq = self.running.popleft()
            if self._reserve_slot(seq):
                batch.append(seq)
        assert batch
        self.running.extendleft(reversed(batch))
        return batch, False

    def preempt(self, seq: Sequence):
        seq.status = SequenceStatus.WAITING
        self.block_manager.deallocate(seq)
        self.waiting.appendleft(seq)

    def postprocess(self, seqs: list[Sequence], token_ids: list[int]):
        for seq, tok in zip(seqs, token_ids):
            seq.append_token(tok, last_hidden_state=None)
            if (not seq.ignore_eos and tok == self.eos) or seq.num_completion_tokens >= seq.max_tokens:
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                self.running.remove(seq)
