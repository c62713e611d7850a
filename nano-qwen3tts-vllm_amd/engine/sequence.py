"""Per-request state (reference: P/engine/sequence.py:15-90).

Sequences are fed embeddings, never token ids: `num_tokens` starts at the prompt's
embedding count and `token_ids` holds only generated ids, so `num_prompt_tokens`
is 0 and `num_completion_tokens == num_tokens` as in the reference."""
from __future__ import annotations

from copy import copy
from enum import Enum, auto
from itertools import count
from typing import Optional

import torch

from ..sampling_params import SamplingParams


class SequenceStatus(Enum):
    WAITING = auto()
    RUNNING = auto()
    FINISHED = auto()


class Sequence:
    block_size = 256
    counter = count()

    def __init__(self, token_ids: Optional[list[int]], sampling_params: SamplingParams | None = None,
                 input_embeds: Optional[torch.Tensor] = None, request_id: Optional[str] = None):
        sp = sampling_params if sampling_params is not None else SamplingParams()
        self.seq_id = next(Sequence.counter)
        self.status = SequenceStatus.WAITING
        self.input_embeds = input_embeds
        self.request_id = request_id
        self.decode_input_embeds: Optional[torch.Tensor] = None
        self.token_ids = copy(token_ids) if token_ids else []
        self.last_token = self.token_ids[-1] if self.token_ids else None
        self.num_tokens = len(self.token_ids) if self.token_ids else input_embeds.shape[1]
        self.num_prompt_tokens = len(self.token_ids)
        self.num_cached_tokens = 0
        self.block_table: list[int] = []
        self.temperature = sp.temperature
        self.max_tokens = sp.max_tokens
        self.ignore_eos = sp.ignore_eos
        self.do_sample = getattr(sp, "do_sample", True)
        self.top_k = getattr(sp, "top_k", 50)
        self.generation_steps = 0
        self.last_hidden_state = None

    def __len__(self):
        return self.num_tokens

    def __getitem__(self, key):
        return self.token_ids[key]

    @property
    def is_finished(self):
        return self.status == SequenceStatus.FINISHED

    @property
    def num_completion_tokens(self):
        return self.num_tokens - self.num_prompt_tokens

    @property
    def prompt_token_ids(self):
        return self.token_ids[:self.num_prompt_tokens]

    @property
    def completion_token_ids(self):
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_cached_blocks(self):
        return self.num_cached_tokens // self.block_size

    @property
    def num_blocks(self):
        return (self.num_tokens + self.block_size - 1) // self.block_size

    @property
    def last_block_num_tokens(self):
        return self.num_tokens - (self.num_blocks - 1) * self.block_size

    def append_token(self, token_id: int, last_hidden_state=None):
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1
        self.last_hidden_state = last_hidden_state
