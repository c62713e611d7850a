"""Host-side record of one request in an engine (the role of P/engine/sequence.py:15-90).

Requests are fed embeddings, never token ids, so the record is built around what the HIP runners
and the schedulers actually read: how many positions the prompt holds, the ids generated so far,
the KV blocks, the sampling options and the embedding waiting to be fed next.  Lengths are
derived (`len(seq) = prompt positions + generated ids`) rather than counted separately; the
reference's bookkeeping names the schedulers and engines use (`num_completion_tokens`,
`last_block_num_tokens`, `completion_token_ids`, ...) are views of that state.  As in the
reference the whole length counts as "completion" (its `num_prompt_tokens` is the number of
prompt *ids*, which is zero for an embedding prompt), which is what makes `max_tokens = 17` end
a predictor burst after 15 generated ids on a 2-position prompt.
"""
from __future__ import annotations

import enum
import itertools
from typing import Optional

import torch

from ..sampling_params import SamplingParams

KV_BLOCK = 256                       # tokens per KV block (P/engine/sequence.py:16, P/config.py:52)


class SequenceStatus(enum.Enum):
    WAITING = "waiting"
    RUNNING = "running"
    FINISHED = "finished"


_ids = itertools.count()


class Sequence:
    __slots__ = ("seq_id", "request_id", "status", "sampling", "input_embeds", "decode_input_embeds", "prompt_positions",
                 "prompt_ids", "generated", "block_table", "num_cached_tokens", "last_hidden_state", "generation_steps")
    block_size = KV_BLOCK

    def __init__(self, token_ids: Optional[list[int]], sampling_params: SamplingParams | None = None,
                 input_embeds: Optional[torch.Tensor] = None, request_id: Optional[str] = None):
        """token_ids: an id prompt (unused by the TTS engines, kept for the reference's call form) or empty;
        input_embeds [1, L, H]: the embedding prompt."""
        self.seq_id = next(_ids)
        self.request_id = request_id
        self.status = SequenceStatus.WAITING
        self.sampling = sampling_params if sampling_params is not None else SamplingParams()
        self.input_embeds = input_embeds
        self.decode_input_embeds: Optional[torch.Tensor] = None      # the next decode step's input, fed by add_request
        self.prompt_ids = list(token_ids) if token_ids else []
        self.prompt_positions = len(self.prompt_ids) if self.prompt_ids else int(input_embeds.shape[1])
        self.generated: list[int] = []
        self.block_table: list[int] = []
        self.num_cached_tokens = 0
        self.last_hidden_state = None
        self.generation_steps = 0

    # ---- lengths
    def __len__(self) -> int:
        return self.prompt_positions + len(self.generated)

    num_tokens = property(__len__)

    @property
    def num_prompt_tokens(self) -> int:
        return len(self.prompt_ids)

    @property
    def num_completion_tokens(self) -> int:
        return len(self) - len(self.prompt_ids)

    @property
    def num_blocks(self) -> int:
        return -(-len(self) // KV_BLOCK)

    @property
    def last_block_num_tokens(self) -> int:
        return len(self) - (self.num_blocks - 1) * KV_BLOCK

    # ---- ids
    @property
    def token_ids(self) -> list[int]:
        return self.prompt_ids + self.generated

    @property
    def completion_token_ids(self) -> list[int]:
        return list(self.generated)

    @property
    def last_token(self) -> Optional[int]:
        return self.generated[-1] if self.generated else (self.prompt_ids[-1] if self.prompt_ids else None)

    def __getitem__(self, key):
        return self.token_ids[key]

    def append_token(self, token_id: int, last_hidden_state=None) -> None:
        self.generated.append(int(token_id))
        self.last_hidden_state = last_hidden_state

    # ---- state
    @property
    def is_finished(self) -> bool:
        return self.status is SequenceStatus.FINISHED

    # ---- sampling options, read per step by the runners
    @property
    def temperature(self) -> float:
        return self.sampling.temperature

    @property
    def max_tokens(self) -> int:
        return self.sampling.max_tokens

    @property
    def ignore_eos(self) -> bool:
        return self.sampling.ignore_eos

    @property
    def do_sample(self) -> bool:
        return getattr(self.sampling, "do_sample", True)

    @property
    def top_k(self) -> int:
        return getattr(self.sampling, "top_k", 50)
