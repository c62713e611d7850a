"""Talker engine (reference: P/engine/llm_engine/talker_llm_engine.py:11-134)."""
from __future__ import annotations

import torch

from ...sampling_params import SamplingParams
from ..model_runner.talker_mode_runner import TalkerModeModelRunner
from ..scheduler import Scheduler
from ..sequence import Sequence, SequenceStatus
from .base import LLMEngine


class TalkerScheduler(Scheduler):
    """Prefill first; otherwise decode only the sequences whose next input embedding has
    been fed (the API loop feeds one per frame), reference :16-59."""

    def __init__(self, config):
        super().__init__(config)
        self.request_id_to_seq: dict[str, Sequence] = {}

    def schedule(self):
        batch = self._admit_waiting()
        if batch:
            return batch, True
        for _ in range(len(self.running)):
            if not self.running or len(batch) >= self.max_num_seqs:
                break
            seq = self.running.popleft()
            if len(seq) > 0 and seq.decode_input_embeds is None:
                self.running.append(seq)          # not fed yet: keep it, look at the next one
                continue
            if self._reserve_slot(seq):
                batch.append(seq)
        if not batch:
            return [], False
        self.running.extendleft(reversed(batch))
        return batch, False

    def clear_request(self, request_id: str):
        seq = self.request_id_to_seq.pop(request_id, None)
        if seq is None:
            return
        self.block_manager.deallocate(seq)
        if seq in self.running:
            self.running.remove(seq)
        if seq in self.waiting:
            self.waiting.remove(seq)

    def postprocess(self, seqs, token_ids, hidden_states):
        for i, (seq, tok) in enumerate(zip(seqs, token_ids)):
            seq.append_token(tok, hidden_states[i])
            seq.decode_input_embeds = None
            finish = (not seq.ignore_eos) and tok == self.eos
            if seq.request_id is None:
                finish = finish or seq.num_completion_tokens >= seq.max_tokens
            if finish:
                seq.status = SequenceStatus.FINISHED
                if seq.request_id is not None:
                    self.request_id_to_seq.pop(seq.request_id, None)
                self.block_manager.deallocate(seq)
                self.running.remove(seq)


class TalkerLLMEngine(LLMEngine):
    def __init__(self, model, **kwargs):
        super().__init__(model, **kwargs)
        self.model_runner = TalkerModeModelRunner(self.config, 0, self.events, **self._runner_kwargs)
        self.scheduler = TalkerScheduler(self.config)

    def add_request(self, inputs_embeds: list[torch.Tensor], sampling_params, request_id: str | None = None):
        """First call for a request id creates the sequence (prefill); later calls feed the
        next decode input (reference :93-109)."""
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(inputs_embeds)
        for emb, sp in zip(inputs_embeds, sampling_params):
            if request_id is not None and request_id in self.scheduler.request_id_to_seq:
                self.scheduler.request_id_to_seq[request_id].decode_input_embeds = emb
                return
            seq = Sequence([], input_embeds=emb, sampling_params=sp, request_id=request_id)
            if request_id is not None:
                self.scheduler.request_id_to_seq[request_id] = seq
            self.scheduler.add(seq)

    def clear_request(self, request_id: str):
        self.scheduler.clear_request(request_id)

    def _step(self):
        seqs, is_prefill = self.scheduler.schedule()
        if not seqs:
            return None
        token_ids, hidden = self.model_runner.call("run", seqs, is_prefill)
        self.scheduler.postprocess(seqs, token_ids, hidden)
        finished = [(s.request_id, s.seq_id, s.completion_token_ids, s.last_hidden_state) for s in seqs if s.is_finished]
        num_tokens = sum(len(s) for s in seqs) if is_prefill else -len(seqs)
        return seqs, finished, num_tokens

    def step(self):
        r = self._step()
        if r is None:
            return [], 0
        return r[1], r[2]

    def step_with_outputs(self):
        r = self._step()
        if r is None:
            return [], 0, []
        seqs, finished, num_tokens = r
        outputs_all = [(s.request_id, s.seq_id, s.completion_token_ids, s.last_hidden_state, s.is_finished) for s in seqs]
        return finished, num_tokens, outputs_all
