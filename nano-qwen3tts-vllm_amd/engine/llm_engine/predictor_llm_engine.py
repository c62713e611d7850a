"""Code-predictor engine (reference: P/engine/llm_engine/predictor_llm_engine.py:14-108).

`step()` keeps the reference's return shape.  When every scheduled sequence is a
freshly admitted request and nothing else is in flight, the whole 15-token burst
runs as one captured graph inside that single `step()` (the workers call `step()`
until the scheduler is idle, P/workers/predictor_worker.py:128-134, so the observable
result — one finished output per request — is the same).  Set
QTTS_PREDICTOR_BURST=0 to force the reference's step-by-step schedule."""
from __future__ import annotations

import os

import torch

from ...sampling_params import SamplingParams
from ..model_runner.predictor_model_runner import PredictorModelRunner, PredictorSequence
from ..scheduler import Scheduler
from ..sequence import Sequence, SequenceStatus
from .base import LLMEngine


class PredictorScheduler(Scheduler):
    def __init__(self, config):
        super().__init__(config)
        self.request_id_to_seq: dict[str, Sequence] = {}

    def clear_request(self, request_id: str):
        seq = self.request_id_to_seq.pop(request_id, None)
        if seq is None:
            return
        self.block_manager.deallocate(seq)
        if seq in self.running:
            self.running.remove(seq)
        if seq in self.waiting:
            self.waiting.remove(seq)

    def postprocess(self, seqs, token_ids):
        super().postprocess(seqs, token_ids)
        for seq in seqs:
            if seq.is_finished and seq.request_id is not None:
                self.request_id_to_seq.pop(seq.request_id, None)


class PredictorLLMEngine(LLMEngine):
    def __init__(self, model, **kwargs):
        super().__init__(model, **kwargs)
        self.model_runner = PredictorModelRunner(self.config, 0, self.events, **self._runner_kwargs)
        self.scheduler = PredictorScheduler(self.config)
        self.use_burst = os.environ.get("QTTS_PREDICTOR_BURST", "1") != "0"

    def add_request(self, inputs_embeds: list[torch.Tensor], sampling_params, request_id: str | None = None):
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(inputs_embeds)
        for emb, sp in zip(inputs_embeds, sampling_params):
            seq = PredictorSequence([], input_embeds=emb, sampling_params=sp, generation_steps=0, request_id=request_id)
            if request_id is not None:
                self.scheduler.request_id_to_seq[request_id] = seq
            self.scheduler.add(seq)

    def clear_request(self, request_id: str):
        self.scheduler.clear_request(request_id)

    def _burst_ok(self, seqs, is_prefill) -> bool:
        n_gen = self.model_runner.n_gen
        return (self.use_burst and is_prefill and len(self.scheduler.running) == len(seqs)
                and all(len(s) == 2 and s.generation_steps == 0 and s.max_tokens == n_gen + 2 and s.ignore_eos is False
                        for s in seqs)
                and len({s.top_k for s in seqs}) == 1 and self.config.eos < 0)

    def step(self):
        seqs, is_prefill = self.scheduler.schedule()
        if self._burst_ok(seqs, is_prefill):
            bursts = self.model_runner.call("run_burst", seqs)
            for step in range(self.model_runner.n_gen):
                self.scheduler.postprocess(seqs, [b[step] for b in bursts])
                for s in seqs:
                    s.generation_steps += 1
        else:
            token_ids = self.model_runner.call("run", seqs, is_prefill)
            self.scheduler.postprocess(seqs, token_ids)
            for s in seqs:
                s.generation_steps += 1
        outputs = [(s.request_id, s.seq_id, s.completion_token_ids) for s in seqs if s.is_finished]
        num_tokens = sum(len(s) for s in seqs) if is_prefill else -len(seqs)
        return outputs, num_tokens

    def generate(self, inputs_embeds, sampling_params, use_tqdm: bool = True, request_id: str | None = None):
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(inputs_embeds)
        for emb, sp in zip(inputs_embeds, sampling_params):
            self.add_request(emb, sp, request_id=request_id)     # emb iterates as [1,L,H] items, as in the reference
        done = {}
        while not self.is_finished():
            outputs, _ = self.step()
            for rid, seq_id, toks in outputs:
                done[seq_id] = (rid, toks)
        ordered = [done[k] for k in sorted(done)]
        result = [{"text": None, "token_ids": toks, "hidden_states": None} for _rid, toks in ordered]
        if request_id is not None:
            result = [r for r, (rid, _) in zip(result, ordered) if rid == request_id]
        return result
