"""Engine base (reference: P/engine/llm_engine/base.py:15-102).

Differences by design: one replica per GPU, so no tensor-parallel rank processes are
spawned; the text tokenizer the reference loads only to read `eos_token_id`
(base.py:32-33) is optional — codec ids never equal the text EOS
(P/engine/llm_engine/talker_llm_engine.py:75), so `eos` stays -1 when no tokenizer
files are present."""
from __future__ import annotations

import atexit
from dataclasses import fields

from ...config import Config


class LLMEngine:
    def __init__(self, model, **kwargs):
        names = {f.name for f in fields(Config)}
        self.config = Config(model, **{k: v for k, v in kwargs.items() if k in names})
        self._runner_kwargs = {k: kwargs[k] for k in ("weights",) if k in kwargs}
        self.ps, self.events = [], []
        self.tokenizer = None
        try:
            import os
            if os.path.exists(os.path.join(model, "tokenizer_config.json")):
                from transformers import AutoTokenizer
                self.tokenizer = AutoTokenizer.from_pretrained(model, use_fast=True)
                self.config.eos = self.tokenizer.eos_token_id
        except Exception:
            self.tokenizer = None
        self.model_runner = None
        self.scheduler = None
        atexit.register(self.exit)

    def exit(self):
        atexit.unregister(self.exit)
        if self.model_runner is not None:
            self.model_runner.call("exit")
            self.model_runner = None

    def is_finished(self):
        return self.scheduler.is_finished()
