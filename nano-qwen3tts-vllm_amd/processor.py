"""Text processor of the API object: the tokenizer stored in the model directory behind the call
signature the prompt builder uses (reference: P/processor.py:27-106, a ProcessorMixin around a Qwen2
tokenizer; `P/interface.py:217-222` loads it from the model path)."""
from __future__ import annotations


class Qwen3TTSProcessor:
    def __init__(self, tokenizer=None, chat_template=None):
        self.tokenizer = tokenizer
        self.chat_template = chat_template

    @classmethod
    def from_pretrained(cls, model_path: str, **kwargs):
        from transformers import AutoTokenizer
        kwargs.pop("fix_mistral_regex", None)
        return cls(AutoTokenizer.from_pretrained(model_path, **kwargs))

    def __call__(self, text=None, **kwargs):
        if text is None:
            raise ValueError("You need to specify either a `text` input to process.")
        if not isinstance(text, list):
            text = [text]
        return self.tokenizer(text, **kwargs)

    def batch_decode(self, *args, **kwargs):
        return self.tokenizer.batch_decode(*args, **kwargs)

    def decode(self, *args, **kwargs):
        return self.tokenizer.decode(*args, **kwargs)

    @property
    def model_input_names(self):
        return list(dict.fromkeys(self.tokenizer.model_input_names))
