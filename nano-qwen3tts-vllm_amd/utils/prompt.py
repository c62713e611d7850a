"""Chat-template strings and tokenisation of a TTS request (reference: P/utils/prompt.py:13-79,
P/interface.py:472-481)."""
from __future__ import annotations

from typing import Any, List, Optional, Union

import torch


def _ensure_list(x: Any) -> List[Any]:
    return x if isinstance(x, list) else [x]


def _build_assistant_text(text: str) -> str:
    return f"<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n"


def _build_instruct_text(instruct: str) -> str:
    return f"<|im_start|>user\n{instruct}<|im_end|>\n"


def _build_ref_text(text: str) -> str:
    return f"<|im_start|>assistant\n{text}<|im_end|>\n"


def _tokenize_texts(texts: List[str], processor, device) -> List[torch.Tensor]:
    """One [1, T] id tensor per text, from `processor(text=..., return_tensors="pt", padding=True)`."""
    out = []
    for text in texts:
        ids = processor(text=text, return_tensors="pt", padding=True)["input_ids"].to(device)
        out.append(ids.unsqueeze(0) if ids.dim() == 1 else ids)
    return out


def _broadcast(value, n: int, default):
    if isinstance(value, list):
        items = value
    else:
        items = [value if value is not None else default] * n
    return items * n if len(items) == 1 and n > 1 else items


def prepare_custom_voice_prompt(text: Union[str, List[str]], speaker: Union[str, List[str]],
                                language: Union[str, List[str]], instruct: Optional[Union[str, List[str]]] = None,
                                non_streaming_mode: bool = True, processor=None, model_size: str = "1.7b",
                                device: torch.device = torch.device("cpu")):
    """(input_ids, instruct_ids, speakers, languages), one entry per text; single values are repeated over the
    batch; the 0.6B models take no instruction."""
    texts = _ensure_list(text)
    n = len(texts)
    if model_size in ("0b6", "0.6b"):
        instruct = None
    languages = _broadcast(language, n, "Auto")
    speakers = _broadcast(_ensure_list(speaker), n, "")
    instructs = _broadcast(instruct, n, "")
    if not (n == len(languages) == len(speakers) == len(instructs)):
        raise ValueError(f"Batch size mismatch: text={n}, language={len(languages)}, speaker={len(speakers)}, "
                         f"instruct={len(instructs)}")
    input_ids = _tokenize_texts([_build_assistant_text(t) for t in texts], processor, device)
    instruct_ids = [None if not ins else _tokenize_texts([_build_instruct_text(ins)], processor, device)[0]
                    for ins in instructs]
    return input_ids, instruct_ids, speakers, languages
