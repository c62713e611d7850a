"""The tables a request needs BEFORE the engines see it: text embedding, codec embedding, the text
projection MLP and the predictor's 15 codebook tables
(reference: P/utils/embedding_loader.py:17-112, which instantiates fp32 nn.Modules and loads the bf16
checkpoint tensors into them; the same fp32 copies are kept here as plain tensors on the device).
"""
from __future__ import annotations

import os

import torch

from ..config import Qwen3TTSConfig


class Table:
    """ids -> rows of an fp32 table (what nn.Embedding.forward does)."""

    def __init__(self, weight: torch.Tensor):
        self.weight = weight

    def __call__(self, ids: torch.Tensor) -> torch.Tensor:
        return self.weight[ids.to(self.weight.device).long()]

    def to(self, device):
        return Table(self.weight.to(device))


class ResizeMLP:
    """text_hidden -> text_hidden -> talker hidden with biases and SiLU
    (P/models/qwen3_tts_talker.py:22-31, `Qwen3TTSTalkerResizeMLP`)."""

    def __init__(self, w1, b1, w2, b2):
        self.w1, self.b1, self.w2, self.b2 = w1, b1, w2, b2

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        h = torch.nn.functional.silu(torch.nn.functional.linear(x, self.w1, self.b1))
        return torch.nn.functional.linear(h, self.w2, self.b2)


def load_embeddings_only(model_path: str, device: str = "cpu"):
    """(model_config, text_embedding, input_embedding, text_projection, predictor_input_embeddings).
    text_embedding / text_projection are None when the checkpoint carries no text tables (decode-only
    synthetic model directories)."""
    from safetensors import safe_open

    cfg = Qwen3TTSConfig.from_model_dir(model_path)
    dev = torch.device(device)
    want_prefixes = ("talker.model.text_embedding.", "talker.model.codec_embedding.", "talker.model.embed_tokens.",
                     "talker.text_projection.", "talker.code_predictor.model.codec_embedding.")
    t = {}
    with safe_open(os.path.join(model_path, "model.safetensors"), framework="pt", device="cpu") as f:
        for k in f.keys():
            if k.startswith(want_prefixes):
                t[k] = f.get_tensor(k).float().to(dev)
    codec = t.get("talker.model.codec_embedding.weight", t.get("talker.model.embed_tokens.weight"))
    if codec is None:
        raise KeyError("checkpoint has no talker codec embedding table")
    input_embedding = Table(codec)
    text_embedding = Table(t["talker.model.text_embedding.weight"]) if "talker.model.text_embedding.weight" in t else None
    text_projection = None
    if "talker.text_projection.linear_fc1.weight" in t:
        text_projection = ResizeMLP(t["talker.text_projection.linear_fc1.weight"], t.get("talker.text_projection.linear_fc1.bias"),
                                    t["talker.text_projection.linear_fc2.weight"], t.get("talker.text_projection.linear_fc2.bias"))
    n = cfg.talker_config.code_predictor_config.num_code_groups - 1
    preds = [Table(t[f"talker.code_predictor.model.codec_embedding.{i}.weight"]) for i in range(n)]
    return cfg, text_embedding, input_embedding, text_projection, preds
