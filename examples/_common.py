"""Shared by the example drivers: a synthetic model directory of the released architecture (there are no
checkpoints on an offline box) and synthetic requests shaped like the reference's prompt builder output."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FRAME_SECONDS = 0.08          # 12.5 Hz codec frames


class CharProcessor:
    """Stand-in tokenizer for an offline box (the released tokenizer files are not available): the chat markers
    are single ids, every other character folds into the text vocabulary.  Prompt lengths therefore count
    characters, not word pieces."""
    MARKS = {"<|im_start|>": 1, "<|im_end|>": 2}

    def __init__(self, vocab: int):
        self.vocab = vocab

    def __call__(self, text=None, return_tensors="pt", padding=True):
        text = text[0] if isinstance(text, list) else text
        ids, i = [], 0
        while i < len(text):
            for m, v in self.MARKS.items():
                if text.startswith(m, i):
                    ids.append(v)
                    i += len(m)
                    break
            else:
                ids.append(3 + ord(text[i]) % (self.vocab - 10))
                i += 1
        return {"input_ids": torch.tensor([ids], dtype=torch.long)}


def model_dir(size: str, path: str | None = None, seed: int = 1234) -> str:
    """config.json + model.safetensors with N(0, 0.02^2) bf16 weights (embeddings 0.05) including the text
    tables of the prompt front-end, generated on the GPU and written once; reused if it is already there."""
    from safetensors.torch import save_file
    from nano_qwen3tts_vllm.utils import synthetic
    path = path or os.path.join(os.environ.get("TMPDIR", "/tmp"), f"qtts_synthetic_{size}")
    if os.path.exists(os.path.join(path, "model.safetensors")):
        return path
    os.makedirs(path, exist_ok=True)
    cfg = synthetic.make_config(size)
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    g = torch.Generator(device=dev).manual_seed(seed)
    sd = {}
    for k, shp in synthetic.checkpoint_shapes(cfg, with_text=True).items():
        if k.endswith("norm.weight") or k.endswith("layernorm.weight"):
            sd[k] = torch.ones(shp, dtype=torch.bfloat16)
        else:
            std = 0.05 if "embedding" in k else 0.02
            sd[k] = (torch.randn(shp, generator=g, device=dev) * std).to(torch.bfloat16).cpu()
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    t0 = time.time()
    save_file(sd, os.path.join(path, "model.safetensors"))
    print(f"[synthetic] wrote {path} ({sum(v.numel() for v in sd.values()) * 2 / 1e9:.2f} GB, {time.time() - t0:.1f} s)",
          file=sys.stderr)
    return path


def request(hidden: int, prompt_len: int, text_frames: int, seed: int):
    """(inputs_embeds [1,L,H], trailing_text_hiddens [1,T,H], tts_pad_embed [1,1,H]) — what the reference's
    prepare_inputs hands to generate_async (P/interface.py:1184-1190)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(1, prompt_len, hidden, generator=g) * 0.5,
            torch.randn(1, text_frames, hidden, generator=g) * 0.1,
            torch.randn(1, 1, hidden, generator=g) * 0.1)
