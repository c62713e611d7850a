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


def model_dir(size: str, path: str | None = None, seed: int = 1234) -> str:
    """config.json + model.safetensors with N(0, 0.02^2) bf16 weights (embeddings 0.05), generated on the GPU
    and written once; reused if it is already there."""
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
    for k, shp in synthetic.checkpoint_shapes(cfg).items():
        if k.endswith("norm.weight") or k.endswith("layernorm.weight"):
            sd[k] = torch.ones(shp, dtype=torch.bfloat16)
        else:
            std = 0.05 if "codec_embedding" in k else 0.02
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
