"""Synthetic Qwen3-TTS checkpoints with the released key layout.

No weights ship with this repository and there is no network, so tests and the
benchmark build random-initialised models of the right architecture.  The key
layout is the single-file `model.safetensors` one the reference loads
(P/engine/model_runner/talker_mode_runner.py:44-47; prefixes handled in
P/models/qwen3_tts_talker.py:81-109 and P/models/qwen3_tts_predictor.py:63-80):
`talker.model.layers.N...`, `talker.codec_head.weight`,
`talker.code_predictor.model.layers.N...`, `talker.code_predictor.lm_head.i.weight`.
"""
from __future__ import annotations

import json
import os

import torch

# Released-checkpoint shapes (re-read from config.json by the engines; these are
# only used to synthesise a config when none exists).
SHAPES = {
    "1.7B": dict(hidden_size=2048, intermediate_size=6144, num_hidden_layers=28),
    "0.6B": dict(hidden_size=1024, intermediate_size=3072, num_hidden_layers=28),
    "tiny": dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2),
    "tiny-id": dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2),
}


def make_config(size: str = "1.7B") -> dict:
    """A config.json dict in the reference's schema (P/config.py:288-313,473-509,575-588)."""
    s = SHAPES[size]
    tiny = size.startswith("tiny")
    talker = dict(
        vocab_size=512 if tiny else 3072,
        hidden_size=s["hidden_size"],
        intermediate_size=s["intermediate_size"],
        num_hidden_layers=s["num_hidden_layers"],
        num_attention_heads=4 if tiny else 16,
        num_key_value_heads=2 if tiny else 8,
        head_dim=128,
        rms_norm_eps=1e-6,
        rope_theta=1000000.0,
        rope_scaling={"mrope_section": [24, 20, 20], "interleaved": True, "rope_type": "default"},
        attention_bias=False,
        max_position_embeddings=32768,
        num_code_groups=16,
        text_vocab_size=64 if tiny else 151936,
        text_hidden_size=64 if tiny else 2048,
        code_predictor_config=dict(
            vocab_size=256 if tiny else 2048,
            hidden_size=128 if tiny else 1024,
            intermediate_size=256 if tiny else 3072,
            num_hidden_layers=2 if tiny else 5,
            num_attention_heads=4 if tiny else 16,
            num_key_value_heads=2 if tiny else 8,
            head_dim=128,
            rms_norm_eps=1e-6,
            rope_theta=1000000.0,
            attention_bias=False,
            max_position_embeddings=32768,
            num_code_groups=16,
        ),
    )
    # special ids of the prompt layout, placed at the top of each vocabulary
    V, TV = talker["vocab_size"], talker["text_vocab_size"]
    talker.update(codec_pad_id=V - 12, codec_bos_id=V - 11, codec_eos_token_id=V - 10, codec_think_id=V - 9,
                  codec_nothink_id=V - 8, codec_think_bos_id=V - 7, codec_think_eos_id=V - 6,
                  codec_language_id={"english": V - 20, "chinese": V - 19, "sichuan_dialect": V - 18},
                  spk_id={"vivian": V - 30, "eric": V - 29}, spk_is_dialect={"eric": "sichuan_dialect"}, hidden_act="silu")
    return dict(talker_config=talker, tts_model_size=size, tts_model_type="voice_design",
                tts_pad_token_id=TV - 4, tts_bos_token_id=TV - 3, tts_eos_token_id=TV - 2,
                im_start_token_id=TV - 6, im_end_token_id=TV - 5)


def _decoder_keys(prefix, n_layers, hidden, heads, kv_heads, head_dim, inter):
    shapes = {}
    for i in range(n_layers):
        p = f"{prefix}layers.{i}."
        shapes[p + "self_attn.q_proj.weight"] = (heads * head_dim, hidden)
        shapes[p + "self_attn.k_proj.weight"] = (kv_heads * head_dim, hidden)
        shapes[p + "self_attn.v_proj.weight"] = (kv_heads * head_dim, hidden)
        shapes[p + "self_attn.o_proj.weight"] = (hidden, heads * head_dim)
        shapes[p + "self_attn.q_norm.weight"] = (head_dim,)
        shapes[p + "self_attn.k_norm.weight"] = (head_dim,)
        shapes[p + "mlp.gate_proj.weight"] = (inter, hidden)
        shapes[p + "mlp.up_proj.weight"] = (inter, hidden)
        shapes[p + "mlp.down_proj.weight"] = (hidden, inter)
        shapes[p + "input_layernorm.weight"] = (hidden,)
        shapes[p + "post_attention_layernorm.weight"] = (hidden,)
    shapes[prefix + "norm.weight"] = (hidden,)
    return shapes


def checkpoint_shapes(config: dict, with_text: bool = False) -> dict:
    t = config["talker_config"]
    p = t["code_predictor_config"]
    H, h = t["hidden_size"], p["hidden_size"]
    shapes = _decoder_keys("talker.model.", t["num_hidden_layers"], H, t["num_attention_heads"],
                           t["num_key_value_heads"], t["head_dim"], t["intermediate_size"])
    shapes["talker.model.codec_embedding.weight"] = (t["vocab_size"], H)
    shapes["talker.codec_head.weight"] = (t["vocab_size"], H)
    if with_text:
        th = t["text_hidden_size"]
        shapes["talker.model.text_embedding.weight"] = (t["text_vocab_size"], th)
        shapes["talker.text_projection.linear_fc1.weight"] = (th, th)
        shapes["talker.text_projection.linear_fc1.bias"] = (th,)
        shapes["talker.text_projection.linear_fc2.weight"] = (H, th)
        shapes["talker.text_projection.linear_fc2.bias"] = (H,)
    shapes.update(_decoder_keys("talker.code_predictor.model.", p["num_hidden_layers"], h,
                                p["num_attention_heads"], p["num_key_value_heads"], p["head_dim"],
                                p["intermediate_size"]))
    for i in range(p["num_code_groups"] - 1):
        shapes[f"talker.code_predictor.model.codec_embedding.{i}.weight"] = (p["vocab_size"], H)
        shapes[f"talker.code_predictor.lm_head.{i}.weight"] = (p["vocab_size"], h)
    if h != H:
        shapes["talker.code_predictor.small_to_mtp_projection.weight"] = (h, H)
        shapes["talker.code_predictor.small_to_mtp_projection.bias"] = (h,)
    return shapes


def make_state_dict(config: dict, seed: int = 1234, norm_jitter: float = 0.0, std: float = 0.02,
                    with_text: bool = False, embed_std: float | None = None, head_sharp: float | None = None,
                    head_std: float | None = None) -> dict:
    """bf16 tensors, N(0, std^2); norm weights 1 (+ jitter*N(0,1) when asked, so
    that the bf16 `* weight` rounding step is exercised).
    head_std / head_sharp shape the output heads (`codec_head`, `lm_head.i`): rows N(0, head_std^2), each row then
    scaled by exp(head_sharp * N(0,1)).  Random heads give logits whose two largest values are often a bf16 step
    apart (greedy ids of such a model are decided by the last bit of a sum); log-normal row scales spread the top
    of the distribution the way a trained head does, so that almost every greedy pick has a clear winner."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in checkpoint_shapes(config, with_text).items():
        if k.endswith("norm.weight") or k.endswith("layernorm.weight"):
            w = torch.ones(shp)
            if norm_jitter:
                w = w + norm_jitter * torch.randn(shp, generator=g)
        elif "codec_embedding" in k and embed_std is not None:
            w = torch.randn(shp, generator=g) * embed_std
        elif ("codec_head" in k or "lm_head" in k) and (head_sharp is not None or head_std is not None):
            w = torch.randn(shp, generator=g) * (head_std if head_std is not None else std)
            if head_sharp:
                w = w * torch.exp(head_sharp * torch.randn(shp[0], 1, generator=g))
        else:
            w = torch.randn(shp, generator=g) * std
        sd[k] = w.to(torch.bfloat16)
    return sd


def split_state_dict(sd: dict):
    """(talker weights, predictor weights) with the prefixes stripped the way the
    reference's convert_state_dict methods do."""
    tw, pw = {}, {}
    for k, v in sd.items():
        if k.startswith("talker.code_predictor."):
            pw[k[len("talker.code_predictor."):]] = v
        elif k.startswith("speaker_encoder."):
            continue
        elif k.startswith("talker."):
            tw[k[len("talker."):]] = v
        else:
            tw[k] = v
    return tw, pw


def write_model_dir(path: str, size: str = "tiny", seed: int = 1234, **kw) -> dict:
    """Write config.json + model.safetensors the way a released model dir looks."""
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    cfg = make_config(size)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    save_file(make_state_dict(cfg, seed, **kw), os.path.join(path, "model.safetensors"))
    return cfg


def write_tokenizer(path: str, text_vocab_size: int = 64) -> None:
    """A character-level tokenizer in the model directory (`tokenizer.json` + `tokenizer_config.json`, loadable
    offline by `AutoTokenizer.from_pretrained(path)` as the processor does): the released directories ship a Qwen2
    BPE tokenizer, which cannot be fetched here.  The chat markers are single ids; lower-cased letters, blank,
    newline and basic punctuation are ids of their own; everything else is `<unk>`.  All ids stay below the tts
    special ids at the top of the text vocabulary (make_config)."""
    from tokenizers import AddedToken, Regex, Tokenizer, models, normalizers, pre_tokenizers

    vocab = {"<unk>": 0, "<|im_start|>": 1, "<|im_end|>": 2, "<pad>": 3}
    for c in " abcdefghijklmnopqrstuvwxyz.,!?'\n":
        vocab[c] = len(vocab)
    assert len(vocab) <= text_vocab_size - 8, "text vocabulary too small for the character set"
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.normalizer = normalizers.Lowercase()
    tok.pre_tokenizer = pre_tokenizers.Split(Regex("."), behavior="isolated")
    tok.add_special_tokens([AddedToken(t, special=True, normalized=False) for t in ("<|im_start|>", "<|im_end|>", "<pad>")])
    os.makedirs(path, exist_ok=True)
    tok.save(os.path.join(path, "tokenizer.json"))
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "unk_token": "<unk>", "pad_token": "<pad>"}, f)
