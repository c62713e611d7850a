"""Engine configuration (reference: P/config.py:41-66) and a thin reader for the
model directory's config.json (reference schema: P/config.py:288-313,473-509,575-609)."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass


@dataclass
class Config:
    model: str
    max_num_batched_tokens: int = 16384
    max_num_seqs: int = 512
    max_model_len: int = 4096
    gpu_memory_utilization: float = 0.9
    tensor_parallel_size: int = 1
    enforce_eager: bool = False
    hf_config: object | None = None
    eos: int = -1
    kvcache_block_size: int = 256
    num_kvcache_blocks: int = -1
    distributed_port: int = 29500
    process_gpu_memory_fraction: float | None = None

    def __post_init__(self):
        assert os.path.isdir(self.model), f"model directory not found: {self.model}"
        assert self.kvcache_block_size % 256 == 0
        # Concurrent streams are sharded over GPUs as independent replicas; one request
        # does not shard, so tensor parallelism is accepted as a keyword and must be 1.
        assert self.tensor_parallel_size == 1, "this engine runs one replica per GPU (tensor_parallel_size must be 1)"
        self.max_model_len = 1024          # fixed by the reference (P/config.py:64)
        assert self.max_num_batched_tokens >= self.max_model_len


class _NS:
    """Attribute view over a config dict with defaults."""

    def __init__(self, d: dict, defaults: dict):
        self._d = {**defaults, **d}

    def __getattr__(self, k):
        try:
            return self._d[k]
        except KeyError:
            raise AttributeError(k)

    def to_dict(self):
        return dict(self._d)


_PRED_DEFAULTS = dict(vocab_size=2048, hidden_size=1024, intermediate_size=3072, num_hidden_layers=5,
                      num_attention_heads=16, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6,
                      rope_theta=10000, rope_scaling=None, attention_bias=False, num_code_groups=16,
                      max_position_embeddings=32768)
_TALKER_DEFAULTS = dict(vocab_size=3072, hidden_size=1024, intermediate_size=2048, num_hidden_layers=20,
                        num_attention_heads=16, num_key_value_heads=2, head_dim=None, rms_norm_eps=1e-6,
                        rope_theta=10000, rope_scaling=None, attention_bias=False, num_code_groups=16,
                        text_vocab_size=151936, text_hidden_size=2048, max_position_embeddings=32768, hidden_act="silu",
                        # special codec ids and the speaker / language tables of the prompt layout (P/config.py:473-509)
                        codec_eos_token_id=4198, codec_think_id=4202, codec_nothink_id=4203, codec_think_bos_id=4204,
                        codec_think_eos_id=4205, codec_pad_id=4196, codec_bos_id=4197, spk_id=None, spk_is_dialect=None,
                        codec_language_id=None)
_TOP_DEFAULTS = dict(tts_model_size=None, tts_model_type=None, im_start_token_id=151644, im_end_token_id=151645,
                     tts_pad_token_id=151671, tts_bos_token_id=151672, tts_eos_token_id=151673)     # P/config.py:575-609


class Qwen3TTSConfig:
    def __init__(self, talker_config=None, **kw):
        tc = dict(talker_config or {})
        pc = dict(tc.pop("code_predictor_config", None) or {})
        self.talker_config = _NS(tc, _TALKER_DEFAULTS)
        self.talker_config._d["code_predictor_config"] = _NS(pc, _PRED_DEFAULTS)
        if self.talker_config.head_dim is None:
            self.talker_config._d["head_dim"] = self.talker_config.hidden_size // self.talker_config.num_attention_heads
        for k in ("spk_id", "spk_is_dialect", "codec_language_id"):
            if self.talker_config._d[k] is None:
                self.talker_config._d[k] = {}
        self.extra = kw
        for k, v in _TOP_DEFAULTS.items():
            setattr(self, k, kw.get(k, v))

    @staticmethod
    def from_model_dir(path: str) -> "Qwen3TTSConfig":
        with open(os.path.join(path, "config.json")) as f:
            return Qwen3TTSConfig(**json.load(f))
