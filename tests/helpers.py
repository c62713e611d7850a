"""Shared helpers for the parity tests (fixtures → torch tensors, synthetic weights)."""
import os

import numpy as np
import torch

from nano_qwen3tts_vllm.utils import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def bf(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16)


def golden_weights(size, seed):
    """The same seeded synthetic checkpoint oracle/gen_golden.py loaded into the reference."""
    cfg = synthetic.make_config(size)
    sd = synthetic.make_state_dict(cfg, seed=seed, norm_jitter=0.1, with_text=True, embed_std=0.05)
    tw, pw = synthetic.split_state_dict(sd)
    return cfg, sd, tw, pw


def weight_checksum(sd):
    return sum(float(v.float().abs().sum()) for v in sd.values())


def oracle_cfgs(cfg):
    import qtts_oracle as O
    t = cfg["talker_config"]
    p = t["code_predictor_config"]
    tc = O.DecoderCfg(t["hidden_size"], t["num_hidden_layers"], t["num_attention_heads"], t["num_key_value_heads"],
                      t["head_dim"], t["intermediate_size"], t["vocab_size"], t["rms_norm_eps"], t["rope_theta"])
    pc = O.DecoderCfg(p["hidden_size"], p["num_hidden_layers"], p["num_attention_heads"], p["num_key_value_heads"],
                      p["head_dim"], p["intermediate_size"], p["vocab_size"], p["rms_norm_eps"], p["rope_theta"],
                      talker_hidden_size=t["hidden_size"], num_code_groups=p["num_code_groups"])
    return tc, pc
