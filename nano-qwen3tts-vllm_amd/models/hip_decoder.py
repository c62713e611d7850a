"""Device-resident decoder stack: weights packed for the HIP kernels, paged KV
cache and workspace owned here, one C-ABI call per forward.

Replaces the nn.Module stacks of the reference
(P/models/qwen3_tts_talker.py:34-62, P/models/qwen3_tts_predictor.py:17-44,
layer body P/models/qwen3_tts_share.py:13-176) and the KV allocation of
P/engine/model_runner/base.py:131-164.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch

from .. import hip

BLOCK = 256


@dataclass
class DecoderShape:
    hidden_size: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    head_dim: int
    intermediate_size: int
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0

    @staticmethod
    def from_config(c: dict) -> "DecoderShape":
        hd = c.get("head_dim") or c["hidden_size"] // c["num_attention_heads"]
        return DecoderShape(c["hidden_size"], c["num_hidden_layers"], c["num_attention_heads"],
                            c["num_key_value_heads"], hd, c["intermediate_size"],
                            c.get("rms_norm_eps", 1e-6), c.get("rope_theta", 10000.0))


def rope_tables(head_dim: int, theta: float, max_pos: int):
    """bf16 cos/sin [max_pos, head_dim/2], computed on the host in fp32 exactly as
    P/layers/rotary_embedding.py:98,134-139 does (the runners only pass 1-D positions,
    so the talker's three M-RoPE rows are equal and reduce to this table)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    freqs = torch.arange(max_pos, dtype=torch.int64).float()[:, None] * inv_freq[None, :]
    return freqs.cos().to(torch.bfloat16), freqs.sin().to(torch.bfloat16)


class HipDecoder:
    def __init__(self, weights: dict, shape: DecoderShape, num_kv_blocks: int, max_rows: int,
                 max_context: int = 1024, device="cuda", max_pos: int = 2048, prefix: str = "model.",
                 weight_dtype: str | None = None):
        """weight_dtype "bf16" (the reference's) or "fp8": the four GEMM weights of every layer as e4m3 with a
        power-of-two scale per row (an extension; default from QTTS_WEIGHT_DTYPE, else bf16)."""
        self.weight_dtype = weight_dtype or os.environ.get("QTTS_WEIGHT_DTYPE", "bf16")
        assert self.weight_dtype in ("bf16", "fp8"), self.weight_dtype
        fp8 = self.weight_dtype == "fp8"
        self.shape = s = shape
        self.device = torch.device(device)
        self.max_rows, self.max_context, self.max_pos = max_rows, max_context, max_pos
        self.num_kv_blocks = num_kv_blocks
        self._keep = []          # device tensors the C side points into
        self._layer_calls = []   # (entry point name, arguments) per layer, for twins of this stack

        def dev(t):
            t = t.to(device=self.device, dtype=torch.bfloat16).contiguous()
            self._keep.append(t)
            return t

        h = hip.lib.qtts_model_create(s.hidden_size, s.num_layers, s.num_heads, s.num_kv_heads, s.head_dim,
                                      s.intermediate_size, s.rms_eps)
        if not h:
            raise hip.QttsError("model_create: " + hip.lib.qtts_last_error().decode())
        self._h = h
        for i in range(s.num_layers):
            p = f"{prefix}layers.{i}."
            wqkv = torch.cat([weights[p + "self_attn.q_proj.weight"], weights[p + "self_attn.k_proj.weight"],
                              weights[p + "self_attn.v_proj.weight"]], 0)
            if fp8:
                wqkv_p, s_qkv = hip.pack_weight_fp8(dev(wqkv))
                wo_p, s_o = hip.pack_weight_fp8(dev(weights[p + "self_attn.o_proj.weight"]))
                wgu_p, s_gu = hip.pack_gate_up_fp8(dev(weights[p + "mlp.gate_proj.weight"]), dev(weights[p + "mlp.up_proj.weight"]))
                wd_p, s_d = hip.pack_weight_fp8(dev(weights[p + "mlp.down_proj.weight"]))
            else:
                wqkv_p = hip.pack_weight(dev(wqkv))
                wo_p = hip.pack_weight(dev(weights[p + "self_attn.o_proj.weight"]))
                wgu_p = hip.pack_gate_up(dev(weights[p + "mlp.gate_proj.weight"]), dev(weights[p + "mlp.up_proj.weight"]))
                wd_p = hip.pack_weight(dev(weights[p + "mlp.down_proj.weight"]))
            torch.cuda.synchronize()
            del self._keep[-5:]          # row-major copies are no longer needed once packed
            norms = [dev(weights[p + "input_layernorm.weight"]), dev(weights[p + "self_attn.q_norm.weight"]),
                     dev(weights[p + "self_attn.k_norm.weight"]), dev(weights[p + "post_attention_layernorm.weight"])]
            self._keep += [wqkv_p, wo_p, wgu_p, wd_p]
            if i == 0:
                self.layer0_in_norm, self.layer0_wqkv = norms[0], wqkv_p     # for tables of layer-0 rows
                self.layer0_qkv_scales = s_qkv if fp8 else None
            self.packed_gate_up = getattr(self, "packed_gate_up", []) + [wgu_p]
            P = hip.ptr
            if fp8:
                self._keep += [s_qkv, s_o, s_gu, s_d]
                self.gate_up_scales = getattr(self, "gate_up_scales", []) + [s_gu]
                call = ("qtts_model_set_layer_fp8", (i, P(norms[0]), P(wqkv_p), P(s_qkv), P(norms[1]), P(norms[2]), P(wo_p), P(s_o),
                                                     P(norms[3]), P(wgu_p), P(s_gu), P(wd_p), P(s_d)))
            else:
                call = ("qtts_model_set_layer", (i, P(norms[0]), P(wqkv_p), P(norms[1]), P(norms[2]), P(wo_p), P(norms[3]),
                                                 P(wgu_p), P(wd_p)))
            hip.check(getattr(hip.lib, call[0])(h, *call[1]), call[0])
            self._layer_calls.append(call)
        self.final_norm = dev(weights[prefix + "norm.weight"])
        hip.check(hip.lib.qtts_model_set_final_norm(h, hip.ptr(self.final_norm)), "set_final_norm")
        cos, sin = rope_tables(s.head_dim, s.rope_theta, max_pos)
        self.cos, self.sin = dev(cos), dev(sin)
        hip.check(hip.lib.qtts_model_set_rope(h, hip.ptr(self.cos), hip.ptr(self.sin), max_pos), "set_rope")
        # [2, layers, blocks, 256, kv_heads, head_dim] bf16 — the reference's layout
        self.kv_cache = torch.zeros(2, s.num_layers, num_kv_blocks, BLOCK, s.num_kv_heads, s.head_dim,
                                    dtype=torch.bfloat16, device=self.device)
        hip.check(hip.lib.qtts_model_set_kv_cache(h, hip.ptr(self.kv_cache), num_kv_blocks), "set_kv_cache")
        nbytes = hip.lib.qtts_model_workspace_bytes(h, max_rows, max_context)
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        hip.check(hip.lib.qtts_model_set_workspace(h, hip.ptr(self.workspace), nbytes, max_rows, max_context),
                  "set_workspace")

    def twin(self, max_rows: int, max_context: int | None = None) -> "HipDecoder":
        """A second handle on the same weights, RoPE tables and KV cache with a workspace of its own: forwards through
        the twin may run on another HIP stream at the same time as forwards through this stack (their activations never
        share scratch; the KV blocks they touch must be different sequences')."""
        t = object.__new__(HipDecoder)
        t.shape, t.device, t.weight_dtype = self.shape, self.device, self.weight_dtype
        t.max_rows, t.max_context, t.max_pos = max_rows, max_context or self.max_context, self.max_pos
        t.num_kv_blocks, t.kv_cache, t.final_norm, t.cos, t.sin = self.num_kv_blocks, self.kv_cache, self.final_norm, self.cos, self.sin
        t._keep, t._layer_calls, t._parent = [], self._layer_calls, self        # the parent owns the tensors
        s = self.shape
        h = hip.lib.qtts_model_create(s.hidden_size, s.num_layers, s.num_heads, s.num_kv_heads, s.head_dim,
                                      s.intermediate_size, s.rms_eps)
        if not h:
            raise hip.QttsError("model_create: " + hip.lib.qtts_last_error().decode())
        t._h = h
        for name, args in self._layer_calls:
            hip.check(getattr(hip.lib, name)(h, *args), name)
        hip.check(hip.lib.qtts_model_set_final_norm(h, hip.ptr(self.final_norm)), "set_final_norm")
        hip.check(hip.lib.qtts_model_set_rope(h, hip.ptr(self.cos), hip.ptr(self.sin), self.max_pos), "set_rope")
        hip.check(hip.lib.qtts_model_set_kv_cache(h, hip.ptr(self.kv_cache), self.num_kv_blocks), "set_kv_cache")
        nbytes = hip.lib.qtts_model_workspace_bytes(h, t.max_rows, t.max_context)
        t.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        hip.check(hip.lib.qtts_model_set_workspace(h, hip.ptr(t.workspace), nbytes, t.max_rows, t.max_context), "set_workspace")
        return t

    def forward_logits(self, x, positions, slot_mapping, context_lens, block_tables, max_context, head_packed, vocab,
                       hidden_out, logits_out, row_seq=None):
        """Layers + final norm + output head in one C call; writes hidden_out [N,hidden], logits_out [N,vocab]."""
        N = x.shape[0]
        assert x.dtype == torch.bfloat16 and x.is_contiguous() and block_tables.is_contiguous()
        hip.check(hip.lib.qtts_model_forward_logits(self._h, hip.ptr(x), hip.ptr(positions), hip.ptr(slot_mapping),
                                                    hip.ptr(row_seq), hip.ptr(context_lens), hip.ptr(block_tables),
                                                    block_tables.shape[1], max_context, N, hip.ptr(hidden_out),
                                                    hip.ptr(head_packed), vocab, hip.ptr(logits_out), hip.stream()),
                  "model_forward_logits")

    def forward(self, x, positions, slot_mapping, context_lens, block_tables, max_context, row_seq=None, out=None):
        """x [N,hidden] bf16; returns post-final-norm hidden [N,hidden]."""
        N = x.shape[0]
        assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[1] == self.shape.hidden_size
        assert positions.dtype == torch.int64 and context_lens.dtype == torch.int32 and block_tables.dtype == torch.int32
        assert slot_mapping is None or slot_mapping.dtype == torch.int32
        assert block_tables.is_contiguous()
        if out is None:
            out = torch.empty_like(x)
        hip.check(hip.lib.qtts_model_forward(self._h, hip.ptr(x), hip.ptr(positions), hip.ptr(slot_mapping),
                                             hip.ptr(row_seq), hip.ptr(context_lens), hip.ptr(block_tables),
                                             block_tables.shape[1], max_context, N, hip.ptr(out), hip.stream()),
                  "model_forward")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                hip.lib.qtts_model_destroy(self._h)
                self._h = None
        except Exception:
            pass
