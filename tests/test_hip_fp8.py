"""fp8 (e4m3) weights through the decoder stack: an extension without a counterpart in the reference, so the
parity statement is internal — the fp8 stack returns bit for bit what the bf16 stack returns on the dequantised
weights q * scale (bf16 numbers), whose arithmetic the other tests pin to the oracle."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import qtts_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _weights(shape, seed):
    g = torch.Generator().manual_seed(seed)
    H, I, hd = shape.hidden_size, shape.intermediate_size, shape.head_dim
    w = {}
    for i in range(shape.num_layers):
        p = f"model.layers.{i}."
        for name, (n, k) in {"self_attn.q_proj": (shape.num_heads * hd, H), "self_attn.k_proj": (shape.num_kv_heads * hd, H),
                             "self_attn.v_proj": (shape.num_kv_heads * hd, H), "self_attn.o_proj": (H, shape.num_heads * hd),
                             "mlp.gate_proj": (I, H), "mlp.up_proj": (I, H), "mlp.down_proj": (H, I)}.items():
            w[p + name + ".weight"] = (torch.randn(n, k, generator=g) * 0.04).bfloat16()
        for name, n in {"input_layernorm": H, "post_attention_layernorm": H, "self_attn.q_norm": hd, "self_attn.k_norm": hd}.items():
            w[p + name + ".weight"] = (1 + 0.1 * torch.randn(n, generator=g)).bfloat16()
    w["model.norm.weight"] = (1 + 0.1 * torch.randn(H, generator=g)).bfloat16()
    return w


def _dequantised(w):
    return {k: (O.quantize_fp8_rows(v)[2] if "_proj.weight" in k else v) for k, v in w.items()}


def test_fp8_stack_equals_bf16_stack_on_dequantised_weights():
    from nano_qwen3tts_vllm.models.hip_decoder import DecoderShape, HipDecoder
    shape = DecoderShape(hidden_size=512, num_layers=3, num_heads=4, num_kv_heads=2, head_dim=128, intermediate_size=1024,
                         rope_theta=1000000.0)
    w = _weights(shape, 21)
    a = HipDecoder(w, shape, num_kv_blocks=4, max_rows=256, device=DEV, weight_dtype="fp8")
    b = HipDecoder(_dequantised(w), shape, num_kv_blocks=4, max_rows=256, device=DEV, weight_dtype="bf16")
    g = torch.Generator().manual_seed(4)
    # a 150-token prompt in one block, then three decode steps of one row
    L = 150
    x = (torch.randn(L, 512, generator=g) * 0.5).bfloat16().to(DEV)
    pos = torch.arange(L, dtype=torch.int64, device=DEV)
    slots = (256 * 2 + torch.arange(L, dtype=torch.int32)).to(DEV)
    ctx = torch.arange(1, L + 1, dtype=torch.int32, device=DEV)
    rs = torch.zeros(L, dtype=torch.int32, device=DEV)
    bt = torch.tensor([[2, 0, 0, 0]], dtype=torch.int32, device=DEV)
    ha = a.forward(x, pos, slots, ctx, bt, L, row_seq=rs)
    hb = b.forward(x, pos, slots, ctx, bt, L, row_seq=rs)
    torch.cuda.synchronize()
    assert torch.equal(ha, hb)
    assert torch.equal(a.kv_cache, b.kv_cache)
    prompt_hidden = ha.clone()
    for step in range(3):
        n = L + step
        xd = (torch.randn(1, 512, generator=g) * 0.5).bfloat16().to(DEV)
        args = (torch.tensor([n], dtype=torch.int64, device=DEV), torch.tensor([256 * 2 + n], dtype=torch.int32, device=DEV),
                torch.tensor([n + 1], dtype=torch.int32, device=DEV), bt, 256)
        ha, hb = a.forward(xd, *args), b.forward(xd, *args)
        torch.cuda.synchronize()
        assert torch.equal(ha, hb), step
    # and the quantisation itself stays small against the stack over the original weights
    c = HipDecoder(w, shape, num_kv_blocks=4, max_rows=256, device=DEV, weight_dtype="bf16")
    hc = c.forward(x, pos, slots, ctx, bt, L, row_seq=rs)
    err = (prompt_hidden.float() - hc.float()).abs().mean() / hc.float().abs().mean()
    assert float(err) < 0.1, float(err)


def test_prompt_stack_with_k_split_down_proj():
    """A 70-token prompt through a stack whose down_proj runs K-split (intermediate size 4096), where the launch that
    finishes down_proj also applies the next layer's / the final RMSNorm: the bf16 stack agrees with the oracle's
    decoder stack (P/models/qwen3_tts_share.py:163-176 with P/layers/layernorm.py:17-27 between the layers) within
    the engine tests' hidden-state tolerance, and the fp8 stack returns bit for bit the bf16 stack's rows on the
    dequantised weights."""
    from nano_qwen3tts_vllm.models.hip_decoder import DecoderShape, HipDecoder
    shape = DecoderShape(hidden_size=512, num_layers=3, num_heads=4, num_kv_heads=2, head_dim=128, intermediate_size=4096,
                         rope_theta=1000000.0)
    w = _weights(shape, 33)
    g = torch.Generator().manual_seed(8)
    L = 70
    x = (torch.randn(L, 512, generator=g) * 0.5).bfloat16()
    pos = torch.arange(L, dtype=torch.int64, device=DEV)
    slots = (256 + torch.arange(L, dtype=torch.int32)).to(DEV)
    ctx = torch.arange(1, L + 1, dtype=torch.int32, device=DEV)
    rs = torch.zeros(L, dtype=torch.int32, device=DEV)
    bt = torch.tensor([[1, 0, 0, 0]], dtype=torch.int32, device=DEV)
    outs = {}
    for name, weights, dt in (("bf16", w, "bf16"), ("fp8", w, "fp8"), ("deq", _dequantised(w), "bf16")):
        dec = HipDecoder(weights, shape, num_kv_blocks=4, max_rows=256, device=DEV, weight_dtype=dt)
        outs[name] = dec.forward(x.to(DEV), pos, slots, ctx, bt, L, row_seq=rs).clone()
        torch.cuda.synchronize()
    assert torch.equal(outs["fp8"], outs["deq"])
    cfg = O.DecoderCfg(512, 3, 4, 2, 128, 4096, 0, 1e-6, 1000000.0)
    meta = O.AttnMeta(True, torch.arange(L), None, cu_seqlens=torch.tensor([0, L], dtype=torch.int32))
    want = O.decoder_stack(w, cfg, x, meta, None)
    got = outs["bf16"].float().cpu()
    assert torch.allclose(got, want.float(), atol=4e-2, rtol=4e-2), float((got - want.float()).abs().max())
    assert float(got.abs().mean()) > 0.1          # (not trivially zero)
