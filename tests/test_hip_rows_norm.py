"""Steps of more than 16 rows through the decoder stack (prompts, and decode steps of 17+ streams): the GEMMs take the
prompt kernel, the residual stream stays in fragment order between the launches of a layer, and the GEMM that consumes
normalised rows (gate/up, the next layer's qkv, the output head) normalises them itself from the sums of squares the
producing GEMM (o_proj, a single-pass down_proj) left behind - no RMSNorm launch in between.  Checked against the
oracle's decoder stack (P/models/qwen3_tts_share.py:79-176 with P/layers/layernorm.py:17-27 between the GEMMs) at the
two widths the released models have: hidden 2048 / intermediate 6144 (talker; down_proj K-split, whose finishing
launch applies the next norm) and hidden 1024 / intermediate 3072 (code predictor; down_proj in one pass)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import qtts_oracle as O  # noqa: E402
from test_hip_fp8 import _dequantised, _weights  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = {"talker": (2048, 6144, 3072), "predictor": (1024, 3072, 2048)}


def _stack(kind, weight_dtype="bf16", seed=5, layers=2, max_rows=320, kv_blocks=48):
    from nano_qwen3tts_vllm import hip
    from nano_qwen3tts_vllm.models.hip_decoder import DecoderShape, HipDecoder
    H, I, V = SHAPES[kind]
    shape = DecoderShape(hidden_size=H, num_layers=layers, num_heads=16, num_kv_heads=8, head_dim=128, intermediate_size=I,
                         rope_theta=1000000.0)
    w = _weights(shape, seed)
    g = torch.Generator().manual_seed(seed + 1)
    w["head.weight"] = (torch.randn(V, H, generator=g) * 0.05).bfloat16()
    dec = HipDecoder(w, shape, num_kv_blocks=kv_blocks, max_rows=max_rows, device=DEV, weight_dtype=weight_dtype)
    head = hip.pack_weight(w["head.weight"].to(DEV))
    cfg = O.DecoderCfg(H, layers, 16, 8, 128, I, V, 1e-6, 1000000.0)
    return dec, head, w, cfg


def _close(got, want, alt, what):
    """bf16 results of the same arithmetic in another fp32 summation order, judged against what the oracle differs from
    itself by when its GEMM sums are re-ordered (`alt`: qtts_oracle.other_summation_order): on average no more than
    twice that, at most three times its largest difference (+ one bf16 step of the largest value) - a wrong row scale
    or a misplaced fragment shows on every element."""
    got, want, alt = got.float().cpu(), want.float(), alt.float()
    assert torch.isfinite(got).all(), what
    assert float(want.abs().mean()) > 0.05, what            # (not trivially zero)
    d, noise = (got - want).abs(), (alt - want).abs()
    step = float(want.abs().max()) * 2.0 ** -7
    assert float(d.mean()) <= 2.0 * float(noise.mean()) + 5e-4, f"{what}: mean diff {float(d.mean()):.5f}, the oracle's own {float(noise.mean()):.5f}"
    assert float(d.max()) <= 3.0 * float(noise.max()) + step, f"{what}: max diff {float(d.max()):.4f}, the oracle's own {float(noise.max()):.4f}"
    return float(d.max()), float(d.mean())


@pytest.mark.parametrize("kind,lens", [("talker", (20,)), ("talker", (33,)), ("talker", (64,)), ("talker", (100,)), ("talker", (200,)),
                                       ("talker", (17, 30)), ("predictor", (20,)), ("predictor", (40,)), ("predictor", (64,)),
                                       ("predictor", (130,)), ("predictor", (2,) * 12), ("predictor", (2,) * 40),
                                       # either side of the row counts up to which the norms ride in the GEMMs
                                       ("talker", (65,)), ("talker", (48, 16)), ("predictor", (256,)), ("predictor", (257,))])
def test_prompt_rows_against_the_oracle_stack(kind, lens):
    """Packed prompts of 17..257 rows (the predictor's 2-position prompts of 12 and 40 streams included) through two
    layers + final norm + head: hidden state and logits against the oracle."""
    dec, head, w, cfg = _stack(kind)
    H, V = cfg.hidden_size, cfg.vocab_size
    g = torch.Generator().manual_seed(sum(lens) + len(lens))
    N = sum(lens)
    x = (torch.randn(N, H, generator=g) * 0.5).bfloat16()
    pos = torch.cat([torch.arange(L) for L in lens]).to(torch.int64)
    rs = torch.cat([torch.full((L,), i) for i, L in enumerate(lens)]).to(torch.int32)
    # sequence i owns the KV blocks from first[i] on (a prompt of more than 256 tokens has two)
    nblk = [(L + 255) // 256 for L in lens]
    first = [sum(nblk[:i]) for i in range(len(lens))]
    slots = (torch.tensor(first)[rs.long()] * 256 + pos).to(torch.int32)
    ctx = (pos + 1).to(torch.int32)
    bt = torch.zeros(len(lens), 4, dtype=torch.int32)
    for i in range(len(lens)):
        bt[i, :nblk[i]] = torch.arange(first[i], first[i] + nblk[i])
    hidden = torch.zeros(N, H, dtype=torch.bfloat16, device=DEV)
    logits = torch.zeros(N, V, dtype=torch.bfloat16, device=DEV)
    dec.forward_logits(x.to(DEV), pos.to(DEV), slots.to(DEV), ctx.to(DEV), bt.to(DEV), max(lens), head, V, hidden, logits,
                       row_seq=rs.to(DEV))
    torch.cuda.synchronize()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    want = O.decoder_stack(w, cfg, x, O.AttnMeta(True, pos, None, cu_seqlens=cu), None)
    with O.other_summation_order():
        alt = O.decoder_stack(w, cfg, x, O.AttnMeta(True, pos, None, cu_seqlens=cu), None)
    mx, mean = _close(hidden, want, alt, f"{kind} {lens} hidden")
    lin = torch.nn.functional.linear
    _close(logits, lin(want, w["head.weight"]), lin(alt, w["head.weight"]), f"{kind} {lens} logits")
    # the hidden state without a head (the stand-alone final norm over fragment-order rows) is the same arithmetic
    h2 = dec.forward(x.to(DEV), pos.to(DEV), slots.to(DEV), ctx.to(DEV), bt.to(DEV), max(lens), row_seq=rs.to(DEV))
    torch.cuda.synchronize()
    _close(h2, want, alt, f"{kind} {lens} hidden (no head)")
    print(f"[rows] {kind} prompts {lens}: hidden within {mx:.4f} of the oracle (mean {mean:.5f})")


@pytest.mark.parametrize("kind,B,L", [("talker", 20, 6), ("talker", 40, 3), ("predictor", 24, 5), ("predictor", 48, 2),
                                      ("predictor", 17, 9), ("talker", 64, 2), ("talker", 72, 2), ("predictor", 128, 2)])
def test_decode_steps_of_more_than_sixteen_rows_against_the_oracle_stack(kind, B, L):
    """B sequences prefilled with L tokens each (one packed step), then ONE decode step of B rows: every row against the
    oracle's stack over the sequence's L + 1 tokens (causal attention: the last row of that prompt is the decode step).
    This is the step a frame of 17+ streams runs - decode attention writing its rows in fragment order for o_proj, the
    norms inside the GEMMs."""
    dec, head, w, cfg = _stack(kind, seed=9, kv_blocks=max(48, B))
    H, V = cfg.hidden_size, cfg.vocab_size
    g = torch.Generator().manual_seed(B * 100 + L)
    xs = [(torch.randn(L + 1, H, generator=g) * 0.5).bfloat16() for _ in range(B)]
    pos = torch.arange(L).repeat(B).to(torch.int64)
    rs = torch.arange(B).repeat_interleave(L).to(torch.int32)
    slots = (rs.to(torch.int64) * 256 + pos).to(torch.int32)
    bt = torch.zeros(B, 4, dtype=torch.int32)
    bt[:, 0] = torch.arange(B)
    xp = torch.cat([x[:L] for x in xs]).to(DEV)
    dec.forward(xp, pos.to(DEV), slots.to(DEV), (pos + 1).to(torch.int32).to(DEV), bt.to(DEV), L, row_seq=rs.to(DEV))
    xd = torch.stack([x[L] for x in xs]).to(DEV)
    hidden = torch.zeros(B, H, dtype=torch.bfloat16, device=DEV)
    logits = torch.zeros(B, V, dtype=torch.bfloat16, device=DEV)
    dec.forward_logits(xd, torch.full((B,), L, dtype=torch.int64, device=DEV),
                       (torch.arange(B, dtype=torch.int32) * 256 + L).to(DEV),
                       torch.full((B,), L + 1, dtype=torch.int32, device=DEV), bt.to(DEV), 128, head, V, hidden, logits)
    torch.cuda.synchronize()
    meta = O.AttnMeta(True, torch.arange(L + 1), None, cu_seqlens=torch.tensor([0, L + 1], dtype=torch.int32))
    want = torch.stack([O.decoder_stack(w, cfg, x, meta, None)[L] for x in xs])
    with O.other_summation_order():
        alt = torch.stack([O.decoder_stack(w, cfg, x, meta, None)[L] for x in xs])
    mx, mean = _close(hidden, want, alt, f"{kind} decode {B} rows hidden")
    lin = torch.nn.functional.linear
    _close(logits, lin(want, w["head.weight"]), lin(alt, w["head.weight"]), f"{kind} decode {B} rows logits")
    print(f"[rows] {kind} decode step of {B} rows: hidden within {mx:.4f} of the oracle (mean {mean:.5f})")


@pytest.mark.parametrize("kind,N", [("talker", 48), ("predictor", 40), ("predictor", 100)])
def test_fp8_rows_equal_bf16_rows_on_dequantised_weights(kind, N):
    """The same forms over fp8 weights return bit for bit what the bf16 kernels return on the dequantised weights."""
    from nano_qwen3tts_vllm.models.hip_decoder import DecoderShape, HipDecoder
    H, I, V = SHAPES[kind]
    shape = DecoderShape(hidden_size=H, num_layers=2, num_heads=16, num_kv_heads=8, head_dim=128, intermediate_size=I,
                         rope_theta=1000000.0)
    w = _weights(shape, 13)
    g = torch.Generator().manual_seed(N)
    x = (torch.randn(N, H, generator=g) * 0.5).bfloat16().to(DEV)
    pos = torch.arange(N, dtype=torch.int64, device=DEV)
    slots = torch.arange(N, dtype=torch.int32, device=DEV)
    ctx = torch.arange(1, N + 1, dtype=torch.int32, device=DEV)
    rs = torch.zeros(N, dtype=torch.int32, device=DEV)
    bt = torch.zeros(1, 4, dtype=torch.int32, device=DEV)
    outs = []
    for weights, dt in ((w, "fp8"), (_dequantised(w), "bf16")):
        dec = HipDecoder(weights, shape, num_kv_blocks=4, max_rows=256, device=DEV, weight_dtype=dt)
        outs.append(dec.forward(x, pos, slots, ctx, bt, N, row_seq=rs).clone())
        torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
