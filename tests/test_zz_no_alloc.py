"""GPU: qtts_attention_decode_fused_static (the entry point that allocates nothing: split partials and arrival
counters in a static device array inside the library) against qtts_attention_decode_fused with a caller-provided
workspace — same kernels, so the outputs and the KV cache must be bit-identical — on two calls in a row, the second
with a smaller input (fewer rows, shorter contexts), and its size limits."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(hip, B, Hq, Hkv, ctxs, seed):
    from nano_qwen3tts_vllm.models.hip_decoder import rope_tables
    g = torch.Generator().manual_seed(seed)
    nblk = 4 * B + 1
    kc = torch.randn(nblk, 256, Hkv, 128, generator=g).bfloat16()
    vc = torch.randn(nblk, 256, Hkv, 128, generator=g).bfloat16()
    qkv = torch.randn(B, (Hq + 2 * Hkv) * 128, generator=g).bfloat16().to(DEV)
    qw = (1 + 0.1 * torch.randn(128, generator=g)).bfloat16().to(DEV)
    kw = (1 + 0.1 * torch.randn(128, generator=g)).bfloat16().to(DEV)
    bt = (torch.randperm(nblk - 1, generator=g) + 1)[:4 * B].reshape(B, 4).to(torch.int32)
    cl = torch.tensor(ctxs, dtype=torch.int32)
    pos = torch.tensor([max(c, 1) for c in ctxs], dtype=torch.int64).to(DEV)
    slots = torch.tensor([int(bt[i, (c - 1) // 256]) * 256 + (c - 1) % 256 if c > 0 else -1 for i, c in enumerate(ctxs)],
                         dtype=torch.int32).to(DEV)
    ct, st = (t.to(DEV) for t in rope_tables(128, 1000000.0, 2048))
    btd, cld = bt.to(DEV), cl.to(DEV)
    maxc = max(max(ctxs), 1)
    res = {}
    for name in ("workspace", "static"):
        kcd, vcd = kc.to(DEV), vc.to(DEV)
        out = torch.full((B, Hq * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
        common = (hip.ptr(qkv), qkv.stride(0), hip.ptr(qw), hip.ptr(kw), 1e-6, hip.ptr(ct), hip.ptr(st), 2048, hip.ptr(pos),
                  hip.ptr(slots), hip.ptr(kcd), hip.ptr(vcd), hip.ptr(btd), 4, hip.ptr(cld), maxc, 128 ** -0.5, hip.ptr(out),
                  Hq * 128, B, Hq, Hkv)
        if name == "workspace":
            ws = torch.zeros(hip.lib.qtts_attn_decode_workspace_bytes(B, Hq, Hkv, maxc), dtype=torch.uint8, device=DEV)
            hip.check(hip.lib.qtts_attention_decode_fused(*common, hip.ptr(ws), hip.stream()), "attention_decode_fused")
        else:
            hip.check(hip.lib.qtts_attention_decode_fused_static(*common, hip.stream()), "attention_decode_fused_static")
        torch.cuda.synchronize()
        res[name] = (out.clone(), kcd, vcd)
    assert bool(torch.isfinite(res["workspace"][0].float()).all())
    for a, b in zip(res["workspace"], res["static"]):
        assert torch.equal(a, b)


def test_static_scratch_entry_point_equals_workspace_entry_point():
    from nano_qwen3tts_vllm import hip
    # the largest size it takes (16 rows x 16 heads, contexts up to 1024: 128- and 256-token splits, merges) ...
    _case(hip, 16, 16, 8, [1024, 1000, 700, 513, 512, 385, 300, 257, 256, 200, 129, 128, 17, 1, 0, 640], seed=1)
    # ... then a smaller one over the same array, which still holds the first call's partials
    _case(hip, 3, 16, 8, [300, 130, 5], seed=2)
    _case(hip, 5, 8, 8, [258, 40, 0, 511, 129], seed=3)


def test_static_scratch_entry_point_rejects_what_does_not_fit():
    from nano_qwen3tts_vllm import hip
    z = torch.zeros(64, dtype=torch.uint8, device=DEV)
    P = hip.ptr(z)
    args = lambda B, Hq, Hkv, maxc, mb: (P, 4096, P, P, 1e-6, P, P, 2048, P, P, P, P, P, mb, P, maxc, 0.1, P, 2048, B, Hq, Hkv,
                                         hip.stream())
    assert hip.lib.qtts_attention_decode_fused_static(*args(17, 16, 8, 128, 4)) == -1        # B * Hq = 272 > 256
    assert b"exceeds the static scratch" in hip.lib.qtts_last_error()
    assert hip.lib.qtts_attention_decode_fused_static(*args(4, 16, 8, 2048, 8)) == -1        # max_context > 1024
    assert b"max_context" in hip.lib.qtts_last_error()
