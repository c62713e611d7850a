"""GPU parity: each HIP operator (called through the C ABI) against the CPU oracle.

Tolerances.  Integer outputs (argmax, slot scatter) are exact.  bf16 outputs of
fp32-accumulated reductions may differ from the oracle by the summation order:
allowed difference is one bf16 ulp of the value (relative 2^-7) on at most 2 % of
the elements; attention (fp32 softmax, parity unpinned upstream) atol 1e-2 / rtol 2e-2.
"""
import numpy as np
import pytest
import torch

import qtts_oracle as O
from helpers import bf, load_npz

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from nano_qwen3tts_vllm import hip as h
    assert torch.cuda.is_available()
    return h


def assert_bf16_close(got, want, max_frac=0.02, what=""):
    got = got.float().cpu()
    want = want.float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    diff = (got - want).abs()
    # one bf16 ulp is 2^-7 relative at the bottom of a binade and 2^-8 at the top; a flip across a binade
    # boundary is one ulp of the LARGER binade, hence the 2^-6 factor on the smaller magnitude
    ulp = torch.maximum(want.abs(), got.abs()) * 2.0 ** -7 * 1.01 + 1e-5 * float(want.abs().mean())
    bad = diff > ulp
    assert not bad.any(), f"{what}: {int(bad.sum())} elements beyond 1 bf16 ulp, max diff {float(diff.max())}"
    frac = float((diff > 0).float().mean())
    assert frac <= max_frac, f"{what}: {frac:.4f} of elements differ (allowed {max_frac})"


@pytest.mark.parametrize("M,N,K", [(1, 512, 256), (3, 256, 128), (8, 4096, 2048), (8, 2048, 6144), (16, 1024, 3072),
                                   (17, 2048, 1024), (40, 512, 512), (8, 3072, 2048)])
def test_linear_plain(hip, M, N, K):
    g = torch.Generator().manual_seed(M * 1000 + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    want = torch.nn.functional.linear(x, w)
    wp = hip.pack_weight(w.to(DEV))
    got = hip.linear(x.to(DEV), wp, N)
    torch.cuda.synchronize()
    assert_bf16_close(got, want, what=f"linear {M}x{N}x{K}")


def test_linear_bias_residual_norm(hip):
    g = torch.Generator().manual_seed(5)
    M, N, K = 6, 1024, 2048
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.03).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    r = torch.randn(M, N, generator=g).bfloat16()
    nw = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16()
    wp = hip.pack_weight(w.to(DEV))
    # bias (the predictor's small_to_mtp_projection, P/models/qwen3_tts_predictor.py:59)
    assert_bf16_close(hip.linear(x.to(DEV), wp, N, bias=b.to(DEV)), torch.nn.functional.linear(x, w, b), what="bias")
    # residual: `residual + linear(x)` (P/models/qwen3_tts_share.py:171)
    assert_bf16_close(hip.linear(x.to(DEV), wp, N, residual=r.to(DEV)), r + torch.nn.functional.linear(x, w), what="residual")
    # fused RMSNorm prologue
    want = torch.nn.functional.linear(O.rms_norm(x, nw, 1e-6), w)
    assert_bf16_close(hip.linear(x.to(DEV), wp, N, norm_w=nw.to(DEV), eps=1e-6), want, what="norm")


@pytest.mark.parametrize("M,I,K", [(2, 256, 128), (8, 6144, 2048), (8, 3072, 1024), (20, 512, 256)])
def test_linear_swiglu(hip, M, I, K):
    g = torch.Generator().manual_seed(I + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    wg = (torch.randn(I, K, generator=g) * 0.05).bfloat16()
    wu = (torch.randn(I, K, generator=g) * 0.05).bfloat16()
    nw = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16()
    h = O.rms_norm(x, nw, 1e-6)
    want = O.silu_mul(torch.nn.functional.linear(h, wg), torch.nn.functional.linear(h, wu))
    wp = hip.pack_gate_up(wg.to(DEV), wu.to(DEV))
    want = want.float()
    for fn in (hip.linear_swiglu,):
        got = fn(x.to(DEV), wp, I, norm_w=nw.to(DEV), eps=1e-6).float().cpu()
        # two chained roundings: a 1-ulp flip of gate or up moves the product by up to 2 ulp
        diff = (got - want).abs()
        assert float((diff > want.abs() * 2.0 ** -6 + 1e-6).float().mean()) == 0.0
        assert float((diff > 0).float().mean()) < 0.03


def test_linear_rejects_bad_shapes(hip):
    x = torch.zeros(2, 100, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(16 * 128, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(hip.QttsError):
        hip.linear(x, w, 16)          # K=100 is not a multiple of 32


def test_rmsnorm_golden(hip):
    g = load_npz("ops.npz")
    got = hip.rmsnorm(bf(g["rms_x"]).to(DEV), bf(g["rms_w"]).to(DEV), 1e-6)
    assert_bf16_close(got, bf(g["rms_y"]), max_frac=0.005, what="rmsnorm")


def _rope_tables(theta, max_pos=2048):
    from nano_qwen3tts_vllm.models.hip_decoder import rope_tables
    c, s = rope_tables(128, theta, max_pos)
    return c.to(DEV), s.to(DEV)


def test_qknorm_rope_store_golden(hip):
    """Fixture vectors produced by the reference's Qwen3TTSRMSNorm + TalkerRotaryEmbedding."""
    g = load_npz("ops.npz")
    pos = torch.from_numpy(g["rope_pos"])
    q, k = bf(g["rope_q"]), bf(g["rope_k"])         # [6,4,128], [6,2,128]
    N, Hq, Hkv = 6, 4, 2
    v = torch.randn(N, Hkv, 128).bfloat16()
    ones = torch.ones(128).bfloat16()
    # norm weights 1 and pre-normalised inputs are not available in the fixture; use the oracle for
    # the norm step (itself pinned by test_oracle_golden) and the fixture for RoPE-only parity below.
    qw = (1 + 0.1 * torch.randn(128)).bfloat16()
    kw = (1 + 0.1 * torch.randn(128)).bfloat16()
    qn, kn = O.rms_norm(q, qw, 1e-6), O.rms_norm(k, kw, 1e-6)
    cos, sin = O.rope_tables(pos, 128, 1000000.0)
    qo, ko = O.apply_rope(qn, kn, cos, sin)
    qkv = torch.cat([q.reshape(N, -1), k.reshape(N, -1), v.reshape(N, -1)], 1).contiguous().to(DEV)
    kc = torch.zeros(4, 256, Hkv, 128, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    slots = torch.tensor([5, 256 + 7, -1, 3 * 256 + 255, 0, 600], dtype=torch.int32)
    ct, st = _rope_tables(1000000.0)
    hip.qknorm_rope_store(qkv, qw.to(DEV), kw.to(DEV), 1e-6, ct, st, pos.to(DEV), slots.to(DEV), kc, vc, Hq, Hkv)
    torch.cuda.synchronize()
    got = qkv.cpu()
    assert_bf16_close(got[:, :Hq * 128].reshape(N, Hq, 128), qo, max_frac=0.01, what="q rope")
    assert_bf16_close(got[:, Hq * 128:(Hq + Hkv) * 128].reshape(N, Hkv, 128), ko, max_frac=0.01, what="k rope")
    kcf, vcf = kc.cpu().view(-1, Hkv, 128), vc.cpu().view(-1, Hkv, 128)
    for i, s in enumerate(slots.tolist()):
        if s < 0:
            continue
        assert torch.equal(kcf[s], got[i, Hq * 128:(Hq + Hkv) * 128].reshape(Hkv, 128))
        assert torch.equal(vcf[s], v[i])
    used = [s for s in slots.tolist() if s >= 0]
    mask = torch.ones(kcf.shape[0], dtype=torch.bool)
    mask[used] = False
    assert float(kcf[mask].abs().sum()) == 0.0 and float(vcf[mask].abs().sum()) == 0.0   # slot -1 skipped, nothing else touched

    # RoPE alone against the reference's own outputs: unit norm weights and inputs whose RMS is exactly 1
    # are not constructible in bf16, so compare through the oracle path above and, separately, check the
    # fixture's reference outputs with the oracle tables (bit-exact on CPU in test_oracle_golden).
    qo_ref, ko_ref = O.apply_rope(q, k, cos, sin)
    assert np.array_equal(qo_ref.float().numpy(), g["mrope_qo"]) and np.array_equal(ko_ref.float().numpy(), g["mrope_ko"])


def test_store_kvcache(hip):
    N, Hkv = 7, 8
    k = torch.randn(N, Hkv, 128).bfloat16().to(DEV)
    v = torch.randn(N, Hkv, 128).bfloat16().to(DEV)
    kc = torch.zeros(3, 256, Hkv, 128, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    slots = torch.tensor([0, 1, -1, 300, 767, -1, 256], dtype=torch.int32, device=DEV)
    hip.store_kvcache(k, v, kc, vc, slots)
    kco = torch.zeros(3, 256, Hkv, 128, dtype=torch.bfloat16)
    vco = torch.zeros_like(kco)
    O.store_kvcache(k.cpu(), v.cpu(), kco, vco, slots.cpu())
    assert torch.equal(kc.cpu(), kco) and torch.equal(vc.cpu(), vco)


@pytest.mark.parametrize("Hq,Hkv,ctxs", [(16, 8, [1, 17, 128, 129, 300, 1024]), (4, 2, [3, 260, 0, 511]), (16, 8, [2, 3, 16])])
def test_paged_attention_decode(hip, Hq, Hkv, ctxs):
    g = torch.Generator().manual_seed(sum(ctxs))
    B = len(ctxs)
    nblk = 4 * B + 1
    kc = torch.randn(nblk, 256, Hkv, 128, generator=g).bfloat16()
    vc = torch.randn(nblk, 256, Hkv, 128, generator=g).bfloat16()
    q = torch.randn(B, Hq, 128, generator=g).bfloat16()
    perm = torch.randperm(nblk - 1, generator=g) + 1          # scattered block ids
    bt = perm[:4 * B].reshape(B, 4).to(torch.int32)
    cl = torch.tensor(ctxs, dtype=torch.int32)
    want = O.attention_decode(q, kc, vc, bt, cl, 128 ** -0.5)
    maxc = max(max(ctxs), 1)
    got = hip.paged_attention(q.reshape(B, -1).to(DEV), kc.to(DEV), vc.to(DEV), bt.to(DEV), cl.to(DEV), maxc,
                              128 ** -0.5, Hq, Hkv)
    torch.cuda.synchronize()
    assert torch.allclose(got.float().cpu().view(B, Hq, 128), want.float(), atol=1e-2, rtol=2e-2)
    for i, c in enumerate(ctxs):
        if c == 0:
            assert float(got[i].abs().sum()) == 0.0


@pytest.mark.parametrize("Hq,Hkv,lens", [(4, 2, [5, 140]), (16, 8, [200]), (2, 2, [33, 17, 300]), (4, 2, [16, 16, 7, 1, 1, 30, 2]),
                                         (16, 8, [600]), (4, 2, [12]), (8, 4, [257, 31]), (4, 2, [16, 16, 7, 1, 1, 30, 2, 45, 3]),
                                         (16, 8, [48]), (8, 8, [100, 11]), (4, 1, [130])])
def test_paged_attention_prefill_rows(hip, Hq, Hkv, lens):
    """Causal prefill expressed as per-token rows over the just-stored cache equals
    the packed varlen formulation (P/layers/attention.py:88-91): single and several prompts per launch, row tiles
    that straddle prompts, contexts over several KV blocks.  From 112 rows a launch takes the MFMA kernel (both of its
    head ratios and ragged packed prompts are among the cases), below that and with four query heads per KV head the
    per-row kernel."""
    g = torch.Generator().manual_seed(3 + sum(lens))
    N = sum(lens)
    q = torch.randn(N, Hq, 128, generator=g).bfloat16()
    k = torch.randn(N, Hkv, 128, generator=g).bfloat16()
    v = torch.randn(N, Hkv, 128, generator=g).bfloat16()
    cu = [0]
    for L in lens:
        cu.append(cu[-1] + L)
    want = O.attention_prefill(q, k, v, cu, 128 ** -0.5)
    mb = (max(lens) + 255) // 256
    nblk = mb * len(lens) + 1
    kc = torch.zeros(nblk, 256, Hkv, 128, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    bt = (torch.randperm(nblk - 1, generator=g) + 1)[:mb * len(lens)].reshape(len(lens), mb).to(torch.int32)
    slots, row_seq, ctx = [], [], []
    for s, L in enumerate(lens):
        for t in range(L):
            slots.append(int(bt[s, t // 256]) * 256 + t % 256)
            row_seq.append(s)
            ctx.append(t + 1)
    O.store_kvcache(k, v, kc, vc, torch.tensor(slots))
    got = hip.paged_attention(q.reshape(N, -1).to(DEV), kc.to(DEV), vc.to(DEV), bt.to(DEV),
                              torch.tensor(ctx, dtype=torch.int32, device=DEV), max(lens), 128 ** -0.5, Hq, Hkv,
                              row_seq=torch.tensor(row_seq, dtype=torch.int32, device=DEV))
    torch.cuda.synchronize()
    assert torch.allclose(got.float().cpu().view(N, Hq, 128), want.float(), atol=1e-2, rtol=2e-2)


def test_argmax_ties_lowest_index(hip):
    lg = torch.randn(5, 3072).bfloat16()
    lg[0, 100] = 50.0
    lg[0, 2000] = 50.0            # exact tie -> index 100
    lg[1, 3071] = 60.0
    lg[2, 0] = 70.0
    got = hip.argmax(lg.to(DEV)).cpu()
    assert torch.equal(got, O.greedy(lg))
    assert got[0] == 100 and got[1] == 3071 and got[2] == 0


def test_sample_topk_distribution(hip):
    """Draws stay inside the reference's top-k support and follow its probabilities
    (P/layers/sampler.py:15-22).  The RNG stream differs from torch's: statistical check only."""
    g = torch.Generator().manual_seed(9)
    V, n = 2048, 4000
    row = (torch.randn(V, generator=g) * 2).bfloat16()
    logits = row.unsqueeze(0).repeat(n, 1).contiguous()
    temps = torch.full((n,), 0.9)
    probs = O.topk_probs(row.unsqueeze(0), temps[:1], 50)[0]
    got = hip.sample_topk(logits.to(DEV), temps.to(DEV), 50, seed=123, offset=0).cpu()
    support = set(torch.nonzero(probs > 0).flatten().tolist())
    assert len(support) >= 50 and set(got.tolist()) <= support
    freq = torch.bincount(got, minlength=V).float() / n
    assert float((freq - probs).abs().max()) < 0.03
    again = hip.sample_topk(logits.to(DEV), temps.to(DEV), 50, seed=123, offset=0).cpu()
    assert torch.equal(again, got)                     # counter-based: reproducible
    other = hip.sample_topk(logits.to(DEV), temps.to(DEV), 50, seed=123, offset=1).cpu()
    assert not torch.equal(other, got)


@pytest.mark.parametrize("V,k", [(3072, 50), (2048, 50), (512, 7), (4096, 1), (100, 100)])
def test_sample_topk_support_rows(hip, V, k):
    """Different rows, coarse logits with many ties at the k-th value: every draw lies in the reference's support
    (all values >= the k-th largest stay), and with several hundred draws per row every row's draws cover more than
    one index when the support has more than one."""
    g = torch.Generator().manual_seed(V + k)
    rows = 24
    base = (torch.randn(rows, V, generator=g) * 1.5).mul(4).round().div(4).bfloat16()       # quarter steps: many ties
    temps = torch.linspace(0.6, 1.3, rows)
    probs = O.topk_probs(base, temps, k)
    rep = 200
    logits = base.repeat_interleave(rep, 0).contiguous()
    got = hip.sample_topk(logits.to(DEV), temps.repeat_interleave(rep).to(DEV), k, seed=7, offset=3).cpu().view(rows, rep)
    for r in range(rows):
        support = set(torch.nonzero(probs[r] > 0).flatten().tolist())
        assert len(support) >= k
        drawn = set(got[r].tolist())
        assert drawn <= support, (r, sorted(drawn - support)[:5])
        if k > 1:
            assert len(drawn) > 1
        else:
            assert drawn == {int(torch.argmax(base[r].float()))}


def test_embedding_gather(hip):
    table = torch.randn(300, 256).bfloat16()
    ids = torch.tensor([0, 299, 5, 5], dtype=torch.int64)
    got = hip.embedding_gather(table.to(DEV), ids.to(DEV)).cpu()
    assert torch.equal(got, table[ids])


def test_linear_fused_variants(hip):
    """The decoder stack's GEMM: gathered rows + bias, RMSNorm prologue with the normalised rows written
    out, SwiGLU applied on the way in, residual epilogue — each against the oracle's op sequence."""
    g = torch.Generator().manual_seed(77)
    M, H, I, V = 8, 1024, 3072, 512
    table = torch.randn(V, H, generator=g).bfloat16()
    ids = torch.tensor([3, 500, 0, 511, 7, 7, 100, 42], dtype=torch.int64)
    w = (torch.randn(256, H, generator=g) * 0.05).bfloat16()
    b = torch.randn(256, generator=g).bfloat16()
    wp = hip.pack_weight(w.to(DEV))
    got = hip.linear_fused(table.to(DEV), wp, 256, gather_ids=ids.to(DEV), bias=b.to(DEV))
    assert_bf16_close(got, torch.nn.functional.linear(table[ids], w, b), what="gather+bias")

    x = torch.randn(M, H, generator=g).bfloat16()
    nw = (1 + 0.1 * torch.randn(H, generator=g)).bfloat16()
    xn_want = O.rms_norm(x, nw, 1e-6)
    xn = torch.zeros(M, H, dtype=torch.bfloat16, device=DEV)
    got = hip.linear_fused(x.to(DEV), wp, 256, norm_w=nw.to(DEV), xnorm_out=xn)
    assert_bf16_close(xn, xn_want, max_frac=0.005, what="xnorm_out")
    assert_bf16_close(got, torch.nn.functional.linear(xn_want, w), what="norm+linear")

    wg = (torch.randn(I, H, generator=g) * 0.05).bfloat16()
    wu = (torch.randn(I, H, generator=g) * 0.05).bfloat16()
    wd = (torch.randn(H, I, generator=g) * 0.03).bfloat16()
    gu = hip.linear_fused(x.to(DEV), hip.pack_weight(torch.cat([wg, wu], 0).to(DEV)), 2 * I, norm_w=nw.to(DEV))
    gu_want = torch.cat([torch.nn.functional.linear(xn_want, wg), torch.nn.functional.linear(xn_want, wu)], 1)
    assert_bf16_close(gu, gu_want, what="gate|up")
    # feed the oracle's own gate|up so that only the fused activation + down GEMM + residual are compared
    got = hip.linear_fused(gu_want.to(DEV), hip.pack_weight(wd.to(DEV)), H, swiglu_input=True, residual=x.to(DEV))
    act = O.silu_mul(gu_want[:, :I], gu_want[:, I:])
    assert_bf16_close(got, x + torch.nn.functional.linear(act, wd), what="swiglu-in + down + residual")


@pytest.mark.parametrize("Hq,Hkv,ctxs", [(16, 8, [1, 17, 128, 257, 300, 1024, 0, 512]), (4, 2, [3, 260, 0, 511]), (16, 8, [2, 3, 16]),
                                         (16, 8, [200, 256, 130, 1]), (8, 8, [250, 140, 129]), (8, 8, [1000] * 5 + [700]),
                                         (16, 8, [300, 384, 129, 7]), (8, 8, [385, 512])])
def test_attention_decode_fused(hip, Hq, Hkv, ctxs):
    """q/k norm + RoPE + KV store + attention in one launch == the oracle's separate ops."""
    g = torch.Generator().manual_seed(sum(ctxs) + 1)
    B = len(ctxs)
    nblk = 4 * B + 1
    kc = torch.randn(nblk, 256, Hkv, 128, generator=g).bfloat16()
    vc = torch.randn(nblk, 256, Hkv, 128, generator=g).bfloat16()
    qkv = torch.randn(B, (Hq + 2 * Hkv) * 128, generator=g).bfloat16()
    qw = (1 + 0.1 * torch.randn(128, generator=g)).bfloat16()
    kw = (1 + 0.1 * torch.randn(128, generator=g)).bfloat16()
    perm = torch.randperm(nblk - 1, generator=g) + 1
    bt = perm[:4 * B].reshape(B, 4).to(torch.int32)
    cl = torch.tensor(ctxs, dtype=torch.int32)
    pos = torch.tensor([max(c, 1) for c in ctxs], dtype=torch.int64)           # talker convention: position = len
    slots = torch.tensor([int(bt[i, (c - 1) // 256]) * 256 + (c - 1) % 256 if c > 0 else -1 for i, c in enumerate(ctxs)],
                         dtype=torch.int32)
    # oracle
    q = qkv[:, :Hq * 128].view(B, Hq, 128)
    k = qkv[:, Hq * 128:(Hq + Hkv) * 128].view(B, Hkv, 128)
    v = qkv[:, (Hq + Hkv) * 128:].view(B, Hkv, 128)
    cos, sin = O.rope_tables(pos, 128, 1000000.0)
    qo, ko = O.apply_rope(O.rms_norm(q, qw, 1e-6), O.rms_norm(k, kw, 1e-6), cos, sin)
    kco, vco = kc.clone(), vc.clone()
    O.store_kvcache(ko, v, kco, vco, slots)
    want = O.attention_decode(qo, kco, vco, bt, cl, 128 ** -0.5)
    # HIP
    from nano_qwen3tts_vllm.models.hip_decoder import rope_tables
    ct, st = rope_tables(128, 1000000.0, 2048)
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    out = torch.empty(B, Hq * 128, dtype=torch.bfloat16, device=DEV)
    maxc = max(max(ctxs), 1)
    # scratch handed over dirty: the call itself clears the arrival counters it depends on
    ws = torch.full((hip.lib.qtts_attn_decode_workspace_bytes(B, Hq, Hkv, maxc),), 0xAB, dtype=torch.uint8, device=DEV)
    qkvd, btd, cld = qkv.to(DEV), bt.to(DEV), cl.to(DEV)
    qwd, kwd, ctd, std, posd, slotsd = qw.to(DEV), kw.to(DEV), ct.to(DEV), st.to(DEV), pos.to(DEV), slots.to(DEV)   # keep alive
    hip.check(hip.lib.qtts_attention_decode_fused(hip.ptr(qkvd), qkvd.stride(0), hip.ptr(qwd), hip.ptr(kwd),
                                                  1e-6, hip.ptr(ctd), hip.ptr(std), 2048, hip.ptr(posd),
                                                  hip.ptr(slotsd), hip.ptr(kcd), hip.ptr(vcd), hip.ptr(btd), 4,
                                                  hip.ptr(cld), maxc, 128 ** -0.5, hip.ptr(out), Hq * 128, B, Hq, Hkv,
                                                  hip.ptr(ws), hip.stream()), "attention_decode_fused")
    torch.cuda.synchronize()
    assert torch.allclose(out.float().cpu().view(B, Hq, 128), want.float(), atol=1e-2, rtol=2e-2)
    # a second launch over the same scratch with other queries: the partials of the first launch, which other
    # compute dies may still hold in their caches, must not leak into the merge
    qkv2 = qkv.clone()
    qkv2[:, :Hq * 128] = torch.randn(B, Hq * 128, generator=g).bfloat16()
    q2o, _ = O.apply_rope(O.rms_norm(qkv2[:, :Hq * 128].view(B, Hq, 128), qw, 1e-6), O.rms_norm(k, kw, 1e-6), cos, sin)
    want2 = O.attention_decode(q2o, kco, vco, bt, cl, 128 ** -0.5)
    qkv2d = qkv2.to(DEV)
    out2 = torch.empty_like(out)
    hip.check(hip.lib.qtts_attention_decode_fused(hip.ptr(qkv2d), qkv2d.stride(0), hip.ptr(qwd), hip.ptr(kwd),
                                                  1e-6, hip.ptr(ctd), hip.ptr(std), 2048, hip.ptr(posd),
                                                  hip.ptr(slotsd), hip.ptr(kcd), hip.ptr(vcd), hip.ptr(btd), 4,
                                                  hip.ptr(cld), maxc, 128 ** -0.5, hip.ptr(out2), Hq * 128, B, Hq, Hkv,
                                                  hip.ptr(ws), hip.stream()), "attention_decode_fused")
    torch.cuda.synchronize()
    assert torch.allclose(out2.float().cpu().view(B, Hq, 128), want2.float(), atol=1e-2, rtol=2e-2)
    # the cache received exactly the oracle's new K/V rows (K up to 1-ulp norm differences) and nothing else changed
    kdiff = (kcd.cpu().float() - kco.float()).abs()
    assert float(kdiff.max()) <= 0.0625 and float((kdiff > 0).float().mean()) < 1e-4
    assert torch.equal(vcd.cpu(), vco)


@pytest.mark.parametrize("M,N,K,kind", [(17, 256, 128, "norm"), (40, 4096, 2048, "norm"), (200, 2048, 2048, "res"),
                                        (200, 2048, 6144, "res"), (130, 1024, 3072, "bias"), (200, 6144, 1024, "swiglu"),
                                        (33, 3072, 2048, "head"), (200, 6144, 1024, "swiglu_plain"), (40, 12288, 2048, "swiglu_plain"),
                                        (250, 4096, 2048, "plain"), (300, 512, 256, "plain"), (24, 256, 128, "res"),
                                        (20, 2048, 2048, "plain"), (32, 6144, 1024, "swiglu_plain"), (64, 1024, 1024, "res"),
                                        (200, 2048, 6144, "ws_res"), (35, 2048, 6144, "ws_bias"), (100, 1024, 3072, "ws_res"),
                                        (1000, 2048, 2048, "res"), (24, 4096, 2048, "norm"), (32, 12288, 2048, "swiglu"),
                                        (64, 4096, 1024, "norm"), (130, 6144, 1024, "swiglu"), (20, 1024, 1024, "norm"),
                                        # row groups of three tiles (K = 2048, 3 / 6 / 9 row tiles) and of two (narrow outputs)
                                        (90, 4096, 2048, "res"), (44, 12288, 2048, "swiglu_plain"), (140, 12288, 2048, "swiglu_plain"),
                                        (48, 2048, 2048, "res"), (130, 2048, 2048, "res")])
def test_linear_prefill_row_counts(hip, M, N, K, kind):
    """Prefill-sized row counts (several 16-row tiles, ragged last tile) through every GEMM variant of the
    decoder stack, against the oracle."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    nw = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16().to(DEV)
    res = torch.randn(M, N, generator=g).bfloat16().to(DEV)
    bias = torch.randn(N, generator=g).bfloat16().to(DEV)
    xd = x.to(DEV)

    def run():
        if kind.startswith("swiglu"):
            wg = (torch.randn(N // 2, K, generator=torch.Generator().manual_seed(1)) * 0.05).bfloat16()
            wu = (torch.randn(N // 2, K, generator=torch.Generator().manual_seed(2)) * 0.05).bfloat16()
            return hip.linear_swiglu(xd, hip.pack_gate_up(wg.to(DEV), wu.to(DEV)), N // 2,
                                     norm_w=nw if kind == "swiglu" else None), (wg, wu)
        w = (torch.randn(N, K, generator=torch.Generator().manual_seed(3)) * 0.05).bfloat16()
        wp = hip.pack_weight(w.to(DEV))
        if kind == "plain":
            return hip.linear(xd, wp, N), w
        if kind == "norm":
            return hip.linear(xd, wp, N, norm_w=nw), w
        if kind == "head":
            xn = torch.zeros(M, K, dtype=torch.bfloat16, device=DEV)
            out = hip.linear_fused(xd, wp, N, norm_w=nw, xnorm_out=xn)
            return torch.cat([out, xn], 1), w
        if kind == "res":
            return hip.linear(xd, wp, N, residual=res), w
        if kind == "ws_res":
            return hip.linear_ws(xd, wp, N, residual=res), w
        if kind == "ws_bias":
            return hip.linear_ws(xd, wp, N, bias=bias, residual=res), w
        return hip.linear(xd, wp, N, bias=bias, residual=res), w

    got, w = run()
    xf = x
    if kind in ("norm", "head", "swiglu"):
        xf = O.rms_norm(x, nw.cpu(), 1e-6)
    if kind.startswith("swiglu"):
        want = O.silu_mul(torch.nn.functional.linear(xf, w[0]), torch.nn.functional.linear(xf, w[1])).float()
        diff = (got.float().cpu() - want).abs()
        # two chained roundings: a 1-ulp flip of gate or up moves the product by about 2 ulp; for gates
        # around -2..-6 silu amplifies the relative error of the rounded gate up to 3x, hence the rare tail
        assert float((diff > want.abs() * 2.0 ** -6 + 1e-6).float().mean()) < 1e-3
        assert float((diff > want.abs() * 2.0 ** -4 + 1e-6).float().mean()) == 0.0
        assert float((diff > 0).float().mean()) < 0.03
        return
    elif kind == "head":
        want = torch.cat([torch.nn.functional.linear(xf, w), xf], 1)
    else:
        want = torch.nn.functional.linear(xf, w, bias.cpu() if kind in ("bias", "ws_bias") else None)
        if kind in ("res", "bias", "ws_res", "ws_bias"):
            # the rounding that may differ by one ulp happens before the residual add, at the GEMM output's
            # magnitude: compare there (the add itself is exact to its own rounding in both)
            lin = want.float()
            want = want + res.cpu()
            diff = (got.float().cpu() - want.float()).abs()
            ulp = (lin.abs() + want.float().abs()) * 2.0 ** -7 * 1.01 + 1e-5     # one ulp at each of the two roundings
            assert not (diff > ulp).any(), f"rows {kind}: max diff {float(diff.max())}"
            assert float((diff > 0).float().mean()) <= 0.02
            return
    assert_bf16_close(got, want, what=f"rows {kind}")


def _fp8_unpack(packed: torch.Tensor, N: int, K: int) -> torch.Tensor:
    """Bytes of the fragment layout back to row-major [N, K] (uint8)."""
    b = packed.cpu().view(N // 16, K // 64, 4, 16, 2, 8)             # tile, j, g, r, half, e
    return b.permute(0, 3, 1, 4, 2, 5).reshape(N, K)                  # tile, r, j, half, g, e


def test_fp8_pack_matches_torch_quantisation(hip):
    """Device packing == the oracle's quantisation (e4m3 OCP encoding, round to nearest even, power-of-two row
    scales), including an all-zero row, a row whose maximum is exactly 448 * 2^k and tiny values."""
    g = torch.Generator().manual_seed(5)
    N, K = 64, 512
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    w[3] = 0
    w[5, 7] = 448.0 * 2.0 ** -6
    w[5] = w[5].clamp(-7.0, 7.0)
    w[9] *= 1e-3
    w[11] *= 300.0
    q, scale, deq = O.quantize_fp8_rows(w)
    packed, scales = hip.pack_weight_fp8(w.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(scales.cpu(), scale)
    assert torch.equal(_fp8_unpack(packed, N, K), q.view(torch.uint8))
    # gate/up interleave
    gate, up = w[:32].contiguous(), w[32:].contiguous()
    pgu, sgu = hip.pack_gate_up_fp8(gate.to(DEV), up.to(DEV))
    rows = torch.cat([torch.cat([gate[8 * t:8 * t + 8], up[8 * t:8 * t + 8]]) for t in range(4)])
    q2, s2, _ = O.quantize_fp8_rows(rows)
    assert torch.equal(sgu.cpu(), s2) and torch.equal(_fp8_unpack(pgu, 64, K), q2.view(torch.uint8))


@pytest.mark.parametrize("M,N,K,kind", [(8, 4096, 2048, "plain"), (8, 2048, 2048, "res"), (3, 1024, 3072, "bias"), (8, 2048, 6144, "res"),
                                        (16, 12288, 2048, "norm"), (8, 4096, 1024, "norm"), (1, 512, 512, "plain"),
                                        (8, 6144, 2048, "swiglu"), (8, 3072, 1024, "swiglu"), (5, 12288, 2048, "swiglu_plain"),
                                        (200, 4096, 2048, "plain"), (40, 2048, 2048, "res"), (200, 2048, 6144, "res"),
                                        (130, 1024, 3072, "bias"), (200, 6144, 2048, "swiglu_plain"), (32, 6144, 1024, "swiglu_plain"),
                                        (128, 12288, 2048, "swiglu_plain"), (24, 4096, 2048, "norm"), (100, 6144, 1024, "swiglu"),
                                        (64, 4096, 1024, "norm"), (90, 4096, 2048, "res"), (44, 12288, 2048, "swiglu_plain"),
                                        (140, 12288, 2048, "swiglu_plain"), (300, 4096, 2048, "res")])
def test_linear_fp8_equals_bf16_on_dequantised_weights(hip, M, N, K, kind):
    """Every GEMM variant over fp8 weights returns bit for bit what the bf16 kernel returns on the dequantised
    weights q * scale (which are bf16 numbers), and the dequantised weights are the oracle's."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16().to(DEV)
    res = torch.randn(M, N, generator=g).bfloat16().to(DEV)
    bias = torch.randn(N, generator=g).bfloat16().to(DEV)
    if kind.startswith("swiglu"):
        I = N // 2
        wg = (torch.randn(I, K, generator=g) * 0.05).bfloat16()
        wu = (torch.randn(I, K, generator=g) * 0.05).bfloat16()
        pq, sq = hip.pack_gate_up_fp8(wg.to(DEV), wu.to(DEV))
        dg, du = O.quantize_fp8_rows(wg)[2], O.quantize_fp8_rows(wu)[2]
        norm = nw if kind == "swiglu" else None
        got = hip.linear_swiglu_fp8(x, pq, sq, I, norm_w=norm)
        want = hip.linear_swiglu(x, hip.pack_gate_up(dg.to(DEV), du.to(DEV)), I, norm_w=norm)
    else:
        w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
        pq, sq = hip.pack_weight_fp8(w.to(DEV))
        wp = hip.pack_weight(O.quantize_fp8_rows(w)[2].to(DEV))
        kw = dict(bias=bias if kind == "bias" else None, residual=res if kind in ("res", "bias") else None,
                  norm_w=nw if kind == "norm" else None)
        got = hip.linear_fp8(x, pq, sq, N, **kw)
        if kind != "norm" and M > 16:
            want = hip.linear_ws(x, wp, N, bias=kw["bias"], residual=kw["residual"])
        else:
            want = hip.linear(x, wp, N, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


@pytest.mark.parametrize("M,N,K", [(17, 4096, 2048), (24, 2048, 6144), (32, 2048, 1024), (33, 2048, 2048), (48, 4096, 1024), (40, 1024, 3072), (64, 2048, 6144), (100, 1024, 3072),
                                   (96, 4096, 2048), (200, 4096, 2048), (600, 2048, 6144)])
def test_prompt_gemm_row_group_forms(hip, M, N, K):
    """From 17 rows a GEMM takes the prompt kernel: row groups of two, three or four 16-row tiles (chosen per shape; two
    for K = 3072, twelve k-steps per wave), K = 6144 as K parts through the workspace; partial last tiles and groups."""
    g = torch.Generator().manual_seed(M * N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    wp = hip.pack_weight(w.to(DEV))
    want = torch.nn.functional.linear(x, w)
    got = hip.linear_ws(x.to(DEV), wp, N)
    torch.cuda.synchronize()
    assert_bf16_close(got, want, what=f"prompt GEMM {M}x{N}x{K}")


@pytest.mark.parametrize("N,K,kind", [(4096, 1024, "norm"), (4096, 2048, "norm"), (2048, 2048, "res"), (1024, 3072, "res"),
                                      (2048, 6144, "res"), (1024, 2048, "plain"), (6144, 1024, "swiglu"), (12288, 2048, "swiglu")])
def test_rows_do_not_depend_on_the_row_count_form(hip, N, K, kind):
    """Up to 8 rows the decode GEMM lets the lanes of the unused MFMA columns share the prologue (half the activation
    requests and half the normalisation per lane); with 9..16 rows every lane does all of it.  Both forms must return
    the same bits for a row — the predictor's input tables are built 16 rows per launch and read by 8-row steps."""
    g = torch.Generator().manual_seed(N + K)
    x = (torch.randn(16, K, generator=g) * 0.7).bfloat16().to(DEV)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16().to(DEV)
    if kind == "swiglu":
        I = N // 2
        wp = hip.pack_gate_up((torch.randn(I, K, generator=g) * 0.05).bfloat16().to(DEV),
                              (torch.randn(I, K, generator=g) * 0.05).bfloat16().to(DEV))
        run = lambda rows: hip.linear_swiglu(x[:rows].contiguous(), wp, I, norm_w=nw)
    else:
        wp = hip.pack_weight((torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV))
        res = torch.randn(16, N, generator=g).bfloat16().to(DEV)
        run = lambda rows: hip.linear(x[:rows].contiguous(), wp, N, norm_w=nw if kind == "norm" else None,
                                      residual=res[:rows].contiguous() if kind == "res" else None)
    full = run(16)
    for rows in (8, 5, 1):
        part = run(rows)
        torch.cuda.synchronize()
        assert torch.equal(part, full[:rows]), f"{kind} {N}x{K}: {rows} rows differ from the 16-row launch"


# ---------------------------------------------------------------- per-frame glue kept on the device
def _glue_tables(V=512, Vp=256, H=256, seed=3):
    g = torch.Generator().manual_seed(seed)
    tw = {"model.codec_embedding.weight": (torch.randn(V, H, generator=g) * 0.05).bfloat16()}
    pw = {f"model.codec_embedding.{i}.weight": (torch.randn(Vp, H, generator=g) * 0.05).bfloat16() for i in range(15)}
    return tw, pw, g


@pytest.mark.parametrize("B,H", [(1, 256), (5, 256), (8, 2048), (16, 1024)])
def test_frame_glue_bit_exact(hip, B, H):
    """qtts_frame_glue against the oracle's glue (P/interface.py:1276-1285: the 16 codebook embeddings summed in fp32,
    plus the frame's text / pad hidden, rounded to bf16 once when the talker takes it): bit for bit, with the ids
    laid out as the burst leaves them (step-major, leading dimension larger than B), per-row frame indices into the
    text table, and the 16 ids written out per row."""
    V, Vp, F, ldp = 512, 256, 7, B + 3
    tw, pw, g = _glue_tables(V, Vp, H, seed=B + H)
    tok0 = torch.randint(0, V, (B,), generator=g)
    pred = torch.randint(0, Vp, (15, ldp), generator=g)
    text = torch.randn(B, F, H, generator=g) * 0.3
    fidx = torch.randint(0, F, (B,), generator=g).to(torch.int32)
    want = torch.stack([O.frame_glue(tw, pw, int(tok0[b]), [int(pred[c, b]) for c in range(15)], text[b, int(fidx[b])])
                        for b in range(B)]).to(torch.bfloat16)
    temb = tw["model.codec_embedding.weight"].to(DEV)
    tabs = [pw[f"model.codec_embedding.{i}.weight"].to(DEV) for i in range(15)]
    ptrs = torch.tensor([t.data_ptr() for t in tabs], dtype=torch.int64, device=DEV)
    nxt = torch.zeros(B, H, dtype=torch.bfloat16, device=DEV)
    codes = torch.full((B, 16), -1, dtype=torch.int64, device=DEV)
    tok0d, predd, textd, fidxd = tok0.to(DEV), pred.to(DEV).contiguous(), text.to(DEV).contiguous(), fidx.to(DEV)
    hip.check(hip.lib.qtts_frame_glue(hip.ptr(temb), hip.ptr(ptrs), hip.ptr(tok0d), hip.ptr(predd), ldp, hip.ptr(textd),
                                      F * H, hip.ptr(fidxd), hip.ptr(nxt), hip.ptr(codes), B, H, V, Vp, hip.stream()), "frame_glue")
    torch.cuda.synchronize()
    assert torch.equal(nxt.cpu(), want)
    assert torch.equal(codes.cpu(), torch.cat([tok0[:, None], pred[:, :B].t()], 1))
    # without frame indices the first text row is used
    hip.check(hip.lib.qtts_frame_glue(hip.ptr(temb), hip.ptr(ptrs), hip.ptr(tok0d), hip.ptr(predd), ldp, hip.ptr(textd),
                                      F * H, None, hip.ptr(nxt), None, B, H, V, Vp, hip.stream()), "frame_glue")
    torch.cuda.synchronize()
    want0 = torch.stack([O.frame_glue(tw, pw, int(tok0[b]), [int(pred[c, b]) for c in range(15)], text[b, 0])
                         for b in range(B)]).to(torch.bfloat16)
    assert torch.equal(nxt.cpu(), want0)


def test_predictor_input_rows(hip):
    """qtts_predictor_input: row pair b = [talker hidden state of b ; codec_embedding(first code of b)]
    (P/interface.py:1242-1252), bit for bit."""
    B, H, V = 6, 256, 512
    tw, _, g = _glue_tables(V, 256, H, seed=9)
    hidden = torch.randn(B, H, generator=g).bfloat16()
    tok0 = torch.randint(0, V, (B,), generator=g)
    want = torch.stack([torch.stack([hidden[b], tw["model.codec_embedding.weight"][tok0[b]]]) for b in range(B)])
    out = torch.zeros(B, 2, H, dtype=torch.bfloat16, device=DEV)
    hd, td, ed = hidden.to(DEV), tok0.to(DEV), tw["model.codec_embedding.weight"].to(DEV)
    hip.check(hip.lib.qtts_predictor_input(hip.ptr(hd), hip.ptr(ed), hip.ptr(td), hip.ptr(out), B, H, V, hip.stream()), "predictor_input")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), want)


def _host_decode_metadata(n, table):
    """What the reference's prepare_decode_talker builds for a sequence of n tokens whose blocks are `table`
    (P/engine/model_runner/talker_mode_runner.py:116-140 with P/engine/sequence.py:47-58): position = len(seq),
    context = len(seq), slot = block_table[-1] * 256 + last_block_num_tokens - 1."""
    num_blocks = (n + 255) // 256
    last_block_num_tokens = n - (num_blocks - 1) * 256
    return n, n, table[num_blocks - 1] * 256 + last_block_num_tokens - 1


def test_advance_decode_follows_prepare_decode_talker(hip):
    """qtts_advance_decode, called once per frame, walks the rows' metadata exactly as the reference rebuilds it on
    the host for a sequence that grew by one token: across 256-token block boundaries, inactive rows (context 0)
    untouched, rows frozen at max_context (1024) and at their own limit, frame index and RNG counter advanced."""
    mb, B = 4, 8
    g = torch.Generator().manual_seed(1)
    tables = torch.stack([torch.randperm(40, generator=g)[:mb] for _ in range(B)]).to(torch.int32)
    start = [0, 5, 254, 255, 510, 1021, 1023, 300]
    limits = [1024, 1024, 1024, 1024, 1024, 1024, 1024, 303]
    ctx = torch.tensor(start, dtype=torch.int32)
    pos = ctx.to(torch.int64).clone()                  # the frame engine starts a row at (L, L) before the first advance
    slots = torch.full((B,), -1, dtype=torch.int32)
    fidx = torch.zeros(B, dtype=torch.int32)
    d = dict(pos=pos.to(DEV), slots=slots.to(DEV), ctx=ctx.to(DEV), tab=tables.to(DEV).contiguous(), fidx=fidx.to(DEV),
             rng=torch.zeros(1, dtype=torch.int64, device=DEV), lim=torch.tensor(limits, dtype=torch.int32, device=DEV))
    n = list(start)
    frames = [0] * B
    for it in range(6):
        hip.check(hip.lib.qtts_advance_decode(hip.ptr(d["pos"]), hip.ptr(d["slots"]), hip.ptr(d["ctx"]), hip.ptr(d["tab"]), mb,
                                              hip.ptr(d["fidx"]), hip.ptr(d["rng"]), hip.ptr(d["lim"]), 1024, B, hip.stream()),
                  "advance_decode")
        torch.cuda.synchronize()
        for b in range(B):
            if n[b] > 0 and n[b] < min(limits[b], 1024):
                n[b] += 1
                frames[b] += 1
        for b in range(B):
            if n[b] == 0:
                assert int(d["ctx"][b]) == 0 and int(d["slots"][b]) == -1 and int(d["fidx"][b]) == 0
                continue
            if frames[b] == 0:
                continue                              # frozen from the start (context already at 1024 - no such row here)
            wp, wc, ws = _host_decode_metadata(n[b], tables[b].tolist())
            assert (int(d["pos"][b]), int(d["ctx"][b]), int(d["slots"][b])) == (wp, wc, ws), (it, b)
            assert int(d["fidx"][b]) == frames[b]
        assert int(d["rng"][0]) == 64 * (it + 1)
    assert n == [0, 11, 260, 261, 516, 1024, 1024, 303]           # rows 5, 6 froze at 1024, row 7 at its own limit
    # a null limit array means max_context alone
    hip.check(hip.lib.qtts_advance_decode(hip.ptr(d["pos"]), hip.ptr(d["slots"]), hip.ptr(d["ctx"]), hip.ptr(d["tab"]), mb,
                                          hip.ptr(d["fidx"]), None, None, 1024, B, hip.stream()), "advance_decode")
    torch.cuda.synchronize()
    assert int(d["ctx"][7]) == 304 and int(d["ctx"][6]) == 1024 and int(d["rng"][0]) == 64 * 6


def test_decode_attention_with_a_context_beyond_the_launch(hip):
    """A row whose device-side context length exceeds the max_context the host launched for is cut to it: its output
    is written, and the arrival counters of the split merge are back at zero for the next launch."""
    Hq, Hkv, B, maxc = 16, 8, 3, 256
    g = torch.Generator().manual_seed(12)
    kc = torch.randn(13, 256, Hkv, 128, generator=g).bfloat16().to(DEV)
    vc = torch.randn(13, 256, Hkv, 128, generator=g).bfloat16().to(DEV)
    qkv = torch.randn(B, (Hq + 2 * Hkv) * 128, generator=g).bfloat16().to(DEV)
    qw = torch.ones(128).bfloat16().to(DEV)
    bt = (1 + torch.arange(12, dtype=torch.int32).reshape(3, 4)).to(DEV)
    cl = torch.tensor([700, 200, 256], dtype=torch.int32, device=DEV)
    pos = cl.to(torch.int64)
    from nano_qwen3tts_vllm.models.hip_decoder import rope_tables
    ct, st = (t.to(DEV) for t in rope_tables(128, 1000000.0, 2048))
    nbytes = hip.lib.qtts_attn_decode_workspace_bytes(B, Hq, Hkv, 1024)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    out = torch.full((B, Hq * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    for _ in range(2):
        hip.check(hip.lib.qtts_attention_decode_fused(hip.ptr(qkv), qkv.stride(0), hip.ptr(qw), hip.ptr(qw), 1e-6, hip.ptr(ct),
                                                      hip.ptr(st), 2048, hip.ptr(pos), None, hip.ptr(kc), hip.ptr(vc), hip.ptr(bt),
                                                      4, hip.ptr(cl), maxc, 128 ** -0.5, hip.ptr(out), Hq * 128, B, Hq, Hkv,
                                                      hip.ptr(ws), hip.stream()), "attention_decode_fused")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out.float()).all())
    part = ((B * Hq * ((maxc + 127) // 128) * 130 * 4 + 255) // 256) * 256
    assert int(ws[part:part + B * Hkv * 4].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("M,I,K", [(8, 1024, 2048), (20, 512, 256), (20, 1024, 2048), (3, 512, 3072)])
def test_swiglu_launches_with_few_tiles_stay_inside_their_output(hip, M, I, K):
    """Narrow GEMMs with a long K share a 16-row weight tile between two workgroups; that form exists for the plain
    bias / residual epilogue of a single row tile only.  SwiGLU launches whose size would qualify (few tiles, a large
    slice per workgroup), with one and with two row tiles, must take the ordinary path: I outputs per row (not 2 I),
    nothing written past the rows' end (the output sits in the middle of a larger buffer of sentinels), oracle
    arithmetic.  An out-of-bounds store of exactly this kind aborted a test run during development."""
    g = torch.Generator().manual_seed(M + I + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    wg = (torch.randn(I, K, generator=g) * 0.05).bfloat16()
    wu = (torch.randn(I, K, generator=g) * 0.05).bfloat16()
    want = O.silu_mul(torch.nn.functional.linear(x, wg), torch.nn.functional.linear(x, wu))
    wp = hip.pack_gate_up(wg.to(DEV), wu.to(DEV))
    buf = torch.full((M + 8, I), 777.0, dtype=torch.bfloat16, device=DEV)
    out = buf[4:4 + M]
    hip.linear_swiglu(x.to(DEV), wp, I, out=out)
    torch.cuda.synchronize()
    assert bool((buf[:4] == 777.0).all()) and bool((buf[4 + M:] == 777.0).all())
    got = out.float().cpu()
    rel = (got - want.float()).abs() / (want.float().abs() + 1e-3)
    assert float((rel > 2.0 ** -6).float().mean()) < 1e-3 and float(rel.max()) < 0.1
    # the same rows through down_proj's form of the activation (silu(gate) * up applied on the way in)
    gu = torch.cat([torch.nn.functional.linear(x, wg), torch.nn.functional.linear(x, wu)], 1).to(DEV)
    wd = (torch.randn(256, I, generator=g) * 0.05).bfloat16()
    buf2 = torch.full((M + 8, 256), 777.0, dtype=torch.bfloat16, device=DEV)
    if I % 256 == 0:
        hip.linear_fused(gu, hip.pack_weight(wd.to(DEV)), 256, swiglu_input=True, out=buf2[4:4 + M])
        torch.cuda.synchronize()
        assert bool((buf2[:4] == 777.0).all()) and bool((buf2[4 + M:] == 777.0).all())
        wantd = torch.nn.functional.linear(want, wd).float()
        assert torch.allclose(buf2[4:4 + M].float().cpu(), wantd, atol=6e-2, rtol=2e-2)      # O(5) sums: a bf16 step is 0.03
