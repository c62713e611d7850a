"""TEST INFRASTRUCTURE — CPU restatement of the reference's talker + code-predictor
`run_step` arithmetic (torch eager, bf16 storage).  It is the checker for the HIP
path; it is never imported by the product package.

Pinning: every function here is checked bit-for-bit against the reference's own
modules imported in the build container (oracle/gen_golden.py writes the vectors,
tests/test_oracle_golden.py replays them).  The one place where the reference's
arithmetic lives in an absent third-party dependency is attention
(`flash_attn`, unpinned wheel; README suggests 2.6.3): there this file restates
the published definition softmax(Q K^T * scale) V in fp32 with a causal mask
aligned to the last key — parity unpinned for that op, tolerance-checked only.

Reference files followed (P = /root/reference/nano-qwen3tts-vllm):
  P/layers/layernorm.py:17-27           rms_norm
  P/layers/rotary_embedding.py:5-9,33-34,98,119-139,190-200   rope tables / apply
  P/layers/attention.py:33-54,73-96     store_kvcache / attention call sites
  P/layers/activation.py:11-13, P/models/qwen3_tts_share.py:79-113,131-135,163-176
  P/models/qwen3_tts_talker.py:53-62,131-136
  P/models/qwen3_tts_predictor.py:87-124
  P/layers/embed_head.py:56-61
  P/engine/model_runner/base.py:172-234, talker_mode_runner.py:116-153,
  predictor_model_runner.py:134-157     batch preparation semantics
  P/interface.py:1241-1290              per-frame glue (fp32)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn.functional as F

BLOCK = 256  # P/engine/sequence.py:16, P/config.py:52


@dataclass
class DecoderCfg:
    hidden_size: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    head_dim: int
    intermediate_size: int
    vocab_size: int
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    # predictor only
    talker_hidden_size: Optional[int] = None
    num_code_groups: int = 16


# --------------------------------------------------------------------------
# element ops
# --------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """P/layers/layernorm.py:17-27 — fp32 statistics, cast to bf16, THEN * weight."""
    dt = x.dtype
    xf = x.float()
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return xf.to(dt) * w


def rope_tables(positions: torch.Tensor, head_dim: int, theta: float, dtype=torch.bfloat16):
    """cos/sin [N, head_dim] rounded to `dtype` (P/layers/rotary_embedding.py:98,134-139).

    The talker's 3-row M-RoPE degenerates to this because the runners pass 1-D
    positions (rotary_embedding.py:179-180 expands them to three equal rows).
    """
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    freqs = positions.float()[:, None] * inv_freq[None, :]        # K=1 matmul == product
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q, k, cos, sin):
    """q [N,Hq,D], k [N,Hkv,D], cos/sin [N,D]; three bf16-rounded ops per tensor."""
    c = cos[:, None, :]
    s = sin[:, None, :]
    return (q * c) + (rotate_half(q) * s), (k * c) + (rotate_half(k) * s)


def silu_mul(gate, up):
    """P/models/qwen3_tts_share.py:133 — silu rounds to bf16, then the product rounds."""
    return F.silu(gate) * up


def store_kvcache(k, v, k_cache, v_cache, slot_mapping):
    """P/layers/attention.py:10-30 — row copy by slot, slot == -1 skipped."""
    N, H, D = k.shape
    kc = k_cache.view(-1, H, D)
    vc = v_cache.view(-1, H, D)
    for i in range(N):
        s = int(slot_mapping[i])
        if s >= 0:
            kc[s] = k[i]
            vc[s] = v[i]


# How the probabilities enter the P.V product.  None: fp32 softmax, fp32 product (the textbook definition, the
# default and what the fixtures were generated with).  torch.bfloat16: the FlashAttention-2 forward as published
# (Dao 2023, Algorithm 1; the flash_attn wheel the reference calls, P/layers/attention.py:88-95, is absent here):
# unnormalised exp(s - max) rounded to the input dtype before the product with V, fp32 accumulation, division by the
# fp32 row sum of the unrounded values at the end.  Switched by tests that quantify the difference between the two.
ATTN_P_DTYPE = None


def _attend(q, k, v, scale, causal_offset):
    Lq, Hq, D = q.shape
    Lk, Hkv, _ = k.shape
    g = Hq // Hkv
    qf = q.float().transpose(0, 1)
    kf = k.float().repeat_interleave(g, dim=1).transpose(0, 1)
    vf = v.float().repeat_interleave(g, dim=1).transpose(0, 1)
    s = torch.matmul(qf, kf.transpose(1, 2)) * scale
    iq = torch.arange(Lq).unsqueeze(1) + causal_offset
    ik = torch.arange(Lk).unsqueeze(0)
    s = s.masked_fill(ik > iq, float("-inf"))
    if ATTN_P_DTYPE is not None:
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        o = torch.matmul(e.to(ATTN_P_DTYPE).float(), vf) / e.sum(dim=-1, keepdim=True)
        return o.transpose(0, 1).to(q.dtype)
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, vf).transpose(0, 1).to(q.dtype)


def attention_prefill(q, k, v, cu_seqlens, scale):
    """Causal attention over packed sequences (call site P/layers/attention.py:88-91)."""
    outs = []
    for i in range(len(cu_seqlens) - 1):
        a, b = int(cu_seqlens[i]), int(cu_seqlens[i + 1])
        outs.append(_attend(q[a:b], k[a:b], v[a:b], scale, 0))
    return torch.cat(outs, dim=0)


def attention_decode(q, k_cache, v_cache, block_tables, context_lens, scale):
    """One query token per sequence over block-table KV (P/layers/attention.py:93-95).
    Rows with context_len == 0 (graph padding) return zeros."""
    B, Hq, D = q.shape
    bs = k_cache.shape[1]
    out = torch.zeros_like(q)
    for b in range(B):
        n = int(context_lens[b])
        if n == 0:
            continue
        nb = (n + bs - 1) // bs
        blocks = [int(x) for x in block_tables[b, :nb]]
        kk = torch.cat([k_cache[i] for i in blocks], dim=0)[:n]
        vv = torch.cat([v_cache[i] for i in blocks], dim=0)[:n]
        out[b] = _attend(q[b:b + 1], kk, vv, scale, n - 1)[0]
    return out


# --------------------------------------------------------------------------
# decoder stack
# --------------------------------------------------------------------------
@dataclass
class AttnMeta:
    is_prefill: bool
    positions: torch.Tensor                      # int64 [N]
    slot_mapping: Optional[torch.Tensor] = None   # int32 [N]
    cu_seqlens: Optional[torch.Tensor] = None     # prefill
    context_lens: Optional[torch.Tensor] = None   # decode
    block_tables: Optional[torch.Tensor] = None   # decode


def decoder_layer(w: dict, prefix: str, cfg: DecoderCfg, x, meta: AttnMeta, kv=None, debug=None):
    """P/models/qwen3_tts_share.py:163-176 (layer) and :79-113 (attention block)."""
    p = prefix
    res = x
    h = rms_norm(x, w[p + "input_layernorm.weight"], cfg.rms_eps)
    q = F.linear(h, w[p + "self_attn.q_proj.weight"]).view(-1, cfg.num_heads, cfg.head_dim)
    k = F.linear(h, w[p + "self_attn.k_proj.weight"]).view(-1, cfg.num_kv_heads, cfg.head_dim)
    v = F.linear(h, w[p + "self_attn.v_proj.weight"]).view(-1, cfg.num_kv_heads, cfg.head_dim)
    q = rms_norm(q, w[p + "self_attn.q_norm.weight"], cfg.rms_eps)
    k = rms_norm(k, w[p + "self_attn.k_norm.weight"], cfg.rms_eps)
    cos, sin = rope_tables(meta.positions, cfg.head_dim, cfg.rope_theta, q.dtype)
    q, k = apply_rope(q, k, cos, sin)
    if kv is not None and meta.slot_mapping is not None:
        store_kvcache(k, v, kv[0], kv[1], meta.slot_mapping)
    scale = cfg.head_dim ** -0.5
    if meta.is_prefill:
        o = attention_prefill(q, k, v, meta.cu_seqlens, scale)
    else:
        o = attention_decode(q, kv[0], kv[1], meta.block_tables, meta.context_lens, scale)
    if debug is not None:
        debug.update(q=q, k=k, v=v, attn=o)
    a = F.linear(o.flatten(1, -1), w[p + "self_attn.o_proj.weight"])
    x = res + a
    res = x
    h = rms_norm(x, w[p + "post_attention_layernorm.weight"], cfg.rms_eps)
    g = F.linear(h, w[p + "mlp.gate_proj.weight"])
    u = F.linear(h, w[p + "mlp.up_proj.weight"])
    d = F.linear(silu_mul(g, u), w[p + "mlp.down_proj.weight"])
    return res + d


def decoder_stack(w, cfg: DecoderCfg, x, meta: AttnMeta, kv_cache=None):
    """N x layer + final norm (P/models/qwen3_tts_talker.py:53-62,
    qwen3_tts_predictor.py:30-44).  kv_cache: [2, layers, blocks, BLOCK, Hkv, D]."""
    for i in range(cfg.num_layers):
        kv = None if kv_cache is None else (kv_cache[0, i], kv_cache[1, i])
        x = decoder_layer(w, f"model.layers.{i}.", cfg, x, meta, kv)
    return rms_norm(x, w["model.norm.weight"], cfg.rms_eps)


def talker_logits(w, hidden):
    """P/layers/embed_head.py:61 (rows already reduced to one per sequence)."""
    return F.linear(hidden, w["codec_head.weight"])


def predictor_project(w, x):
    """P/models/qwen3_tts_predictor.py:58-61,106 — Linear with bias, or identity."""
    if "small_to_mtp_projection.weight" in w:
        return F.linear(x, w["small_to_mtp_projection.weight"], w["small_to_mtp_projection.bias"])
    return x


def predictor_logits(w, hidden_last, generation_steps):
    """P/models/qwen3_tts_predictor.py:110-124 — one head per sequence by step."""
    rows = [F.linear(hidden_last[i:i + 1], w[f"lm_head.{g}.weight"]) for i, g in enumerate(generation_steps)]
    return torch.cat(rows, dim=0)


def greedy(logits):
    """Build extension keyed on SamplingParams.do_sample=False: argmax of fp32
    logits, lowest index on ties."""
    return torch.argmax(logits.float(), dim=-1)


def topk_probs(logits, temperature, top_k=50):
    """P/layers/sampler.py:15-21 — the distribution multinomial draws from."""
    lg = logits.float() / temperature.unsqueeze(1)
    kth = torch.topk(lg, top_k)[0][..., -1, None]
    lg = lg.masked_fill(lg < kth, float("-inf"))
    return torch.softmax(lg, dim=-1)


# --------------------------------------------------------------------------
# engines with paged KV (runner semantics)
# --------------------------------------------------------------------------
class _Seq:
    def __init__(self, n_prompt):
        self.num_tokens = n_prompt
        self.block_table: list[int] = []


class _Paged:
    def __init__(self, cfg: DecoderCfg, num_blocks: int):
        self.cfg = cfg
        self.kv = torch.zeros(2, cfg.num_layers, num_blocks, BLOCK, cfg.num_kv_heads, cfg.head_dim,
                              dtype=torch.bfloat16)
        self.free = list(range(num_blocks))

    def ensure(self, seq: _Seq):
        need = (seq.num_tokens + BLOCK - 1) // BLOCK
        while len(seq.block_table) < need:
            seq.block_table.append(self.free.pop(0))

    def release(self, seq: _Seq):
        self.free.extend(reversed(seq.block_table))
        seq.block_table = []

    def tables(self, seqs):
        m = max(len(s.block_table) for s in seqs)
        return torch.tensor([s.block_table + [-1] * (m - len(s.block_table)) for s in seqs], dtype=torch.int32)


class OracleTalker:
    """Talker run_step: prefill packs prompts (base.py:172-216); decode uses
    position = len(seq) but slot = len(seq)-1 (talker_mode_runner.py:126-128)."""

    def __init__(self, w, cfg: DecoderCfg, num_blocks=16):
        self.w, self.cfg = w, cfg
        self.pages = _Paged(cfg, num_blocks)

    def prefill(self, embeds_list):
        """embeds_list: list of [L,H] (any float dtype). Returns (seqs, token_ids, hidden[S,H], logits)."""
        seqs, xs, pos, slots, cu = [], [], [], [], [0]
        for e in embeds_list:
            s = _Seq(e.shape[0])
            self.pages.ensure(s)
            seqs.append(s)
            xs.append(e.to(torch.bfloat16))
            pos += list(range(e.shape[0]))
            for t in range(e.shape[0]):
                slots.append(s.block_table[t // BLOCK] * BLOCK + t % BLOCK)
            cu.append(cu[-1] + e.shape[0])
        meta = AttnMeta(True, torch.tensor(pos), torch.tensor(slots, dtype=torch.int32),
                        cu_seqlens=torch.tensor(cu, dtype=torch.int32))
        h = decoder_stack(self.w, self.cfg, torch.cat(xs, 0), meta, self.pages.kv)
        last = torch.tensor(cu[1:]) - 1
        hl = h[last].contiguous()
        logits = talker_logits(self.w, hl)
        toks = greedy(logits)
        for s in seqs:
            s.num_tokens += 1
        return seqs, toks.tolist(), hl, logits

    def decode(self, seqs, embeds):
        """embeds [B,H]. Returns (token_ids, hidden[B,H], logits)."""
        pos, slots, ctx = [], [], []
        for s in seqs:
            self.pages.ensure(s)
            n = s.num_tokens
            pos.append(n)
            ctx.append(n)
            slots.append(s.block_table[(n - 1) // BLOCK] * BLOCK + (n - 1) % BLOCK)
        meta = AttnMeta(False, torch.tensor(pos), torch.tensor(slots, dtype=torch.int32),
                        context_lens=torch.tensor(ctx, dtype=torch.int32),
                        block_tables=self.pages.tables(seqs))
        h = decoder_stack(self.w, self.cfg, embeds.to(torch.bfloat16), meta, self.pages.kv)
        logits = talker_logits(self.w, h)
        toks = greedy(logits)
        for s in seqs:
            s.num_tokens += 1
        return toks.tolist(), h, logits


class OraclePredictor:
    """Predictor burst: 2-token prefill then 14 decodes, head/table chosen by
    generation step (predictor_model_runner.py:134-157, qwen3_tts_predictor.py:87-124)."""

    def __init__(self, w, cfg: DecoderCfg, num_blocks=16):
        self.w, self.cfg = w, cfg
        self.pages = _Paged(cfg, num_blocks)

    def burst(self, embeds_list, return_logits=False, forced=None):
        """embeds_list: list of [2,H_talker]. Returns list of 15-token lists.
        forced [B][15] (optional): the ids fed back instead of the arg-max of each step (teacher forcing: the
        logits of every step are then those of a caller-chosen history); the returned ids are still the arg-max."""
        B = len(embeds_list)
        n_gen = self.cfg.num_code_groups - 1
        seqs, xs, pos, slots, cu = [], [], [], [], [0]
        for e in embeds_list:
            s = _Seq(2)
            self.pages.ensure(s)
            seqs.append(s)
            xs.append(e.to(torch.bfloat16))
            pos += [0, 1]
            slots += [s.block_table[0] * BLOCK, s.block_table[0] * BLOCK + 1]
            cu.append(cu[-1] + 2)
        meta = AttnMeta(True, torch.tensor(pos), torch.tensor(slots, dtype=torch.int32),
                        cu_seqlens=torch.tensor(cu, dtype=torch.int32))
        x = predictor_project(self.w, torch.cat(xs, 0))
        h = decoder_stack(self.w, self.cfg, x, meta, self.pages.kv)
        hl = h.view(B, 2, -1)[:, -1, :]
        logits = predictor_logits(self.w, hl, [0] * B)
        all_logits = [logits]
        tok = greedy(logits)
        out = [[int(t)] for t in tok]
        if forced is not None:
            tok = torch.tensor([int(f[0]) for f in forced])
        for s in seqs:
            s.num_tokens += 1
        for step in range(1, n_gen):
            emb = torch.stack([self.w[f"model.codec_embedding.{step - 1}.weight"][int(t)] for t in tok])
            pos = [s.num_tokens - 1 for s in seqs]
            slots = [s.block_table[0] * BLOCK + s.num_tokens - 1 for s in seqs]
            ctx = [s.num_tokens for s in seqs]
            meta = AttnMeta(False, torch.tensor(pos), torch.tensor(slots, dtype=torch.int32),
                            context_lens=torch.tensor(ctx, dtype=torch.int32),
                            block_tables=self.pages.tables(seqs))
            x = predictor_project(self.w, emb)
            h = decoder_stack(self.w, self.cfg, x, meta, self.pages.kv)
            logits = predictor_logits(self.w, h, [step] * B)
            all_logits.append(logits)
            tok = greedy(logits)
            for i, t in enumerate(tok):
                out[i].append(int(t))
            if forced is not None:
                tok = torch.tensor([int(f[step]) for f in forced])
            for s in seqs:
                s.num_tokens += 1
        for s in seqs:
            self.pages.release(s)
        if return_logits:
            return out, all_logits
        return out


def frame_glue(tw, pw, token0, pred_tokens, text_add):
    """P/interface.py:1276-1285 — next talker input = sum of the 16 codebook
    embeddings (fp32 copies of the bf16 tables) + text/pad hidden, fp32.
    token0: int, pred_tokens: 15 ints, text_add: fp32 [H]. Returns fp32 [H]."""
    rows = [tw["model.codec_embedding.weight"][token0].float()]
    rows += [pw[f"model.codec_embedding.{i}.weight"][pred_tokens[i]].float() for i in range(len(pred_tokens))]
    s = torch.stack(rows, 0).unsqueeze(0).sum(1, keepdim=True)[0, 0]
    return s + text_add


def frame_loop_logits(tw, tcfg, pw, pcfg, prompts, text_adds, n_frames, forced=None, num_blocks=32):
    """The frame loop of P/interface.py:1241-1290 (in-process shape: :1322-1366), greedy, all streams in one batch,
    keeping the logits of every pick.  prompts: list of [L,H]; text_adds fp32 [S, n_frames, H] (what the API adds per
    frame: trailing text hidden or the tts pad embedding).
    forced [S, n_frames, 16] (optional): the ids that are FED ON (to the predictor's next step, the glue, the
    talker's next input) instead of each pick's arg-max, so the logits of every pick are those of a caller-chosen
    history — the way a run of the HIP path is checked pick by pick without a near-tie hiding what follows.
    Returns dict: frames [S,NF,16] int64 (the arg-max of each pick), t_logits [S,NF,V], p_logits [S,NF,15,Vp],
    hidden [S,NF,H] (the talker hidden state that opens each frame), all float32."""
    talker = OracleTalker(tw, tcfg, num_blocks)
    pred = OraclePredictor(pw, pcfg, num_blocks)
    S = len(prompts)
    frames = torch.zeros(S, n_frames, 16, dtype=torch.int64)
    t_logits, p_logits, hidden = [], [], []
    seqs, tok, hid, logits = talker.prefill(prompts)
    for f in range(n_frames):
        t_logits.append(logits.float())
        hidden.append(hid.float())
        frames[:, f, 0] = torch.tensor(tok)
        fed0 = [int(forced[i][f][0]) for i in range(S)] if forced is not None else [int(t) for t in tok]
        emb0 = tw["model.codec_embedding.weight"][torch.tensor(fed0)].float()
        pin = [torch.stack([hid[i].float(), emb0[i]], 0) for i in range(S)]
        fp = [[int(v) for v in forced[i][f][1:]] for i in range(S)] if forced is not None else None
        ptoks, plogits = pred.burst(pin, return_logits=True, forced=fp)
        p_logits.append(torch.stack([l.float() for l in plogits], 1))
        nxt = []
        for i in range(S):
            frames[i, f, 1:] = torch.tensor(ptoks[i])
            fed = fp[i] if fp is not None else ptoks[i]
            nxt.append(frame_glue(tw, pw, fed0[i], fed, text_adds[i, f]))
        if f + 1 < n_frames:
            tok, hid, logits = talker.decode(seqs, torch.stack(nxt, 0))
    return dict(frames=frames, t_logits=torch.stack(t_logits, 1), p_logits=torch.stack(p_logits, 1),
                hidden=torch.stack(hidden, 1))


def frame_loop(tw, tcfg, pw, pcfg, prompts, text_adds, n_frames, num_blocks=32):
    """frame_loop_logits reduced to (frames [S,NF,16], top-2 margin of the talker logits per frame, smallest top-2
    margin over the 16 picks of the frame)."""
    r = frame_loop_logits(tw, tcfg, pw, pcfg, prompts, text_adds, n_frames, None, num_blocks)
    t2 = torch.topk(r["t_logits"], 2, dim=-1)[0]
    margins = t2[..., 0] - t2[..., 1]
    p2 = torch.topk(r["p_logits"], 2, dim=-1)[0]
    pick_margins = torch.minimum(margins, (p2[..., 0] - p2[..., 1]).min(dim=-1)[0])
    return r["frames"], margins, pick_margins


class other_summation_order:
    """Context manager: every F.linear of this file runs with another fp32 summation order (K in four chunks, each
    summed in fp32, the chunks added in fp32, one rounding to bf16 at the same point).  The arithmetic and its
    rounding points are unchanged, so the distance between a run under it and a plain run is the noise floor any
    second implementation of the reference's arithmetic sits on (28 layers deep it reaches several bf16 steps): the
    yardstick for the HIP path at full model size."""

    def __enter__(self):
        import types

        real = globals()["F"]

        def linear(x, w, b=None):
            c = max(x.shape[-1] // 4, 1)
            acc = None
            for i in range(0, x.shape[-1], c):
                part = real.linear(x[..., i:i + c].float(), w[:, i:i + c].float())
                acc = part if acc is None else acc + part
            if b is not None:
                acc = acc + b.float()
            return acc.to(x.dtype)

        self._saved = real
        globals()["F"] = types.SimpleNamespace(linear=linear, silu=real.silu)
        return self

    def __exit__(self, *exc):
        globals()["F"] = self._saved
        return False


def bf16_ulp(x: torch.Tensor) -> torch.Tensor:
    """Spacing of bf16 numbers at |x| (8 significant bits): 2^(floor(log2|x|) - 7); the smallest normal's for 0."""
    a = x.float().abs().clamp_min(2.0 ** -126)
    return torch.exp2(torch.floor(torch.log2(a)) - 7)


def judge_picks(ids: torch.Tensor, logits: torch.Tensor, tol_ulps: float = 2.0):
    """How greedy picks made by another implementation of the same arithmetic stand against this file's logits of
    the same history.  ids [...], logits [..., V] (bf16 values).  A pick is
      acceptable  when its logit is within tol_ulps bf16 steps (at the size of the row's largest logit) of the
                  largest — it is the arg-max, or a value that a last-bit difference of two fp32 summation orders
                  can lift to a tie with it;
      binding     when the runner-up is farther than that below the largest: then only the arg-max is acceptable.
    Returns (acceptable [...], binding [...], equal-to-arg-max [...]) as bool tensors."""
    lg = logits.float()
    top2 = torch.topk(lg, 2, dim=-1)[0]
    tol = tol_ulps * bf16_ulp(top2[..., 0])
    chosen = torch.gather(lg, -1, ids.long().unsqueeze(-1)).squeeze(-1)
    acceptable = (top2[..., 0] - chosen) <= tol
    binding = (top2[..., 0] - top2[..., 1]) > tol
    return acceptable, binding, ids.long() == torch.argmax(lg, dim=-1)


def quantize_fp8_rows(w: torch.Tensor):
    """Weight-only fp8 (e4m3, OCP) with one power-of-two scale per row — the arithmetic of the build's fp8
    extension (the reference has bf16 weights only).  Returns (q as float8_e4m3fn, scale fp32 [N], the dequantised
    weights q * scale as bf16, exact): scale is the smallest power of two with max|row| / scale <= 448."""
    wf = w.float()
    amax = wf.abs().amax(dim=1)
    mant, exp = torch.frexp(amax / 448.0)
    exp = torch.where(mant == 0.5, exp - 1, exp)
    scale = torch.where(amax > 0, torch.ldexp(torch.ones_like(amax), exp), torch.ones_like(amax))
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    deq = (q.float() * scale[:, None]).to(torch.bfloat16)
    assert torch.equal(deq.float(), q.float() * scale[:, None])
    return q, scale, deq
