"""Code-predictor run_step on the HIP decoder
(reference: P/engine/model_runner/predictor_model_runner.py:25-265).

Two ways to run a request:
  * `run(seqs, is_prefill)` — one model step, the reference's contract: prefill of the
    2-position prompt or one decode step, head / embedding table chosen per sequence
    by its generation step (P/models/qwen3_tts_predictor.py:87-124);
  * `run_burst(seqs)` — the whole 15-token burst (prefill + 14 decodes) as ONE
    hipGraph with the pick and the token feedback on the device, so a frame costs one
    launch and one host read-back instead of 15 of each.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from ... import hip
from ...config import Config
from ...models.hip_decoder import DecoderShape, HipDecoder
from ...sampling_params import SamplingParams
from ..sequence import Sequence
from .base import ModelRunner, Staging, read_checkpoint


class PredictorSequence(Sequence):
    def __init__(self, token_ids: Optional[list[int]], sampling_params: SamplingParams | None = None,
                 input_embeds: Optional[torch.Tensor] = None, generation_steps: int = 0,
                 request_id: Optional[str] = None):
        super().__init__(token_ids, sampling_params, input_embeds, request_id=request_id)
        self.generation_steps = generation_steps


class PredictorModelRunner(ModelRunner):
    def __init__(self, config: Config, rank: int = 0, event=None, weights: dict | None = None):
        super().__init__(config, rank, event)
        tc = self.full_config.talker_config
        pc = tc.code_predictor_config
        self.model_config = pc
        self.talker_hidden = tc.hidden_size
        self.shape = s = DecoderShape.from_config(pc.to_dict())
        self.n_gen = pc.num_code_groups - 1                 # 15 codebooks per burst
        self.vocab = pc.vocab_size
        if weights is None:
            weights = read_checkpoint(config.model, "talker.code_predictor.", "talker.code_predictor.")
        dev = lambda t: t.to(self.device, torch.bfloat16).contiguous()
        self.max_bs = min(config.max_num_seqs, 512)
        self.max_rows = max(config.max_num_batched_tokens, 2 * self.max_bs)
        block_bytes = 2 * s.num_layers * self.block_size * s.num_kv_heads * s.head_dim * 2
        n_blocks = self.kv_blocks_for(block_bytes, reserve_bytes=(1 << 30))
        self.decoder = HipDecoder(weights, s, n_blocks, max_rows=self.max_rows, max_context=config.max_model_len,
                                  device=self.device)
        self.heads = [hip.pack_weight(dev(weights[f"lm_head.{i}.weight"])) for i in range(self.n_gen)]
        self.tables = [dev(weights[f"model.codec_embedding.{i}.weight"]) for i in range(self.n_gen)]
        if "small_to_mtp_projection.weight" in weights:
            self.proj_w = hip.pack_weight(dev(weights["small_to_mtp_projection.weight"]))
            self.proj_b = dev(weights["small_to_mtp_projection.bias"])
        else:
            assert self.talker_hidden == s.hidden_size
            self.proj_w = self.proj_b = None
        Ht, h = self.talker_hidden, s.hidden_size
        # Decode steps of a burst start from rows of a table (codec_embedding[i] of the previous token): the
        # projection of every table row, and layer 0's input norm + qkv projection of it, depend on the weights
        # only and are computed once here with the same GEMM launches a step would use (same bits); a step then
        # gathers instead of projecting.  QTTS_FOLD_INPUT=0 keeps the per-step launches.
        self.fold = os.environ.get("QTTS_FOLD_INPUT", "1") == "1"
        self.pair_prefill = os.environ.get("QTTS_PAIR_PREFILL", "1") == "1"   # 0: the general prompt path
        self.x_tables, self.qkv0_tables = [], []
        if self.fold:
            self._fold_tables()
        self.e_buf = torch.zeros(self.max_rows, Ht, dtype=torch.bfloat16, device=self.device)   # talker-width inputs
        self.x_buf = torch.zeros(self.max_rows, h, dtype=torch.bfloat16, device=self.device)    # after projection
        self.h_buf = torch.zeros(self.max_rows, h, dtype=torch.bfloat16, device=self.device)
        self.logits = torch.zeros(self.max_bs, self.vocab, dtype=torch.bfloat16, device=self.device)
        self.tokens = torch.zeros(self.max_bs, dtype=torch.int64, device=self.device)
        self.sampled = torch.zeros(self.max_bs, dtype=torch.int64, device=self.device)
        self.temps = torch.ones(self.max_bs, dtype=torch.float32, device=self.device)
        self.ids_buf = torch.zeros(self.max_bs, dtype=torch.int64, device=self.device)
        mb = self.max_blocks_per_seq
        self.pstage = Staging([("positions", self.max_rows, torch.int64), ("slots", self.max_rows, torch.int32),
                               ("ctx", self.max_rows, torch.int32), ("row_seq", self.max_rows, torch.int32),
                               ("tables", self.max_bs * mb, torch.int32)], self.device)
        # burst graphs: rows = 2*bs prefill rows followed by 14 groups of bs decode rows
        self.burst_bs = [b for b in [1, 2, 4, 8] if b <= self.max_bs] + list(range(16, self.max_bs + 1, 16))
        if self.burst_bs[-1] < self.max_bs:                 # e.g. 3, 5..7, 9..15, 24: the largest batch is a bucket of its own
            self.burst_bs.append(self.max_bs)
        self.burst_graphs: dict[int, hip.Graph] = {}
        rows = (2 + self.n_gen - 1) * self.max_bs
        self.bstage = Staging([("positions", rows, torch.int64), ("slots", rows, torch.int32),
                               ("ctx", rows, torch.int32), ("row_seq", 2 * self.max_bs, torch.int32),
                               ("tables", self.max_bs * mb, torch.int32), ("rng", 1, torch.int64),
                               ("temps", self.max_bs, torch.float32), ("greedy", self.max_bs, torch.int32)], self.device)
        self.burst_tokens = torch.zeros(self.n_gen * self.max_bs, dtype=torch.int64, device=self.device)
        # every step of a burst leaves its logits in a slice of its own (rows 0..bs-1 of burst_logits[step]): the
        # distribution behind each of the 15 picks of a frame stays readable after the burst
        self.burst_logits = torch.zeros(self.n_gen, self.max_bs, self.vocab, dtype=torch.bfloat16, device=self.device)

    # ---- pieces ---------------------------------------------------------------------
    def _fold_tables(self):
        """x_tables[i] = small_to_mtp_projection(codec_embedding[i].weight) (P/models/qwen3_tts_predictor.py:58-61,
        94-95,106); qkv0_tables[i] = layer 0's qkv_proj(input_layernorm(.)) of those rows
        (P/models/qwen3_tts_share.py:96-99,168-169).  Sixteen rows per launch: the decode GEMM kernel itself."""
        s, lib = self.shape, hip.lib
        h, Ht = s.hidden_size, self.talker_hidden
        qkvw = (s.num_heads + 2 * s.num_kv_heads) * s.head_dim
        d = self.decoder
        for tb in self.tables:
            rows = tb.shape[0]
            ids = torch.arange(rows, dtype=torch.int64, device=self.device)
            if self.proj_w is not None:
                xt = torch.empty(rows, h, dtype=torch.bfloat16, device=self.device)
                for c in range(0, rows, 16):
                    n = min(16, rows - c)
                    hip.check(lib.qtts_linear_fused(hip.ptr(tb), Ht, hip.ptr(ids) + 8 * c, rows, hip.ptr(self.proj_w),
                                                    hip.ptr(self.proj_b), None, 0, None, 0.0, None, 0, 0,
                                                    hip.ptr(xt) + 2 * c * h, h, n, h, Ht, hip.stream()), "fold proj")
            else:
                xt = tb
            qt = torch.empty(rows, qkvw, dtype=torch.bfloat16, device=self.device)
            for c in range(0, rows, 16):
                n = min(16, rows - c)
                if d.layer0_qkv_scales is not None:       # fp8 weights: the fp8 launch a step would make
                    hip.check(lib.qtts_linear_fp8(hip.ptr(xt) + 2 * c * h, h, hip.ptr(d.layer0_wqkv), hip.ptr(d.layer0_qkv_scales),
                                                  None, None, 0, hip.ptr(d.layer0_in_norm), s.rms_eps,
                                                  hip.ptr(qt) + 2 * c * qkvw, qkvw, n, qkvw, h, None, 0, hip.stream()), "fold qkv0")
                    continue
                hip.check(lib.qtts_linear_fused(hip.ptr(xt) + 2 * c * h, h, None, 0, hip.ptr(d.layer0_wqkv), None, None, 0,
                                                hip.ptr(d.layer0_in_norm), s.rms_eps, None, 0, 0,
                                                hip.ptr(qt) + 2 * c * qkvw, qkvw, n, qkvw, h, hip.stream()), "fold qkv0")
            self.x_tables.append(xt)
            self.qkv0_tables.append(qt)
        torch.cuda.synchronize()

    def _project(self, n_rows: int):
        """small_to_mtp_projection (P/models/qwen3_tts_predictor.py:58-61,106)."""
        if self.proj_w is None:
            return self.e_buf[:n_rows]
        hip.linear(self.e_buf[:n_rows], self.proj_w, self.shape.hidden_size, bias=self.proj_b, out=self.x_buf[:n_rows])
        return self.x_buf[:n_rows]

    def _logits_by_step(self, hidden_rows: torch.Tensor, steps: list[int]):
        """lm_head[generation_step] per sequence (P/models/qwen3_tts_predictor.py:110-124)."""
        S = len(steps)
        if len(set(steps)) == 1:
            hip.linear(hidden_rows, self.heads[steps[0]], self.vocab, out=self.logits[:S])
            return
        for st in sorted(set(steps)):
            idx = [i for i, g in enumerate(steps) if g == st]
            rows = hidden_rows.index_select(0, torch.tensor(idx, device=self.device)).contiguous()
            self.logits[torch.tensor(idx, device=self.device)] = hip.linear(rows, self.heads[st], self.vocab)

    def _pick(self, seqs, S) -> list[int]:
        hip.check(hip.lib.qtts_argmax(hip.ptr(self.logits), self.vocab, hip.ptr(self.tokens), S, self.vocab,
                                      hip.stream()), "argmax")
        if any(s.do_sample for s in seqs):
            self.temps[:S].copy_(torch.tensor([s.temperature for s in seqs], dtype=torch.float32), non_blocking=True)
            hip.check(hip.lib.qtts_sample_topk(hip.ptr(self.logits), self.vocab, hip.ptr(self.temps), seqs[0].top_k,
                                               self.rng_seed, self.next_rng_offset(), hip.ptr(self.sampled), S,
                                               self.vocab, hip.stream()), "sample_topk")
            greedy, drawn = self.tokens[:S].tolist(), self.sampled[:S].tolist()
            return [d if s.do_sample else g for s, g, d in zip(seqs, greedy, drawn)]
        return self.tokens[:S].tolist()

    def _upload_tables(self, st: Staging, seqs, S):
        mb = self.max_blocks_per_seq
        tab = st.h("tables")[:S * mb].view(S, mb)
        tab.fill_(0)
        for i, s in enumerate(seqs):
            tab[i, :len(s.block_table)] = torch.tensor(s.block_table, dtype=torch.int32)

    # ---- one model step (reference contract) ------------------------------------------
    @torch.inference_mode()
    def run(self, seqs: list[PredictorSequence], is_prefill: bool) -> list[int]:
        with self.on_stream():
            return self._run(seqs, is_prefill)

    def _run(self, seqs: list[PredictorSequence], is_prefill: bool) -> list[int]:
        S, mb, st = len(seqs), self.max_blocks_per_seq, self.pstage
        steps = [s.generation_steps for s in seqs]
        if is_prefill:
            positions, slots, row_seq, ctx, cu = self.prefill_metadata(seqs, self.block_size)
            N = len(positions)
            rows = [s.input_embeds.reshape(-1, s.input_embeds.shape[-1]) for s in seqs]
            self.e_buf[:N].copy_(torch.cat(rows, 0).to(torch.bfloat16), non_blocking=True)
            max_ctx = max(len(s) for s in seqs)
            last = torch.tensor(cu[1:], dtype=torch.int64, device=self.device) - 1
        else:
            N = S
            positions = [len(s) - 1 for s in seqs]                      # P/engine/model_runner/base.py:225
            ctx = [len(s) for s in seqs]
            slots = [s.block_table[-1] * self.block_size + s.last_block_num_tokens - 1 for s in seqs]
            row_seq = list(range(S))
            max_ctx = max(ctx)
            # codec_embedding[generation_step - 1](last_token) per sequence
            ids = torch.tensor([s.last_token for s in seqs], dtype=torch.int64, device=self.device)
            if len(set(steps)) == 1:
                tb = self.tables[steps[0] - 1]
                hip.check(hip.lib.qtts_embedding_gather(hip.ptr(tb), hip.ptr(ids), hip.ptr(self.e_buf),
                                                        self.talker_hidden, S, self.talker_hidden, tb.shape[0],
                                                        hip.stream()), "embedding_gather")
            else:
                for i, g in enumerate(steps):
                    self.e_buf[i] = hip.embedding_gather(self.tables[g - 1], ids[i:i + 1])[0]
            last = None
        st.h("positions")[:N] = torch.tensor(positions, dtype=torch.int64)
        st.h("slots")[:N] = torch.tensor(slots, dtype=torch.int32)
        st.h("ctx")[:N] = torch.tensor(ctx, dtype=torch.int32)
        st.h("row_seq")[:N] = torch.tensor(row_seq, dtype=torch.int32)
        self._upload_tables(st, seqs, S)
        st.upload()
        x = self._project(N)
        self.decoder.forward(x, st.d("positions")[:N], st.d("slots")[:N], st.d("ctx")[:N],
                             st.d("tables")[:S * mb].view(S, mb), max_ctx, row_seq=st.d("row_seq")[:N],
                             out=self.h_buf[:N])
        hidden = self.h_buf[:N] if last is None else self.h_buf[:N].index_select(0, last)
        self._logits_by_step(hidden, steps)
        return self._pick(seqs, S)

    # ---- the fused burst ----------------------------------------------------------------
    def _burst_body(self, bs: int):
        st, mb, h, Ht = self.bstage, self.max_blocks_per_seq, self.shape.hidden_size, self.talker_hidden
        tables = st.d("tables")[:bs * mb].view(bs, mb)
        stream = hip.stream

        def pick(step):
            hip.check(hip.lib.qtts_pick_token(hip.ptr(self.burst_logits[step]), self.vocab, hip.ptr(self._burst_temps),
                                              hip.ptr(self._greedy_bytes), self._top_k, self.rng_seed, step,
                                              hip.ptr(self._burst_rng), hip.ptr(self.burst_tokens) + 8 * step * bs, 1,
                                              bs, self.vocab, stream()), "pick_token")

        # prefill: 2*bs rows
        n0 = 2 * bs
        x = self._project(n0)
        if n0 <= 16 and self.pair_prefill:
            # one attention launch per layer for both positions, final norm inside the head GEMM (second rows only)
            hip.check(hip.lib.qtts_model_forward_logits_pairs(self.decoder._h, hip.ptr(x), hip.ptr(st.d("positions")),
                                                              hip.ptr(st.d("slots")), n0, hip.ptr(self.h_buf),
                                                              hip.ptr(self.heads[0]), self.vocab, hip.ptr(self.burst_logits[0]),
                                                              stream()), "model_forward_logits_pairs")
        else:
            self.decoder.forward(x, st.d("positions")[:n0], st.d("slots")[:n0], st.d("ctx")[:n0], tables, 2,
                                 row_seq=st.d("row_seq")[:n0], out=self.h_buf[:n0])
            # last position of each sequence = odd rows: a strided view, no gather needed
            hip.check(hip.lib.qtts_linear(hip.ptr(self.h_buf) + 2 * h, 2 * h, hip.ptr(self.heads[0]), None, None, 0, None,
                                          0.0, hip.ptr(self.burst_logits[0]), self.vocab, bs, self.vocab, h, stream()), "head0")
        pick(0)
        off = n0
        for step in range(1, self.n_gen):
            tb = self.tables[step - 1]
            prev = hip.ptr(self.burst_tokens) + 8 * (step - 1) * bs
            if self.fold:                            # (more than 16 rows: the rows are gathered once per step, layer 0's qkv rows still come from their table)
                xt, qt = self.x_tables[step - 1], self.qkv0_tables[step - 1]
                if getattr(self, "_uniform_rows", False):
                    # row r is sequence r in KV block r and every row is at context step + 2: no metadata arrays, the
                    # attention launches have nothing to wait for before they request K/V
                    meta = (None, None, None, None, 1)
                else:
                    meta = (hip.ptr(st.d("positions")[off:off + bs]), hip.ptr(st.d("slots")[off:off + bs]),
                            hip.ptr(st.d("ctx")[off:off + bs]), hip.ptr(tables), mb)
                hip.check(hip.lib.qtts_model_forward_logits_gathered(
                    self.decoder._h, hip.ptr(xt), hip.ptr(qt), prev, xt.shape[0], *meta, step + 2,
                    bs, hip.ptr(self.h_buf), hip.ptr(self.heads[step]), self.vocab, hip.ptr(self.burst_logits[step]), stream()),
                    "model_forward_logits_gathered")
                pick(step)
                off += bs
                continue
            if self.proj_w is not None:
                # codec_embedding[step-1](token) -> small_to_mtp_projection in one launch (gathered GEMM rows)
                hip.check(hip.lib.qtts_linear_fused(hip.ptr(tb), Ht, prev, tb.shape[0], hip.ptr(self.proj_w),
                                                    hip.ptr(self.proj_b), None, 0, None, 0.0, None, 0, 0,
                                                    hip.ptr(self.x_buf), h, bs, h, Ht, stream()), "gather+proj")
                x = self.x_buf[:bs]
            else:
                hip.check(hip.lib.qtts_embedding_gather(hip.ptr(tb), prev, hip.ptr(self.e_buf), Ht, bs, Ht,
                                                        tb.shape[0], stream()), "embedding_gather")
                x = self.e_buf[:bs]
            self.decoder.forward_logits(x, st.d("positions")[off:off + bs], st.d("slots")[off:off + bs],
                                        st.d("ctx")[off:off + bs], tables, step + 2, self.heads[step], self.vocab,
                                        hidden_out=self.h_buf[:bs], logits_out=self.burst_logits[step, :bs])
            pick(step)
            off += bs

    @torch.inference_mode()
    def run_burst(self, seqs: list[PredictorSequence]) -> list[list[int]]:
        """All `n_gen` tokens of freshly admitted sequences (each holding a [1,2,H] prompt)."""
        with self.on_stream():
            return self._run_burst(seqs)

    def _run_burst(self, seqs: list[PredictorSequence]) -> list[list[int]]:
        S, mb, st = len(seqs), self.max_blocks_per_seq, self.bstage
        bs = next(b for b in self.burst_bs if b >= S)
        assert all(len(s) == 2 and s.generation_steps == 0 for s in seqs)
        pos, slot, ctx, rs = st.h("positions"), st.h("slots"), st.h("ctx"), st.h("row_seq")
        n0 = 2 * bs
        total = n0 + (self.n_gen - 1) * bs
        pos[:total] = 0
        slot[:total] = -1
        ctx[:total] = 0
        base = [s.block_table[0] * self.block_size for s in seqs]
        for i in range(bs):
            rs[2 * i] = rs[2 * i + 1] = min(i, S - 1)
        for i, b in enumerate(base):
            pos[2 * i], pos[2 * i + 1] = 0, 1
            slot[2 * i], slot[2 * i + 1] = b, b + 1
            ctx[2 * i], ctx[2 * i + 1] = 1, 2
            for step in range(1, self.n_gen):
                r = n0 + (step - 1) * bs + i
                pos[r] = step + 1                  # len(seq) - 1 with len = step + 2
                slot[r] = b + step + 1
                ctx[r] = step + 2
        self._upload_tables(st, seqs, S)
        st.h("rng")[0] = self.next_rng_offset() * 64
        st.h("temps")[:S] = torch.tensor([s.temperature for s in seqs], dtype=torch.float32)
        st.h("temps")[S:bs] = 1.0
        flags = torch.ones(bs, dtype=torch.int32)
        flags[:S] = torch.tensor([0 if s.do_sample else 1 for s in seqs], dtype=torch.int32)
        st.h("greedy")[:bs] = flags
        st.upload()
        self._greedy_bytes = self._greedy_u8(bs)
        self._top_k = seqs[0].top_k
        self._burst_temps, self._burst_rng = st.d("temps"), st.d("rng")
        self._uniform_rows = False                   # sequences sit in whatever blocks the block manager gave them
        rows = [s.input_embeds.reshape(-1, s.input_embeds.shape[-1]) for s in seqs]
        self.e_buf[:2 * S].copy_(torch.cat(rows, 0).to(torch.bfloat16), non_blocking=True)
        if self.enforce_eager:
            self._burst_body(bs)
        else:
            key = (bs, self._top_k)
            g = self.burst_graphs.get(key)
            if g is None:
                self._burst_body(bs)
                g = self.burst_graphs[key] = hip.Graph.capture(lambda: self._burst_body(bs))
                # the warm-up launch consumed this burst's inputs (e_buf is overwritten by the
                # decode steps), so refill before the replay that produces the result
                self.e_buf[:2 * S].copy_(torch.cat(rows, 0).to(torch.bfloat16), non_blocking=True)
            g.launch()
        toks = self.burst_tokens[:self.n_gen * bs].view(self.n_gen, bs)[:, :S].t().tolist()
        return toks

    def _greedy_u8(self, bs: int) -> torch.Tensor:
        """Compact uint8 flag vector next to the staged int32 flags (the kernel reads bytes)."""
        if not hasattr(self, "_gflags"):
            self._gflags = torch.zeros(self.max_bs, dtype=torch.uint8, device=self.device)
        self._gflags[:bs].copy_(self.bstage.d("greedy")[:bs].to(torch.uint8))
        return self._gflags

    def exit(self):
        self.burst_graphs.clear()
        super().exit()
