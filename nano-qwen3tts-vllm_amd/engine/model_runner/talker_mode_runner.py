"""Talker run_step on the HIP decoder (reference: P/engine/model_runner/talker_mode_runner.py:25-234).

`run(seqs, is_prefill)` returns `(token_ids, hidden_states)` like the reference:
hidden_states is the post-final-norm last-token row of each sequence, bf16 on the
device.  Decode batches replay a hipGraph captured per batch-size bucket
(forward + codec head + greedy pick); prefill runs the same C entry points eagerly.
"""
from __future__ import annotations

import torch

from ... import hip
from ...config import Config
from ...models.hip_decoder import DecoderShape, HipDecoder
from ..sequence import Sequence
from .base import ModelRunner, Staging, read_checkpoint


class TalkerModeModelRunner(ModelRunner):
    def __init__(self, config: Config, rank: int = 0, event=None, weights: dict | None = None):
        super().__init__(config, rank, event)
        tc = self.full_config.talker_config
        self.model_config = tc
        self.shape = DecoderShape.from_config(tc.to_dict())
        if weights is None:
            weights = read_checkpoint(config.model, "talker.", "talker.",
                                      skip_prefixes=("talker.code_predictor.", "talker.model.text_embedding.",
                                                     "talker.text_projection."))
        H = self.shape.hidden_size
        self.vocab = tc.vocab_size
        self.max_bs = min(config.max_num_seqs, 512)
        self.max_rows = max(config.max_num_batched_tokens, self.max_bs)
        s = self.shape
        block_bytes = 2 * s.num_layers * self.block_size * s.num_kv_heads * s.head_dim * 2
        n_blocks = self.kv_blocks_for(block_bytes, reserve_bytes=self._estimate_bytes(weights))
        self.decoder = HipDecoder(weights, s, n_blocks, max_rows=self.max_rows, max_context=config.max_model_len,
                                  device=self.device)
        self.codec_head = hip.pack_weight(weights["codec_head.weight"].to(self.device, torch.bfloat16).contiguous())
        self.codec_embedding = weights["model.codec_embedding.weight"].to(self.device, torch.bfloat16).contiguous()
        # static buffers
        self.x_buf = torch.zeros(self.max_rows, H, dtype=torch.bfloat16, device=self.device)
        self.h_buf = torch.zeros(self.max_rows, H, dtype=torch.bfloat16, device=self.device)
        self.logits = torch.zeros(self.max_bs, self.vocab, dtype=torch.bfloat16, device=self.device)
        self.tokens = torch.zeros(self.max_bs, dtype=torch.int64, device=self.device)
        self.sampled = torch.zeros(self.max_bs, dtype=torch.int64, device=self.device)
        self.temps = torch.ones(self.max_bs, dtype=torch.float32, device=self.device)
        mb = self.max_blocks_per_seq
        # decode metadata is small and re-uploaded every step; prefill metadata is per token
        self.dstage = Staging([("positions", self.max_bs, torch.int64), ("slots", self.max_bs, torch.int32),
                               ("ctx", self.max_bs, torch.int32), ("tables", self.max_bs * mb, torch.int32)], self.device)
        self.pstage = Staging([("positions", self.max_rows, torch.int64), ("slots", self.max_rows, torch.int32),
                               ("ctx", self.max_rows, torch.int32), ("row_seq", self.max_rows, torch.int32),
                               ("tables", self.max_bs * mb, torch.int32)], self.device)
        self.graph_bs = [b for b in [1, 2, 4, 8] if b <= self.max_bs] + list(range(16, self.max_bs + 1, 16))
        if self.graph_bs[-1] < self.max_bs:                 # e.g. 3, 5..7, 9..15, 24: the largest batch is a bucket of its own
            self.graph_bs.append(self.max_bs)
        self.graphs: dict[int, hip.Graph] = {}

    def _estimate_bytes(self, weights):
        return int(sum(v.numel() for v in weights.values()) * 2 * 1.1) + (2 << 30)

    # ---- the forward pieces (C-ABI calls only; capturable) -------------------------
    def _decode_body(self, bs: int):
        st = self.dstage
        mb = self.max_blocks_per_seq
        self.decoder.forward_logits(self.x_buf[:bs], st.d("positions")[:bs], st.d("slots")[:bs], st.d("ctx")[:bs],
                                    st.d("tables")[:bs * mb].view(bs, mb), self.config.max_model_len, self.codec_head,
                                    self.vocab, hidden_out=self.h_buf[:bs], logits_out=self.logits[:bs])
        hip.check(hip.lib.qtts_argmax(hip.ptr(self.logits), self.vocab, hip.ptr(self.tokens), bs, self.vocab,
                                      hip.stream()), "argmax")

    def _fill_inputs(self, embeds: list[torch.Tensor], n_rows: int, x_buf: torch.Tensor | None = None):
        rows = [e.reshape(-1, e.shape[-1]) for e in embeds]
        x = torch.cat(rows, 0) if len(rows) > 1 else rows[0]
        assert x.shape[0] == n_rows
        (self.x_buf if x_buf is None else x_buf)[:n_rows].copy_(x.to(torch.bfloat16), non_blocking=True)

    def _pick(self, seqs: list[Sequence], S: int) -> list[int]:
        """Greedy ids are already in self.tokens; draw for the rows that sample."""
        if any(s.do_sample for s in seqs):
            self.temps[:S].copy_(torch.tensor([s.temperature for s in seqs], dtype=torch.float32), non_blocking=True)
            hip.check(hip.lib.qtts_sample_topk(hip.ptr(self.logits), self.vocab, hip.ptr(self.temps), seqs[0].top_k,
                                               self.rng_seed, self.next_rng_offset(), hip.ptr(self.sampled), S,
                                               self.vocab, hip.stream()), "sample_topk")
            greedy, drawn = self.tokens[:S].tolist(), self.sampled[:S].tolist()
            return [d if s.do_sample else g for s, g, d in zip(seqs, greedy, drawn)]
        return self.tokens[:S].tolist()

    # ---- reference-shaped entry points ---------------------------------------------
    def prepare_prefill(self, seqs: list[Sequence], stage: Staging | None = None, x_buf: torch.Tensor | None = None):
        positions, slots, row_seq, ctx, cu = self.prefill_metadata(seqs, self.block_size)
        N, S, mb = len(positions), len(seqs), self.max_blocks_per_seq
        assert N <= self.max_rows and S <= self.max_bs
        st = stage or self.pstage
        st.h("positions")[:N] = torch.tensor(positions, dtype=torch.int64)
        st.h("slots")[:N] = torch.tensor(slots, dtype=torch.int32)
        st.h("ctx")[:N] = torch.tensor(ctx, dtype=torch.int32)
        st.h("row_seq")[:N] = torch.tensor(row_seq, dtype=torch.int32)
        tab = st.h("tables")[:S * mb].view(S, mb)
        tab.fill_(0)
        for i, s in enumerate(seqs):
            tab[i, :len(s.block_table)] = torch.tensor(s.block_table, dtype=torch.int32)
        st.upload()
        self._fill_inputs([s.input_embeds for s in seqs], N, x_buf)
        return N, S, cu, max(len(s) for s in seqs)

    def prepare_decode_talker(self, seqs: list[Sequence], bs: int):
        """position = len(seq) (one past the KV slot), slot = len(seq)-1, context = len(seq)
        (P/engine/model_runner/talker_mode_runner.py:116-140); rows >= len(seqs) are graph padding."""
        S, mb = len(seqs), self.max_blocks_per_seq
        st = self.dstage
        pos, slot, ctx = st.h("positions"), st.h("slots"), st.h("ctx")
        tab = st.h("tables")[:bs * mb].view(bs, mb)
        slot[:bs] = -1
        ctx[:bs] = 0
        pos[:bs] = 0
        tab.fill_(0)
        for i, s in enumerate(seqs):
            if s.decode_input_embeds is None:
                raise ValueError(f"Sequence {s.seq_id} has no decode_input_embeds set")
            n = len(s)
            assert n <= self.config.max_model_len, "sequence exceeds max_model_len"
            pos[i] = n
            ctx[i] = n
            slot[i] = s.block_table[-1] * self.block_size + s.last_block_num_tokens - 1
            tab[i, :len(s.block_table)] = torch.tensor(s.block_table, dtype=torch.int32)
        st.upload()
        self._fill_inputs([s.decode_input_embeds for s in seqs], S)

    @torch.inference_mode()
    def run(self, seqs: list[Sequence], is_prefill: bool):
        with self.on_stream():
            return self._run(seqs, is_prefill)

    def _run(self, seqs: list[Sequence], is_prefill: bool):
        S = len(seqs)
        if is_prefill:
            N, S, cu, max_len = self.prepare_prefill(seqs)
            st, mb = self.pstage, self.max_blocks_per_seq
            self.decoder.forward(self.x_buf[:N], st.d("positions")[:N], st.d("slots")[:N], st.d("ctx")[:N],
                                 st.d("tables")[:S * mb].view(S, mb), max_len, row_seq=st.d("row_seq")[:N],
                                 out=self.h_buf[:N])
            last = torch.tensor(cu[1:], dtype=torch.int64, device=self.device) - 1
            hidden = self.h_buf[:N].index_select(0, last)
            hip.linear(hidden, self.codec_head, self.vocab, out=self.logits[:S])
            hip.check(hip.lib.qtts_argmax(hip.ptr(self.logits), self.vocab, hip.ptr(self.tokens), S, self.vocab,
                                          hip.stream()), "argmax")
        else:
            bs = S if self.enforce_eager else next(b for b in self.graph_bs if b >= S)
            self.prepare_decode_talker(seqs, bs)
            if self.enforce_eager:
                self._decode_body(bs)
            else:
                g = self.graphs.get(bs)
                if g is None:
                    self._decode_body(bs)                      # warm-up launch outside capture
                    g = self.graphs[bs] = hip.Graph.capture(lambda: self._decode_body(bs))
                g.launch()
            hidden = self.h_buf[:S].clone()
        token_ids = self._pick(seqs, S)
        return token_ids, hidden

    # ---- prompts prefilled next to running decode steps --------------------------------
    def prefill_detached(self, seqs, hidden_out: torch.Tensor, token_out: torch.Tensor):
        """The prefill of `run(seqs, True)` (one sequence or a list, packed), enqueued on the CURRENT stream without
        any host synchronisation and without touching what the decode steps or the synchronous prefill use: it runs
        through a twin of the decoder (own workspace) with input, output, metadata and logits buffers of its own, and
        leaves the last position's post-norm hidden state of sequence i in hidden_out[i] and its first code in
        token_out[i] (int64; arg-max - a sampling request draws its first code from `self.pre_logits[i]` by the
        caller).  One such prefill may be in flight at a time (the metadata staging buffer is reused: the upload of
        the previous one is waited for here).  Same launches, same arithmetic as the synchronous path."""
        if isinstance(seqs, Sequence):
            seqs, hidden_out = [seqs], hidden_out.reshape(1, -1)
        if not hasattr(self, "_twin"):
            self._twin = self.decoder.twin(self.max_rows, self.config.max_model_len)
            H = self.shape.hidden_size
            self.pre_logits = torch.zeros(self.max_bs, self.vocab, dtype=torch.bfloat16, device=self.device)
            self.pre_x = torch.zeros(self.max_rows, H, dtype=torch.bfloat16, device=self.device)
            self.pre_h = torch.zeros(self.max_rows, H, dtype=torch.bfloat16, device=self.device)
            mb = self.max_blocks_per_seq
            self.pre_stage = Staging([("positions", self.max_rows, torch.int64), ("slots", self.max_rows, torch.int32),
                                      ("ctx", self.max_rows, torch.int32), ("row_seq", self.max_rows, torch.int32),
                                      ("tables", self.max_bs * mb, torch.int32)], self.device)
            self._pre_uploaded = torch.cuda.Event()
            self._pre_uploaded.record()
        self._pre_uploaded.synchronize()                    # the host buffer is about to be rewritten
        N, S, cu, max_len = self.prepare_prefill(seqs, self.pre_stage, self.pre_x)
        self._pre_uploaded.record()
        st, mb = self.pre_stage, self.max_blocks_per_seq
        self._twin.forward(self.pre_x[:N], st.d("positions")[:N], st.d("slots")[:N], st.d("ctx")[:N],
                           st.d("tables")[:S * mb].view(S, mb), max_len, row_seq=st.d("row_seq")[:N], out=self.pre_h[:N])
        # (row copies, not an index tensor: its upload would make the host wait for this stream)
        for i in range(S):
            hidden_out[i].copy_(self.pre_h[cu[i + 1] - 1], non_blocking=True)
        hip.linear(hidden_out[:S], self.codec_head, self.vocab, out=self.pre_logits[:S])
        hip.check(hip.lib.qtts_argmax(hip.ptr(self.pre_logits), self.vocab, hip.ptr(token_out), S, self.vocab,
                                      hip.stream()), "argmax")

    def exit(self):
        self.graphs.clear()
        super().exit()
