"""In-process frame loop with everything between two frames kept on the device.

The reference glues talker and predictor per frame in the API process
(P/interface.py:1241-1290; the in-process form survives as dead code at :1322-1366):
hidden state and ids travel GPU -> host -> pickle -> host -> GPU twice per frame and
the next talker input is summed on the host.  Here one captured hipGraph per frame does

    predictor input  = [hidden ; codec_embedding(token0)]
    15-step predictor burst (pick + token feedback on the device)
    next talker input = sum of the 16 codebook embeddings + text/pad hidden   (fp32 -> bf16)
    advance positions / slots / context lengths
    talker decode step + codec head + pick                                   (-> hidden, token0 of the next frame)

and the host only reads back the 16 ids per stream.  Streams occupy fixed rows;
KV blocks for a stream's whole lifetime are reserved at admission (a stream needs
at most 4 blocks), which also removes the reference's preemption path that cannot
work for embedding-fed sequences.
"""
from __future__ import annotations

import contextlib
import functools
import os
import time

import torch

from .. import hip
from ..llm import PredictorLLM, TalkerLLM
from ..sampling_params import SamplingParams
from .guards import SingleCaller
from .sequence import Sequence

CODEC_EOS = 2150        # hard-coded in the reference's API loop (P/interface.py:1233)


def _single_caller(fn):
    """A public engine call: one host thread at a time (engine/guards.py)."""
    @functools.wraps(fn)
    def inner(self, *a, **kw):
        with self._one_caller:
            return fn(self, *a, **kw)
    return inner


class NoCapacity(RuntimeError):
    """No free row or not enough free KV blocks right now; the request can be admitted later."""


class Stream:
    def __init__(self, sid, row, seq, n_text, max_frames):
        self.sid, self.row, self.seq, self.n_text, self.max_frames = sid, row, seq, n_text, max_frames
        self.frames: list[list[int]] = []
        self.finished = False
        self.active = True            # False while the stream's prefill is still running beside the frames
        self.ready = None             # the event that prefill ends with
        self.row_written = False      # the row's per-stream state has been queued (_write_row)


class FrameEngine:
    overlapped_admission = True       # add_stream(wait=False) prefills beside the running frames
    # When a stream whose prompt is prefilled beside the frames joins the batch.  True: with the next frame - the
    # engine's stream waits for the rest of the prefill (the running streams' frame is late by what is left of it, the
    # newcomer has its first frame as early as it can).  False: with the first frame that starts after the prefill has
    # finished (the running streams never wait; the newcomer may lose most of a frame).
    join_prefills = True
    # step(poll=): a frame is handed to the GPU this long before the talker step in front of it is expected to end.
    # The margin covers the host's wake-up granularity and the time it takes to get the frame's first graph into the
    # queue; the latter is measured per frame (`_commit_s`) and the margin follows it between the two bounds, so a
    # host that shares its cores with other replicas commits earlier instead of leaving the GPU idle.  An instance
    # that sets LAUNCH_MARGIN_S itself keeps that value.
    LAUNCH_MARGIN_S = None
    LAUNCH_MARGIN_MIN_S, LAUNCH_MARGIN_MAX_S = 2.5e-4, 1.0e-3
    WAKE_UP_S = 1.0e-4                 # what a 50 us sleep can overshoot by, plus the poll hook's own time

    def __init__(self, model_dir: str, max_streams: int = 8, max_text_frames: int = 1024, talker_weights=None,
                 predictor_weights=None, enforce_eager: bool = False, codec_eos: int | None = CODEC_EOS, top_k: int = 50,
                 **kw):
        kw = dict(kw)
        self._one_caller = SingleCaller("FrameEngine")     # held by every public call (see engine/guards.py)
        self.top_k = top_k
        kw.setdefault("max_num_seqs", max(max_streams, 1))
        tkw = dict(kw, enforce_eager=enforce_eager)
        if talker_weights is not None:
            tkw["weights"] = talker_weights
        pkw = dict(kw, enforce_eager=enforce_eager, num_kvcache_blocks=max_streams)
        if predictor_weights is not None:
            pkw["weights"] = predictor_weights
        self.talker = TalkerLLM(model_dir, **tkw)
        self.pred = PredictorLLM(model_dir, **pkw)
        self.tr, self.pr = self.talker.model_runner, self.pred.model_runner
        self.device = self.tr.device
        self.max_streams = max_streams
        self.enforce_eager = enforce_eager
        self.codec_eos = codec_eos
        H = self.tr.shape.hidden_size
        self.H = H
        self.max_text_frames = max_text_frames
        dev = self.device
        self.cur_hidden = torch.zeros(max_streams, H, dtype=torch.bfloat16, device=dev)
        self.cur_token = torch.zeros(max_streams, dtype=torch.int64, device=dev)
        self.next_x = torch.zeros(max_streams, H, dtype=torch.bfloat16, device=dev)
        self.codes = torch.zeros(max_streams, 16, dtype=torch.int64, device=dev)
        self.text_add = torch.zeros(max_streams, max_text_frames, H, dtype=torch.float32, device=dev)
        self.frame_idx = torch.zeros(max_streams, dtype=torch.int32, device=dev)
        # per row: the context length its reserved KV blocks cover; the device-side metadata advance stops there
        self.ctx_limit = torch.zeros(max_streams, dtype=torch.int32, device=dev)
        self.rng = torch.zeros(1, dtype=torch.int64, device=dev)
        self.temps_t = torch.ones(max_streams, dtype=torch.float32, device=dev)
        self.temps_p = torch.ones(max_streams, dtype=torch.float32, device=dev)
        # greedy flag per row and per model: a request may sample the talker and take the predictor's arg-max or
        # the reverse (SamplingParams.do_sample of the two requests the reference sends, P/interface.py:1195-1196)
        self.greedy_t = torch.ones(max_streams, dtype=torch.uint8, device=dev)
        self.greedy_p = torch.ones(max_streams, dtype=torch.uint8, device=dev)
        self.ptab_ptrs = torch.tensor([t.data_ptr() for t in self.pr.tables], dtype=torch.int64, device=dev)
        # a prompt prefilled beside the running frames (add_stream(wait=False)) leaves its results here until the
        # stream is switched on between two frames: the frame graphs rewrite cur_hidden / cur_token of every row
        self.pend_hidden = torch.zeros(max_streams, H, dtype=torch.bfloat16, device=dev)
        self.pend_token = torch.zeros(max_streams, dtype=torch.int64, device=dev)
        self._pre_hidden = torch.zeros(max_streams, H, dtype=torch.bfloat16, device=dev)
        self._pre_token = torch.zeros(max_streams, dtype=torch.int64, device=dev)
        self._pre_drawn = torch.zeros(max_streams, dtype=torch.int64, device=dev)
        self._pre_temps = torch.ones(max_streams, dtype=torch.float32, device=dev)
        self.pre_stream = torch.cuda.Stream(device=dev)          # prefill beside the frames
        self._prefill_running: list | None = None                 # the streams of the one detached prefill in flight
        self._prefill_waiting: list = []                          # groups of streams admitted, prefill not started yet
        # a detached prefill is ordered behind this event of the engine's stream, not behind everything queued there:
        # it marks the last moment KV blocks or rows were handed back (a finished stream's last talker step may still
        # be writing its blocks) and the end of the last prefill that ran between two frames
        self._fence = torch.cuda.Event()
        self._fence.record(self.tr.stream)
        self.rows: list[Stream | None] = [None] * max_streams
        self.streams: dict[str, Stream] = {}
        self.graphs: dict[tuple, hip.Graph] = {}            # talker part, per (batch bucket, context bucket)
        self.code_graphs: dict[int, hip.Graph] = {}          # predictor burst + glue, per batch bucket
        self._codes_host = torch.zeros(max_streams, 16, dtype=torch.int64).pin_memory()
        self._codes_event = torch.cuda.Event()
        # step(poll=): when the talker step in flight began (host clock) with its timing events, the events of the
        # one before it, and the running estimate of a talker step's duration
        self._talker_mark = None
        self._talker_pair = None
        self._talker_s = 0.0
        self._codes_s = 0.0                # running estimate: launch of a frame -> its codes on the host (step(poll=))
        self._commit_s = 2.0e-4            # running estimate: end of the wait -> the frame's codes graph is in the queue
        # host time inside step() that was not spent waiting (enqueueing the frame's graphs, reading the codes, stream
        # bookkeeping), and the number of step() calls it was collected over: what a replica costs its host per frame
        self.host_busy_s, self.host_frames = 0.0, 0
        self._pred_meta_bs = None
        self._next_sid = 0
        # values of per-row metadata writes on their way to the device (_sync_row_to_device)
        self._ring = torch.zeros(512, (4 + self.tr.max_blocks_per_seq + 1) & ~1, dtype=torch.int32).pin_memory()     # (8-byte rows)
        self._ring_next = 0
        self.stream = self.tr.stream
        # Run every kernel of the frame once, eagerly, with all rows inactive (context 0, slot -1:
        # nothing is stored, nothing advances) so that no code object is first loaded under capture.
        with self.tr.on_stream():
            self._setup_predictor_meta(1)
            self._frame_body(1, 128)
            self._frame_body(1, 256)
            self.rng.zero_()
            self.tr.stream.synchronize()

    # ---------------------------------------------------------------- admission
    @_single_caller
    def add_stream(self, prompt_embeds: torch.Tensor, text_adds: torch.Tensor, max_frames: int,
                   talker_sampling: SamplingParams | None = None, predictor_sampling: SamplingParams | None = None,
                   sid: str | None = None, wait: bool = True) -> str:
        """prompt_embeds [L,H]; text_adds fp32 [F,H]: what the API adds to the codec-embedding sum at
        frame f (trailing text hiddens, then the tts pad embedding; the last row repeats).
        wait=True: the prompt is prefilled before this returns and the stream takes part in the next frame.
        wait=False (serving loops): the row and the KV blocks are reserved and the call returns; the prefill runs on a
        second HIP stream BESIDE the frames of the running streams instead of between two of them; when the stream
        joins the batch is the engine's `join_prefills` policy (see step).  One prefill at a time, in arrival order.
        Same launches, same ids either way.
        Raises NoCapacity (no free row / KV blocks now: try again when a stream has ended) or ValueError (can never
        be served: too long, wrong shapes, a stream id that is in use)."""
        s = self._reserve(prompt_embeds, text_adds, max_frames, talker_sampling, predictor_sampling, sid)
        if wait:
            self._prefill_now([s])
        else:
            self._prefill_detached([s])
        return s.sid

    @_single_caller
    def add_streams(self, requests: list[dict], wait: bool = True) -> list:
        """Admits several requests at once (dicts of add_stream's arguments, without `wait`): their prompts are
        packed into ONE prefill (all rows of all prompts through each GEMM together - the reference's scheduler packs
        the prompts that arrive within its collection window the same way, P/engine/scheduler.py:31-56), which costs
        about half of prefilling them one after the other.  wait=False: that packed prefill runs beside the frames, as
        add_stream(wait=False).  Returns, per request and in order, the stream id or the exception (NoCapacity /
        ValueError) that add_stream would have raised; the ids a stream then produces do not depend on what it was
        packed with."""
        out, fresh = [], []
        for r in requests:
            try:
                s = self._reserve(r["prompt_embeds"], r["text_adds"], r["max_frames"], r.get("talker_sampling"),
                                  r.get("predictor_sampling"), r.get("sid"))
                fresh.append(s)
                out.append(s.sid)
            except (NoCapacity, ValueError) as e:
                out.append(e)
        if fresh and wait:
            self._prefill_now(fresh)
        elif fresh:
            self._prefill_detached(fresh)
        return out

    def _prefill_detached(self, streams: list[Stream]):
        """Queues the packed prefill of freshly reserved streams for the second HIP stream (groups within the
        runner's row limit) and starts it if none is running."""
        groups, rows = [[]], 0
        for s in streams:
            if groups[-1] and rows + s.prompt_len > self.tr.max_rows:
                groups.append([])
                rows = 0
            groups[-1].append(s)
            rows += s.prompt_len
        for s in streams:
            s.active = False
        self._prefill_waiting.extend(groups)
        self._pump_prefills()                  # the prompt's kernels are queued first: everything below is off their path
        with self._on_engine_stream():
            for s in streams:
                self._write_row(s)             # (read when the stream is switched on, at the start of a later step)

    @contextlib.contextmanager
    def _on_engine_stream(self):
        """The engine's stream made current, ordered after the caller's stream - but, unlike ModelRunner.on_stream, the
        caller's stream is NOT made to wait for the engine's afterwards: what is enqueued here is consumed on the
        engine's stream only, and a caller's stream that waited for every frame queued there could no longer feed a
        prefill that is meant to run beside those frames."""
        outer = torch.cuda.current_stream(self.device)
        self.tr.stream.wait_stream(outer)
        with torch.cuda.stream(self.tr.stream):
            yield

    def _reserve(self, prompt_embeds, text_adds, max_frames, talker_sampling, predictor_sampling, sid) -> Stream:
        """Checks the request, takes a row and the KV blocks of the stream's whole lifetime."""
        tsp = talker_sampling or SamplingParams(temperature=1.0, max_tokens=1)
        psp = predictor_sampling or SamplingParams(temperature=0.9, max_tokens=17)
        # A request's own top_k is carried but not used, as in the reference: its sampler cuts at its default of 50
        # whatever SamplingParams.top_k says (P/layers/sampler.py:15, P/engine/model_runner/base.py:236-241 hands the
        # sampler temperatures only).  Every sampling row of this engine uses the engine's top_k (50 unless the engine
        # was built with another).
        L = prompt_embeds.shape[-2]
        total = L + max_frames + 1
        if total > self.tr.config.max_model_len:
            raise ValueError(f"prompt of {L} tokens + {max_frames} frames exceed max_model_len={self.tr.config.max_model_len}")
        if prompt_embeds.shape[-1] != self.H or text_adds.dim() != 2 or text_adds.shape[-1] != self.H or text_adds.shape[0] < 1:
            raise ValueError(f"prompt_embeds [L,{self.H}] and text_adds [F>=1,{self.H}] expected, got "
                             f"{tuple(prompt_embeds.shape)} and {tuple(text_adds.shape)}")
        if L > self.tr.max_rows:
            raise ValueError(f"prompt of {L} tokens: the engine was built for at most {self.tr.max_rows} prompt rows per step")
        if sid is not None and sid in self.streams:
            # (a second stream under the same id would take over the dict entry: the first one's row and KV blocks
            # could never be handed back)
            raise ValueError(f"stream id {sid!r} is already in use")
        row = next((i for i, s in enumerate(self.rows) if s is None), None)
        if row is None:
            raise NoCapacity("all stream rows are taken")
        bm = self.talker.scheduler.block_manager
        need = (total + 255) // 256
        if len(bm.free_block_ids) < need:
            raise NoCapacity(f"{need} KV blocks needed, {len(bm.free_block_ids)} free")
        sid = sid or f"stream-{self._next_sid}"
        self._next_sid += 1
        seq = Sequence([], input_embeds=prompt_embeds.reshape(1, L, -1), sampling_params=tsp, request_id=sid)
        seq.block_table = [bm._take() for _ in range(need)]
        s = Stream(sid, row, seq, min(text_adds.shape[0], self.max_text_frames), max_frames)
        s.tsp, s.psp, s.text_adds, s.prompt_len, s.need = tsp, psp, text_adds, L, need
        self.rows[row] = s
        self.streams[sid] = s
        return s

    def _write_row(self, s: Stream):
        """The row's per-stream state (on the engine's stream): text additions, sampling, frame counter, context limit
        (the talker's decode metadata follows when the stream is switched on, _sync_row_to_device)."""
        row, F, text_adds = s.row, s.n_text, s.text_adds
        self.text_add[row, :F].copy_(text_adds[:F].to(self.device, torch.float32), non_blocking=True)
        if F < self.max_text_frames:
            self.text_add[row, F:] = text_adds[F - 1].to(self.device, torch.float32)
        s.text_adds = None
        self.frame_idx[row] = 0
        self.ctx_limit[row] = s.prompt_len + s.max_frames      # the talker step that closes frame max_frames - 1
        self.temps_t[row] = s.tsp.temperature
        self.temps_p[row] = s.psp.temperature
        self.greedy_t[row] = 0 if s.tsp.do_sample else 1
        self.greedy_p[row] = 0 if s.psp.do_sample else 1
        s.row_written = True

    def _prefill_now(self, streams: list[Stream]):
        """Prefills the prompts of freshly reserved streams between two frames, packed into as few runner steps as the
        runner's row limit allows, and switches the streams on."""
        groups, rows = [[]], 0
        for s in streams:
            if groups[-1] and rows + s.prompt_len > self.tr.max_rows:
                groups.append([])
                rows = 0
            groups[-1].append(s)
            rows += s.prompt_len
        with self.tr.on_stream():
            for group in groups:
                toks, hidden = self.tr._run([s.seq for s in group], True)      # eager path of the runner
                for i, s in enumerate(group):
                    s.seq.append_token(toks[i], None)
                    s.first_token = toks[i]
                    self._write_row(s)
                    self.cur_hidden[s.row].copy_(hidden[i])
                    self.cur_token[s.row] = toks[i]
                    self._sync_row_to_device(s.row, s)
            self._fence.record(self.tr.stream)

    # ---------------------------------------------------------------- prefill beside the frames
    def _start_prefill(self, group: list[Stream]):
        tr = self.tr
        self.pre_stream.wait_stream(torch.cuda.current_stream(self.device))      # the prompts' producer
        self.pre_stream.wait_event(self._fence)            # the blocks' and rows' earlier tenants have left
        with torch.cuda.stream(self.pre_stream):
            # (no host-built index tensors here: their upload would make the host wait for this stream)
            n = len(group)
            tr.prefill_detached([s.seq for s in group], self._pre_hidden[:n], self._pre_token[:n])
            if any(s.tsp.do_sample for s in group):
                # rows that sample draw their first code from their logits row (the arg-max rows keep theirs)
                for i, s in enumerate(group):
                    self._pre_temps[i] = s.tsp.temperature
                hip.check(hip.lib.qtts_sample_topk(hip.ptr(tr.pre_logits), tr.vocab, hip.ptr(self._pre_temps), self.top_k,
                                                   tr.rng_seed, tr.next_rng_offset(),      # (the offset _pick would take: same first code either way)
                                                   hip.ptr(self._pre_drawn), n, tr.vocab, hip.stream()),
                          "sample_topk")
            for i, s in enumerate(group):
                self.pend_hidden[s.row].copy_(self._pre_hidden[i], non_blocking=True)
                src = self._pre_drawn if s.tsp.do_sample else self._pre_token
                self.pend_token[s.row:s.row + 1].copy_(src[i:i + 1], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.pre_stream)
        for s in group:
            s.seq.append_token(-1, None)                  # the first code itself stays on the device (pend_token)
            s.ready = ready
        self._prefill_running = group

    def _pump_prefills(self, block: bool = False):
        """Switches streams on once their prefill has finished (between two frames: called by step / run_frames before
        they launch) and starts the next waiting prefill.  block: do not ask whether the running prefill has finished
        but make the engine's stream wait for it (the host does not wait)."""
        while True:
            group = self._prefill_running
            if group is not None:
                ready = group[0].ready
                live = [s for s in group if self.streams.get(s.sid) is s]       # (the others were cancelled meanwhile)
                if not live:
                    ready.synchronize()
                    self._prefill_running = None
                    continue
                if not (block or ready.query()) or not all(s.row_written for s in live):
                    return
                with self._on_engine_stream():
                    self.tr.stream.wait_event(ready)
                    for s in live:
                        self.cur_hidden[s.row].copy_(self.pend_hidden[s.row], non_blocking=True)
                        self.cur_token[s.row:s.row + 1].copy_(self.pend_token[s.row:s.row + 1], non_blocking=True)
                        self._sync_row_to_device(s.row, s)
                        s.active = True
                self._prefill_running = None
                block = False
            self._prefill_waiting = [g for g in ([s for s in grp if self.streams.get(s.sid) is s] for grp in self._prefill_waiting) if g]
            if not self._prefill_waiting:
                return
            self._start_prefill(self._prefill_waiting.pop(0))

    def _activate_or_wait(self) -> bool:
        """Before a frame: admit what is ready; when no stream is active yet but prefills are under way, wait for the
        first.  Returns whether any stream is active."""
        self._pump_prefills(block=self.join_prefills)
        while not any(s.active for s in self.streams.values()):
            if self._prefill_running is None and not self._prefill_waiting:
                return False
            self._pump_prefills(block=True)
        return True

    def _sync_row_to_device(self, row, s: Stream | None):
        """The talker's decode metadata of a row, written on the current stream: a stream that is switched on gets
        prepare_decode_talker's values for prompt + first code, one step back (advance_decode adds 1 before the first
        decode step); a row that is handed back gets context 0 / slot -1 (nothing is stored, nothing advances).
        Every call takes a fresh slot of a pinned ring for its values.  (A copy from pinned memory may read its source
        only when it executes, behind whatever frames are queued; with one staging entry per row the host could by then
        have written the row's next tenant into it, whose context length would switch the row on before its prefill.
        The runtime in use was not seen to read that late for copies this small, but nothing here should depend on it;
        a slot is reused after 512 calls, several seconds of serving later.)"""
        st, mb = self.tr.dstage, self.tr.max_blocks_per_seq
        k = self._ring_next
        self._ring_next = (k + 1) % self._ring.shape[0]
        slot = self._ring[k]                                 # int32 [4 + mb]: position (int64 as two words), ctx, slot, table
        n1 = 0 if s is None else s.prompt_len                # = len(prompt + first code) - 1
        slot[0], slot[1], slot[2], slot[3] = n1, 0, n1, -1
        for j in range(mb):
            slot[4 + j] = s.seq.block_table[j] if (s is not None and j < s.need) else 0
        st.d("positions")[row:row + 1].copy_(slot[0:2].view(torch.int64), non_blocking=True)
        st.d("ctx")[row:row + 1].copy_(slot[2:3], non_blocking=True)
        st.d("slots")[row:row + 1].copy_(slot[3:4], non_blocking=True)
        if s is not None:
            st.d("tables")[row * mb:(row + 1) * mb].copy_(slot[4:4 + mb], non_blocking=True)

    @_single_caller
    def remove_stream(self, sid: str):
        s = self.streams.pop(sid, None)
        if s is None:
            return
        if self._prefill_running is not None and s in self._prefill_running:
            s.ready.synchronize()                    # its prefill still writes the KV blocks that are about to be freed
        self._prefill_waiting = [g for g in ([w for w in grp if w is not s] for grp in self._prefill_waiting) if g]
        self.rows[s.row] = None
        with self._on_engine_stream():
            self._sync_row_to_device(s.row, None)
            self._fence.record(self.tr.stream)       # whoever gets the row or the blocks next starts behind this
        self.talker.scheduler.block_manager.deallocate(s.seq)

    # ---------------------------------------------------------------- the frame
    def _bucket(self):
        hi = max((i for i, s in enumerate(self.rows) if s is not None), default=-1) + 1
        return next(b for b in self.tr.graph_bs if b >= max(hi, 1))

    def _setup_predictor_meta(self, bs):
        """Row r of the predictor batch always uses predictor KV block r, so the burst's metadata is
        a constant of the bucket size."""
        if self._pred_meta_bs == bs:
            return
        pr, st = self.pr, self.pr.bstage
        n_gen, B = pr.n_gen, 256
        n0 = 2 * bs
        pos, slot, ctx, rs = st.h("positions"), st.h("slots"), st.h("ctx"), st.h("row_seq")
        i = torch.arange(bs)
        for k in (0, 1):                                   # rows (2i, 2i+1): the two prompt positions of sequence i
            rs[k:n0:2] = i
            pos[k:n0:2] = k
            slot[k:n0:2] = i * B + k
            ctx[k:n0:2] = k + 1
        for step in range(1, n_gen):                       # then bs rows per decode step
            r0 = n0 + (step - 1) * bs
            pos[r0:r0 + bs] = step + 1
            slot[r0:r0 + bs] = i * B + step + 1
            ctx[r0:r0 + bs] = step + 2
        mb = pr.max_blocks_per_seq
        tab = st.h("tables")[:bs * mb].view(bs, mb)
        tab.fill_(0)
        tab[:, 0] = torch.arange(bs, dtype=torch.int32)
        st.upload()
        self._pred_meta_bs = bs

    def _frame_body(self, bs: int, max_context: int):
        """One whole frame: the 16 codes of this frame, then the talker step that starts the next one."""
        self._codes_part(bs)
        self._talker_part(bs, max_context)

    def _codes_part(self, bs: int):
        """Predictor input, 15-step burst, glue: after this the frame's 16 codes (and the next talker input) exist."""
        tr, pr, lib, s = self.tr, self.pr, hip.lib, hip.stream
        H = self.H
        hip.check(lib.qtts_predictor_input(hip.ptr(self.cur_hidden), hip.ptr(tr.codec_embedding), hip.ptr(self.cur_token),
                                           hip.ptr(pr.e_buf), bs, H, tr.vocab, s()), "predictor_input")
        pr._greedy_bytes, pr._top_k = self.greedy_p, self.top_k
        pr._burst_temps, pr._burst_rng = self.temps_p, self.rng
        pr._uniform_rows = os.environ.get("QTTS_UNIFORM_BURST", "1") == "1"     # row r = predictor KV block r (_setup_predictor_meta)
        pr._burst_body(bs)
        hip.check(lib.qtts_frame_glue(hip.ptr(tr.codec_embedding), hip.ptr(self.ptab_ptrs), hip.ptr(self.cur_token),
                                      hip.ptr(pr.burst_tokens), bs, hip.ptr(self.text_add),
                                      self.max_text_frames * H, hip.ptr(self.frame_idx), hip.ptr(self.next_x),
                                      hip.ptr(self.codes), bs, H, tr.vocab, pr.vocab, s()), "frame_glue")

    def _talker_part(self, bs: int, max_context: int):
        """Metadata advance, talker decode step + codec head, pick: hidden state and first code of the next frame."""
        tr, lib, s = self.tr, hip.lib, hip.stream
        st, mb = tr.dstage, tr.max_blocks_per_seq
        hip.check(lib.qtts_advance_decode(hip.ptr(st.d("positions")), hip.ptr(st.d("slots")), hip.ptr(st.d("ctx")),
                                          hip.ptr(st.d("tables")), mb, hip.ptr(self.frame_idx), hip.ptr(self.rng),
                                          hip.ptr(self.ctx_limit), tr.config.max_model_len, bs, s()), "advance_decode")
        tr.decoder.forward_logits(self.next_x[:bs], st.d("positions")[:bs], st.d("slots")[:bs], st.d("ctx")[:bs],
                                  st.d("tables")[:bs * mb].view(bs, mb), max_context, tr.codec_head, tr.vocab,
                                  hidden_out=self.cur_hidden[:bs], logits_out=tr.logits[:bs])
        hip.check(lib.qtts_pick_token(hip.ptr(tr.logits), tr.vocab, hip.ptr(self.temps_t), hip.ptr(self.greedy_t),
                                      self.top_k, tr.rng_seed, 32, hip.ptr(self.rng), hip.ptr(self.cur_token), 1, bs,
                                      tr.vocab, s()), "pick_token")

    def _max_ctx_bucket(self, extra_frames: int = 1) -> int:
        mx = max((len(s.seq) + len(s.frames) + extra_frames for s in self.rows if s is not None), default=1)
        mx = min(mx + 1, self.tr.config.max_model_len)
        return min(((mx + 127) // 128) * 128, self.tr.config.max_model_len)

    def _launch_codes(self, bs):
        if self.enforce_eager:
            self._codes_part(bs)
            return
        g = self.code_graphs.get(bs)
        if g is None:
            g = self.code_graphs[bs] = hip.Graph.capture(lambda: self._codes_part(bs))
        g.launch()

    def _launch_talker(self, bs, max_ctx):
        if self.enforce_eager:
            self._talker_part(bs, max_ctx)
            return
        key = (bs, max_ctx)
        g = self.graphs.get(key)
        if g is None:
            g = self.graphs[key] = hip.Graph.capture(lambda: self._talker_part(bs, max_ctx))
        g.launch()

    @_single_caller
    def capture_all(self, max_context: int | None = None) -> int:
        """Captures the graphs of every (batch bucket, context bucket) this engine can meet — one codes graph per
        batch bucket, one talker graph per (batch, context) bucket — so that no capture happens while streams are
        being served (the reference captures its decode graphs at start-up too,
        P/engine/model_runner/talker_mode_runner.py:157-192).  Returns the number of talker graphs held."""
        if self.enforce_eager:
            return 0
        top = min(max_context or self.tr.config.max_model_len, self.tr.config.max_model_len)
        with self.tr.on_stream():
            for bs in (b for b in self.tr.graph_bs if b <= self._bucket_for(self.max_streams)):
                self._setup_predictor_meta(bs)
                if bs not in self.code_graphs:
                    self.code_graphs[bs] = hip.Graph.capture(lambda: self._codes_part(bs))
                for ctx in range(128, top + 1, 128):
                    if (bs, ctx) not in self.graphs:
                        self.graphs[(bs, ctx)] = hip.Graph.capture(lambda: self._talker_part(bs, ctx))
            self.tr.stream.synchronize()
        return len(self.graphs)

    def _bucket_for(self, n: int) -> int:
        return next(b for b in self.tr.graph_bs if b >= max(n, 1))

    @_single_caller
    def run_frames(self, n: int) -> torch.Tensor:
        """Generate n frames for every active stream back to back (no EOS handling: the caller asked
        for exactly n; a stream with fewer than n frames left of its max_frames is an error, its KV blocks were
        reserved for max_frames).  Returns ids [n, bucket, 16] on the host; one synchronisation at the end."""
        self._pump_prefills(block=True)
        while self._prefill_running is not None or self._prefill_waiting:
            self._pump_prefills(block=True)
        for s in self.streams.values():
            if len(s.frames) + n > s.max_frames:
                raise ValueError(f"stream {s.sid}: {n} more frames exceed its max_frames={s.max_frames} "
                                 f"({len(s.frames)} generated)")
        bs = self._bucket()
        out = torch.empty(n, bs, 16, dtype=torch.int64).pin_memory()
        with self.tr.on_stream():
            self._setup_predictor_meta(bs)
            for f in range(n):
                self._launch_codes(bs)
                out[f].copy_(self.codes[:bs], non_blocking=True)
                self._launch_talker(bs, self._max_ctx_bucket(extra_frames=f + 1))
            self.tr.stream.synchronize()
        for s in self.rows:
            if s is not None:
                s.frames.extend(out[:, s.row].tolist())
        return out

    @_single_caller
    def step(self, poll=None) -> dict[str, list[int]]:
        """One frame for every active stream; streams whose talker id is the codec EOS finish
        without yielding that frame (as P/interface.py:1233-1239); a stream that has produced its max_frames
        finishes with that frame (the explicit length guard the reference lacks).
        poll: called (with no arguments; at least once, then every 50 us or so) while the host has nothing to do but
        wait; a serving loop admits what has arrived meanwhile from it with add_stream(s)(wait=False), so a prompt's
        prefill starts beside the frame in flight instead of after the next time the loop gets round to looking.
        With poll the frame is also handed to the GPU as late as it can be without leaving it idle (see below)."""
        t_in = time.perf_counter()
        waited = 0.0                     # of this call: time the host spent waiting (for the GPU, or asleep between polls)
        if poll is not None and self._talker_mark is not None:
            # The talker step that closed the previous frame is still running (about a millisecond) and this frame's
            # codes cannot start before it ends: the frame is not committed to the GPU's queue until shortly before
            # then.  A request that shows up in this window is prefilled right behind that talker step and is part of
            # this frame; had the frame been launched at once, the prefill would have queued behind the whole of it.
            begun, pair = self._talker_mark
            deadline = begun + self._talker_s - self.launch_margin_s()
            while time.perf_counter() < deadline and not pair[1].query():
                poll()
                t_w = time.perf_counter()
                time.sleep(self._poll_sleep(deadline - t_w))
                waited += time.perf_counter() - t_w
        self._talker_mark = None
        t_commit = time.perf_counter()
        if not self.streams or not self._activate_or_wait():
            return {}
        bs = self._bucket()
        live = [s for s in self.streams.values() if s.active]      # (poll may admit and even switch on others meanwhile)
        pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if poll is not None else None
        with self._on_engine_stream():
            self._setup_predictor_meta(bs)
            # the frame's codes are final once the predictor burst is done: they travel to the host while the talker
            # step that starts the next frame is already running (the reference yields before that step too,
            # P/interface.py:1241-1290)
            self._launch_codes(bs)
            if poll is not None:
                self._commit_s = 0.75 * self._commit_s + 0.25 * min(time.perf_counter() - t_commit, 2e-3)
            self._codes_host[:bs].copy_(self.codes[:bs], non_blocking=True)
            self._codes_event.record(torch.cuda.current_stream())
            if pair:
                pair[0].record(torch.cuda.current_stream())
            self._launch_talker(bs, self._max_ctx_bucket())
            if pair:
                pair[1].record(torch.cuda.current_stream())
        if poll is None:
            t_w = time.perf_counter()
            self._codes_event.synchronize()
            waited += time.perf_counter() - t_w
        else:
            # the codes are due one predictor burst after the launch (tracked like the talker step's duration)
            due = time.perf_counter() + self._codes_s
            t_launch = time.perf_counter()
            while True:
                poll()
                if self._codes_event.query():
                    break
                t_w = time.perf_counter()
                time.sleep(self._poll_sleep(due - t_w))
                waited += time.perf_counter() - t_w
            self._codes_s = 0.5 * self._codes_s + 0.5 * (time.perf_counter() - t_launch)
            # the codes are here, so this frame's talker step has just begun; how long it runs is taken from the
            # previous frame's (finished by now: this frame's codes came after it)
            if self._talker_pair is not None:
                self._talker_s = 0.5 * self._talker_s + 0.5 * self._talker_pair[0].elapsed_time(self._talker_pair[1]) * 1e-3
            self._talker_pair = pair
            self._talker_mark = (time.perf_counter(), pair)
        codes = self._codes_host[:bs].tolist()
        res = {}
        for s in live:
            if self.streams.get(s.sid) is not s:
                continue
            c = codes[s.row]
            if self.codec_eos is not None and c[0] == self.codec_eos:
                s.finished = True
                self.remove_stream(s.sid)
                continue
            s.frames.append(c)
            res[s.sid] = c
            if len(s.frames) >= s.max_frames:
                s.finished = True
                self.remove_stream(s.sid)
        self.host_busy_s += time.perf_counter() - t_in - waited
        self.host_frames += 1
        return res

    def launch_margin_s(self) -> float:
        """How long before the expected end of the talker step in front of it a frame is committed to the GPU's queue."""
        if self.LAUNCH_MARGIN_S is not None:
            return self.LAUNCH_MARGIN_S
        return min(self.LAUNCH_MARGIN_MAX_S, max(self.LAUNCH_MARGIN_MIN_S, self._commit_s + self.WAKE_UP_S))

    @staticmethod
    def _poll_sleep(remaining: float) -> float:
        """How long the host sleeps between two looks at the GPU (and calls of the poll hook) while it waits for an
        event expected `remaining` seconds from now: 50 us when the event is about due - the wake-up then follows it
        closely - and up to 200 us while it is still far away, so that a waiting replica leaves its core mostly idle
        (eight replicas share one host) and a request that arrives meanwhile is still noticed within 0.2 ms."""
        return 5e-5 if remaining < 1.5e-4 else min(2e-4, remaining - 1e-4)

    def exit(self):
        self.graphs.clear()
        self.code_graphs.clear()
        self.talker.exit()
        self.pred.exit()
