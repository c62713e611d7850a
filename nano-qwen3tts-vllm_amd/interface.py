"""Qwen3TTSInterface — the API object applications hold
(reference: P/interface.py:338-1366).

What this build implements of it: the decode path — the asynchronous per-frame loop (`generate_async`,
reference :1184-1308) and the synchronous in-process loop (`generate`, the reference's disabled
`_generate_caller_driven`, :1310-1366), both on the device-resident FrameEngine by default — and the prompt
front-end that turns text, instruction, speaker, language and a ready voice-clone prompt into talker input
embeddings (`generate_voice_design`, `generate_custom_voice(_async)`, `generate_voice_clone(_async)`,
reference :723-1171, on utils/generation.py + utils/prompt.py).  The text entry points need the tokenizer
files in the model directory (or a `processor` assigned by the caller) and the text tables in the
checkpoint.  The networks behind reference audio -> voice-clone prompt (speaker encoder, speech codec) and codes -> PCM
live in third-party packages: `create_voice_clone_prompt` does the audio handling and the mel front-end itself
(utils/ref_audio.py) and calls those networks through `speech_tokenizer` / `speaker_encoder`, raising
RuntimeError when they are absent.
"""
from __future__ import annotations

import asyncio
import functools
import inspect
import json
import logging
import os
import threading
import uuid
from concurrent.futures import ThreadPoolExecutor

import torch

from .config import Qwen3TTSConfig
from .sampling_params import SamplingParams

logger = logging.getLogger(__name__)

CODEC_EOS = 2150                                  # hard-coded in the reference (P/interface.py:1233)


def _estimate_model_params(c) -> int:
    hd = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
    per_layer = c.hidden_size * hd * (2 * c.num_attention_heads + 2 * c.num_key_value_heads) \
        + 3 * c.hidden_size * c.intermediate_size + 2 * c.hidden_size + 2 * hd
    return c.num_hidden_layers * per_layer + c.hidden_size + 2 * c.vocab_size * c.hidden_size


def _kv_block_bytes(c, block_size: int, dtype_bytes: int) -> int:
    hd = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
    return 2 * c.num_hidden_layers * block_size * c.num_key_value_heads * hd * dtype_bytes


def _compute_memory_split(model_path: str, gpu_memory_utilization: float, kvcache_block_size: int = 256) -> dict:
    """One user-facing utilisation number -> per-engine settings (reference :87-197): the process is capped
    at `gpu_memory_utilization` of the device, the talker claims its weights plus half of the KV blocks that
    fit, the predictor claims the rest."""
    cfg = Qwen3TTSConfig.from_model_dir(model_path)
    t, p = cfg.talker_config, cfg.talker_config.code_predictor_config
    tw = _estimate_model_params(t) * 2
    pw = _estimate_model_params(p) * 2
    total = torch.cuda.get_device_properties(0).total_memory if torch.cuda.is_available() else 64 << 30
    eff = total * gpu_memory_utilization
    t_fixed, p_fixed = int(tw * 1.5), int(pw * 1.5)
    kb_t, kb_p = _kv_block_bytes(t, kvcache_block_size, 2), _kv_block_bytes(p, kvcache_block_size, 2)
    budget = eff - t_fixed - p_fixed
    if budget <= 0:
        talker_util = max(0.05, (t_fixed + kb_t) / eff)
    else:
        n = budget / (kb_t + kb_p)
        talker_util = max(0.05, min((t_fixed + n * kb_t) / eff, 0.95))
    return {"talker_util": talker_util, "pred_util": 1.0, "predictor_util": 1.0,
            "process_gpu_memory_fraction": gpu_memory_utilization}


def _get_processor(model_path: str):
    """The tokenizer in the model directory behind the processor call signature (P/interface.py:217-222);
    None when the directory holds no tokenizer files (decode-only synthetic directories)."""
    try:
        from .processor import Qwen3TTSProcessor
        return Qwen3TTSProcessor.from_pretrained(model_path)
    except Exception as e:
        logger.info(f"[interface] no tokenizer in {model_path}: {type(e).__name__}")
        return None


class Qwen3TTSInterface:
    def __init__(self, model_path: str, enforce_eager: bool = False, tensor_parallel_size: int = 1,
                 gpu_memory_utilization: float = 0.9):
        assert tensor_parallel_size == 1, "one replica per GPU: tensor_parallel_size must be 1"
        self.model_path = model_path
        self.enforce_eager = self._enforce_eager = enforce_eager
        self.tensor_parallel_size = self._tensor_parallel_size = tensor_parallel_size
        self._gpu_memory_utilization = gpu_memory_utilization
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        from .utils.embedding_loader import load_embeddings_only
        (self.model_config, self.text_embedding, self.input_embedding, self.text_projection,
         self.predictor_input_embeddings) = load_embeddings_only(model_path, device=str(self.device))
        self.input_embedding_weight = self.input_embedding.weight
        self.predictor_embedding_weights = [t.weight for t in self.predictor_input_embeddings]
        self.processor = _get_processor(model_path)
        self.speech_tokenizer = self._load_speech_tokenizer()
        self.speaker_encoder = None
        self.talker_llm = self.predictor_llm = None
        self._use_mp_engines = True
        self._mp_holder = None
        self._request_queues: dict[str, asyncio.Queue] = {}
        self._queues_lock = asyncio.Lock()
        self._zmq_tasks: list[asyncio.Task] = []
        self._zmq_tasks_started = False
        self._frame_engine = None
        # generate_async is served either by the two worker loops of the reference's design ("workers") or by one
        # in-process loop over the frame engine, which keeps hidden states and ids on the device between the
        # talker and the predictor ("fused", the default)
        self._async_engine = os.environ.get("QTTS_ASYNC_ENGINE", "fused")
        self._fused_task: asyncio.Task | None = None
        self._fused_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="qtts-frame-engine")
        self._fused_pending: list = []
        self._fused_lock = threading.Lock()      # the pending list is filled by the event loop and emptied by the engine's thread
        self._fused_cancelled: list = []
        self._fused_wake: asyncio.Event | None = None
        self.max_streams = int(os.environ.get("QTTS_MAX_STREAMS", "8"))
        # How the fused loop admits a request.  "join" (default): its prompt is prefilled on the engine's second HIP
        # stream from the moment it is seen - also while a frame is in flight - and it joins with the next frame, which
        # waits for the rest of the prefill.  "serial": prefill between two frames on the engine's one stream.
        # "beside" (QTTS_OVERLAP_PREFILL=1 means this): as "join", but the running streams never wait - the newcomer
        # joins with the first frame after its prefill has finished.
        self._admission = os.environ.get("QTTS_ADMISSION") or ("beside" if os.environ.get("QTTS_OVERLAP_PREFILL", "0") == "1" else "join")
        if self._admission not in ("join", "serial", "beside"):
            raise ValueError(f"QTTS_ADMISSION={self._admission!r}: expected join, serial or beside")
        self.do_sample = True            # the reference always samples (T=1.0 talker, T=0.9 predictor, top-k 50)
        self.max_frames = 1024           # explicit stop: the reference has no length guard (ends only on codec EOS)
        self.ignore_eos = False          # benchmarks with random weights: keep generating through codec id 2150

    @staticmethod
    def _load_speech_tokenizer():
        """The codec wrapper applications reach through `interface.speech_tokenizer` (codes -> PCM,
        examples/quick_benchmark.py:66); None when the third-party codec package is not installed, as in the
        reference (P/interface.py:421-440).  QTTS_SPEECH_TOKENIZER names another codec directory / hub id."""
        try:
            from .utils.audio import DEFAULT_CODEC, SpeechTokenizer
            return SpeechTokenizer(os.environ.get("QTTS_SPEECH_TOKENIZER", DEFAULT_CODEC))
        except Exception as e:
            logger.info(f"[interface] speech tokenizer unavailable: {type(e).__name__}: {e}")
            return None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, *, cache_dir=None, force_download=False,
                        local_files_only=False, token=None, revision=None, enforce_eager: bool = False,
                        tensor_parallel_size: int = 1, gpu_memory_utilization: float = 0.9, **_):
        """Local directories are used as they are; a non-existent absolute path is an error; anything else
        is a hub id (reference :226-336) — there is no network here, so hub ids need a populated cache."""
        path = pretrained_model_name_or_path
        if os.path.isdir(path):
            return cls(path, enforce_eager, tensor_parallel_size, gpu_memory_utilization)
        if os.path.isabs(path):
            raise ValueError(f"model path does not exist: {path}")
        try:
            from huggingface_hub import snapshot_download
            path = snapshot_download(path, cache_dir=cache_dir, force_download=force_download,
                                     local_files_only=local_files_only, token=token, revision=revision)
        except Exception as e:
            raise ValueError(f"cannot resolve model {pretrained_model_name_or_path!r}: {e}") from e
        return cls(path, enforce_eager, tensor_parallel_size, gpu_memory_utilization)

    def shutdown(self):
        if self._frame_engine is not None:
            self._frame_engine.exit()
            self._frame_engine = None

    # ---------------------------------------------------------------- async path (workers)
    async def start_zmq_tasks(self) -> None:
        if self._zmq_tasks_started:
            return
        self._zmq_tasks_started = True
        if self._async_engine == "fused":
            fe = self._engine(self.max_streams)
            if os.environ.get("QTTS_CAPTURE_AT_START", "1") != "0":
                fe.capture_all()                     # no graph capture while streams are being served
            self._fused_wake = asyncio.Event()
            self._fused_task = asyncio.create_task(self._run_fused_loop())
            return
        from .workers.client_bridge import start_multiprocess_engines
        from .zmq.engine_loop_mp import run_predictor_loop_mp, run_talker_loop_mp
        holder = start_multiprocess_engines(self.model_path, self._request_queues, self._queues_lock,
                                            gpu_memory_utilization=self._gpu_memory_utilization,
                                            enforce_eager=self._enforce_eager,
                                            tensor_parallel_size=self._tensor_parallel_size)
        self._mp_holder = holder
        self._zmq_tasks += [
            asyncio.create_task(run_talker_loop_mp(holder.talker_client, self._request_queues, self._queues_lock,
                                                   holder.talker_ready)),
            asyncio.create_task(run_predictor_loop_mp(holder.predictor_client, self._request_queues, self._queues_lock,
                                                      holder.predictor_ready))]
        await asyncio.sleep(0.05)

    async def stop_zmq_tasks(self) -> None:
        if self._fused_task is not None:
            self._fused_task.cancel()
            await asyncio.gather(self._fused_task, return_exceptions=True)
            self._fused_task = None
            for q in self._request_queues.values():
                q.put_nowait(None)
            self._zmq_tasks_started = False
            return
        if not self._zmq_tasks:
            return
        for t in self._zmq_tasks:
            t.cancel()
        await asyncio.gather(*self._zmq_tasks, return_exceptions=True)
        self._zmq_tasks.clear()
        if self._mp_holder is not None:
            await self._mp_holder.stop_async()
            self._mp_holder = None
        self._zmq_tasks_started = False

    def _glue_next(self, last_id: int, pred_token_ids: list[int], trailing_text_hiddens, tts_pad_embed, step: int):
        """Sum of the 16 codebook embeddings plus the text / pad hidden, fp32 (reference :1276-1285)."""
        last_id_hidden = self.input_embedding(torch.tensor([last_id], device=self.device)).unsqueeze(0)
        hiddens = [last_id_hidden] + [self.predictor_input_embeddings[i](torch.tensor([pred_token_ids[i]], device=self.device)).unsqueeze(0)
                                      for i in range(len(pred_token_ids))]
        nxt = torch.cat(hiddens, dim=1).sum(1, keepdim=True)
        if step < trailing_text_hiddens.shape[1]:
            return nxt + trailing_text_hiddens[:, step].unsqueeze(1)
        return nxt + tts_pad_embed

    # ---------------------------------------------------------------- async path (one loop over the frame engine)
    async def _run_fused_loop(self) -> None:
        """Continuous batching over the frame engine: streams whose consumer went away are dropped between two frames,
        waiting requests are admitted between two frames and - with an engine whose step takes a `poll` hook - also
        while a frame is in flight (their packed prefill then starts at once on the engine's second stream and they
        join with the next frame); every frame goes to its request's queue, `None` ends a stream (codec EOS, frame
        budget, or no free row / KV block)."""
        from .engine.frame_engine import NoCapacity
        fe = self._frame_engine
        budget: dict[str, int] = {}
        # The engine's blocking calls (a frame: about 3 ms until the codes are on the host; an admission: the prompt's
        # prefill) run on one worker thread, so the event loop keeps serving its other coroutines - an HTTP server's
        # handlers, the streams' consumers - meanwhile.  One call at a time talks to the engine.
        loop = asyncio.get_running_loop()

        def off_loop(fn, *a, **k):
            return loop.run_in_executor(self._fused_pool, functools.partial(fn, *a, **k))

        def takes(fn, name):
            try:
                return name in inspect.signature(fn).parameters
            except (TypeError, ValueError):
                return False
        packed = hasattr(fe, "add_streams")
        # admission while a frame is in flight needs the hook and a prefill that does not wait for the host
        in_flight = self._admission != "serial" and packed and takes(fe.step, "poll") and takes(fe.add_streams, "wait")
        if hasattr(fe, "join_prefills"):
            fe.join_prefills = self._admission != "beside"
        detached = self._admission != "serial" and getattr(fe, "overlapped_admission", False)
        admitted: list = []              # (requests, results) of engine admissions, settled by this coroutine

        def admit():
            """What can be admitted now goes to the engine in ONE call: prompts that arrived together are prefilled
            packed (the reference's scheduler does the same with what its collection window gathers).  Runs on the
            worker thread, between two frames or from the engine's hook while one is in flight; the batch leaves the
            pending list first, requests that arrive meanwhile are appended behind."""
            if not self._fused_pending:
                return
            if admitted and any(isinstance(r, Exception) for r in admitted[-1][1]):
                return                    # a request the engine has just refused for lack of room is not retried at once
            room = max(fe.max_streams - len(fe.streams), 0)
            with self._fused_lock:
                batch = self._fused_pending[:room]
                del self._fused_pending[:room]
            if not batch:
                return
            if packed:
                kw = {"wait": False} if (detached and takes(fe.add_streams, "wait")) else {}
                results = fe.add_streams([dict(prompt_embeds=p_, text_adds=t_, max_frames=n, talker_sampling=tsp,
                                               predictor_sampling=psp, sid=rid) for rid, p_, t_, n, tsp, psp in batch], **kw)
            else:
                results = []
                for rid, p_, t_, n, tsp, psp in batch:
                    try:
                        results.append(fe.add_stream(p_, t_, max_frames=n, talker_sampling=tsp, predictor_sampling=psp, sid=rid,
                                                     wait=not detached))
                    except (NoCapacity, AssertionError, ValueError) as e:
                        results.append(e)
            # (whether the engine was serving anything is judged now: by the time this is settled a stream may have ended)
            admitted.append((batch, results, bool(fe.streams)))

        def settle():
            """Books what `admit` did (on this coroutine).  Returns the streams whose consumer went away during their
            prefill: the caller hands them to the worker thread, the only one that talks to the engine."""
            retry, orphans = [], []
            for batch, results, busy in admitted:
                for req, r in zip(batch, results):
                    rid = req[0]
                    if not isinstance(r, Exception):
                        if rid in self._request_queues:
                            budget[rid] = req[3]
                        else:                                             # its consumer went away during the prefill
                            orphans.append(rid)
                        continue
                    if isinstance(r, NoCapacity) and busy:               # blocks come back when a running stream ends
                        retry.append(req)
                        continue
                    logger.warning(f"[fused_loop] request {rid[:8]} rejected: {r}")     # can never be served
                    q = self._request_queues.get(rid)
                    if q is not None:
                        q.put_nowait(None)
            admitted.clear()
            with self._fused_lock:
                self._fused_pending[:0] = [r for r in retry if r[0] in self._request_queues]
            return orphans

        def drop(rids):
            """Hands streams back to the engine (worker thread: remove_stream may wait for a prefill that is still
            writing the stream's KV blocks, and the engine is only ever touched from that thread)."""
            for rid in rids:
                fe.remove_stream(rid)

        while True:
            gone = list(self._fused_cancelled)
            del self._fused_cancelled[:len(gone)]
            for rid in gone:
                budget.pop(rid, None)
            await off_loop(admit)
            gone += settle()
            if gone:
                await off_loop(drop, gone)
            if not fe.streams:
                self._fused_wake.clear()
                if not self._fused_pending:
                    await self._fused_wake.wait()
                continue
            active = list(fe.streams)
            res = await (off_loop(fe.step, poll=admit) if in_flight else off_loop(fe.step))
            gone = settle()               # requests admitted while the frame was in flight: budgets before frames
            for rid in active + [r for r in res if r not in active]:
                q = self._request_queues.get(rid)
                frame = res.get(rid)
                if frame is not None and q is not None:
                    q.put_nowait(frame)
                    budget[rid] -= 1
                # A stream has ended when the engine has dropped it (codec EOS, its frame budget) or its budget here is
                # used up - not when a frame brought nothing for it: a stream whose prefill still runs beside the
                # frames (or waits behind another group's) is in `fe.streams` but takes no part in the frame yet.
                if rid not in fe.streams or budget.get(rid, 0) <= 0:
                    if rid in fe.streams:
                        gone.append(rid)
                    budget.pop(rid, None)
                    if q is not None:
                        q.put_nowait(None)
            if gone:
                await off_loop(drop, gone)

    async def _generate_async_fused(self, inputs_embeds, trailing_text_hiddens, tts_pad_embed, request_id):
        fe = self._frame_engine
        prompt = inputs_embeds.reshape(-1, inputs_embeds.shape[-1])
        n = min(self.max_frames, fe.tr.config.max_model_len - prompt.shape[0] - 2)
        text = trailing_text_hiddens.reshape(-1, trailing_text_hiddens.shape[-1]).float()
        pad = tts_pad_embed.reshape(1, -1).float().to(text.device)
        text_adds = torch.cat([text, pad], 0)
        tsp = SamplingParams(temperature=1.0, max_tokens=1, do_sample=self.do_sample)
        psp = SamplingParams(temperature=0.9, max_tokens=17, do_sample=self.do_sample)
        q: asyncio.Queue = asyncio.Queue()
        async with self._queues_lock:
            self._request_queues[request_id] = q
        with self._fused_lock:
            self._fused_pending.append((request_id, prompt, text_adds, n, tsp, psp))
        self._fused_wake.set()
        try:
            while True:
                frame = await q.get()
                if frame is None:
                    return
                yield frame
        finally:
            async with self._queues_lock:
                self._request_queues.pop(request_id, None)
            with self._fused_lock:
                self._fused_pending[:] = [r for r in self._fused_pending if r[0] != request_id]
            if request_id in fe.streams:
                self._fused_cancelled.append(request_id)
                self._fused_wake.set()

    async def generate_async(self, inputs_embeds: torch.Tensor, trailing_text_hiddens: torch.Tensor,
                             tts_pad_embed: torch.Tensor, talker_attention_mask: torch.Tensor | None = None,
                             request_id: str | None = None):
        """Async generator of 16-id frames.  `await start_zmq_tasks()` first."""
        if self._fused_task is not None:
            async for frame in self._generate_async_fused(inputs_embeds, trailing_text_hiddens, tts_pad_embed,
                                                          request_id or str(uuid.uuid4())):
                yield frame
            return
        if not (self._use_mp_engines and self._mp_holder is not None):
            raise RuntimeError("generate_async requires start_zmq_tasks() to be called first")
        tsp = SamplingParams(temperature=1.0, max_tokens=1, do_sample=self.do_sample)
        psp = SamplingParams(temperature=0.9, max_tokens=17, do_sample=self.do_sample)
        request_id = request_id or str(uuid.uuid4())
        q: asyncio.Queue = asyncio.Queue()
        async with self._queues_lock:
            self._request_queues[request_id] = q
        tc, pc = self._mp_holder.talker_client, self._mp_holder.predictor_client
        try:
            nxt = inputs_embeds if inputs_embeds.dim() == 3 else inputs_embeds.unsqueeze(0)
            trailing_text_hiddens = trailing_text_hiddens.to(self.device).float()
            tts_pad_embed = tts_pad_embed.to(self.device).float()
            step = 0
            tc.send_add_request(request_id, [nxt], tsp)
            while step < self.max_frames:
                engine_type, msg_type, payload = await q.get()
                if engine_type == "talker" and msg_type == "done":
                    break
                if not (engine_type == "talker" and msg_type == "token"):
                    continue
                last_id = payload["token_ids"][-1]
                if last_id == CODEC_EOS and not self.ignore_eos:
                    break
                last_id_hidden = self.input_embedding(torch.tensor([last_id], device=self.device)).unsqueeze(0)
                h = torch.as_tensor(payload["hidden_states"]).to(self.device).float().reshape(1, 1, -1)
                pc.send_add_request(request_id, [torch.cat((h, last_id_hidden), dim=1)], psp)
                _, _, payload2 = await q.get()
                pred_ids = payload2.get("token_ids", [])
                yield [last_id] + pred_ids
                nxt = self._glue_next(last_id, pred_ids, trailing_text_hiddens, tts_pad_embed, step)
                step += 1
                tc.send_add_request(request_id, [nxt], tsp)
        finally:
            try:
                tc.send_clear_request(request_id)
                pc.send_clear_request(request_id)
            except Exception:
                pass
            async with self._queues_lock:
                self._request_queues.pop(request_id, None)

    # ---------------------------------------------------------------- sync path (in-process frame engine)
    def _engine(self, max_streams: int = 8):
        if self._frame_engine is None:
            from .engine.frame_engine import FrameEngine
            self._frame_engine = FrameEngine(self.model_path, max_streams=max_streams, enforce_eager=self._enforce_eager,
                                             gpu_memory_utilization=self._gpu_memory_utilization / 2.0,
                                             codec_eos=None if self.ignore_eos else CODEC_EOS)
        return self._frame_engine

    def generate(self, inputs_embeds: torch.Tensor, trailing_text_hiddens: torch.Tensor, tts_pad_embed: torch.Tensor,
                 talker_attention_mask: torch.Tensor | None = None, request_id: str | None = None,
                 max_frames: int | None = None):
        """Synchronous generator of 16-id frames for one request, everything between frames on the device."""
        return self._generate_caller_driven(inputs_embeds, trailing_text_hiddens, tts_pad_embed,
                                            request_id or str(uuid.uuid4()),
                                            SamplingParams(temperature=1.0, max_tokens=1, do_sample=self.do_sample),
                                            SamplingParams(temperature=0.9, max_tokens=17, do_sample=self.do_sample),
                                            max_frames=max_frames)

    def _generate_caller_driven(self, inputs_embeds, trailing_text_hiddens, tts_pad_embed, request_id,
                                talker_sampling_params, predictor_sampling_params, max_frames: int | None = None):
        fe = self._engine()
        prompt = inputs_embeds.reshape(-1, inputs_embeds.shape[-1])
        budget = fe.tr.config.max_model_len - prompt.shape[0] - 2
        n = min(max_frames or self.max_frames, budget)
        text = trailing_text_hiddens.reshape(-1, trailing_text_hiddens.shape[-1]).float()
        pad = tts_pad_embed.reshape(1, -1).float().to(text.device)
        text_adds = torch.cat([text, pad], 0)                     # last row (pad) repeats after the text runs out
        sid = fe.add_stream(prompt, text_adds, max_frames=n, talker_sampling=talker_sampling_params,
                            predictor_sampling=predictor_sampling_params, sid=request_id)
        try:
            for _ in range(n):
                res = fe.step()
                if sid not in res:
                    return
                yield res[sid]
        finally:
            fe.remove_stream(sid)

    # ---------------------------------------------------------------- text front-end
    def _need_front_end(self, what: str):
        if self.processor is None:
            raise RuntimeError(f"{what}: no tokenizer found in {self.model_path}; put the tokenizer files there or "
                               "assign `interface.processor` (callable as processor(text=..., return_tensors='pt', padding=True))")
        if self.text_embedding is None or self.text_projection is None:
            raise RuntimeError(f"{what}: the checkpoint in {self.model_path} has no text embedding / projection tables")

    def _prep_custom_voice(self, text, language, speaker, instruct=None, non_streaming_mode=True):
        from .utils.generation import prepare_inputs
        from .utils.prompt import prepare_custom_voice_prompt
        input_ids, instruct_ids, speakers, languages = prepare_custom_voice_prompt(
            text=[text], speaker=[speaker], language=[language], instruct=None if instruct is None else [instruct],
            processor=self.processor, device=self.device)
        return prepare_inputs(config=self.model_config, input_ids=input_ids, instruct_ids=instruct_ids, languages=languages,
                              speakers=None if speaker == "" else speakers, non_streaming_mode=non_streaming_mode,
                              text_embedding=self.text_embedding, input_embedding=self.input_embedding,
                              text_projection=self.text_projection, device=self.device)

    def _prep_voice_clone(self, text, language, ref_audio, ref_text, x_vector_only_mode, voice_clone_prompt, non_streaming_mode):
        """Reference :761-842.  The prompt dict holds ref_code [T,16] or None, ref_spk_embedding [D],
        x_vector_only_mode, icl_mode and optionally ref_text."""
        from .utils.generation import generate_icl_prompt, generate_speaker_prompt, prepare_inputs
        from .utils.prompt import _build_assistant_text, _build_ref_text, _tokenize_texts
        if voice_clone_prompt is None:
            if ref_audio is None:
                raise ValueError("Either `voice_clone_prompt` or `ref_audio` must be provided.")
            voice_clone_prompt = self.create_voice_clone_prompt(ref_audio=ref_audio, ref_text=ref_text,
                                                                x_vector_only_mode=x_vector_only_mode)
            ref_text_for_ids = ref_text
        elif voice_clone_prompt.get("icl_mode", False):
            ref_text_for_ids = ref_text if ref_text is not None else voice_clone_prompt.get("ref_text")
            if ref_text_for_ids is None:
                raise ValueError("ICL mode is enabled in voice_clone_prompt but ref_text is not available. Please provide "
                                 "ref_text when creating voice_clone_prompt or when calling generate_voice_clone.")
        else:
            ref_text_for_ids = None
        input_ids = _tokenize_texts([_build_assistant_text(text)], self.processor, self.device)
        ref_ids = None
        if ref_text_for_ids:
            ref_ids = [_tokenize_texts([_build_ref_text(ref_text_for_ids)], self.processor, self.device)[0]]
        lists = {k: [voice_clone_prompt[k]] for k in ("ref_code", "ref_spk_embedding", "x_vector_only_mode", "icl_mode")}

        def icl_fn(**kw):
            return generate_icl_prompt(config=self.model_config, text_embedding=self.text_embedding,
                                       input_embedding=self.input_embedding, text_projection=self.text_projection,
                                       code_predictor_embeddings=self.predictor_input_embeddings, device=self.device, **kw)
        return prepare_inputs(config=self.model_config, input_ids=input_ids, ref_ids=ref_ids, voice_clone_prompt=lists,
                              languages=[language or "Auto"], non_streaming_mode=non_streaming_mode,
                              text_embedding=self.text_embedding, input_embedding=self.input_embedding,
                              text_projection=self.text_projection, device=self.device,
                              generate_speaker_prompt_fn=lambda p: generate_speaker_prompt(p, self.device),
                              generate_icl_prompt_fn=icl_fn)

    def _run_sync(self, prepared):
        embeds, trailing, pad, _mask = prepared
        return self._generate_caller_driven(embeds, trailing, pad, str(uuid.uuid4()),
                                            SamplingParams(temperature=1.0, max_tokens=1, do_sample=self.do_sample),
                                            SamplingParams(temperature=0.9, max_tokens=17, do_sample=self.do_sample))

    def generate_custom_voice(self, text: str, language: str = "English", speaker: str = "Vivian"):
        """Generator of 16-id frames for a built-in speaker (reference :1121-1143; its sync form is disabled there
        in favour of the async one — both work here)."""
        self._need_front_end("generate_custom_voice")
        return self._run_sync(self._prep_custom_voice(text, language, speaker))

    async def generate_custom_voice_async(self, text: str, language: str = "English", speaker: str = "Vivian"):
        if not self._zmq_tasks_started:
            raise RuntimeError("generate_custom_voice_async requires start_zmq_tasks() to be called first")
        self._need_front_end("generate_custom_voice_async")
        embeds, trailing, pad, mask = self._prep_custom_voice(text, language, speaker)
        async for frame in self.generate_async(embeds, trailing, pad, mask):
            yield frame

    def generate_voice_design(self, text: str, instruct: str, language: str | None = None, non_streaming_mode: bool = True):
        """Voice described by a natural-language instruction, no speaker embedding (reference :1013-1079)."""
        self._need_front_end("generate_voice_design")
        return self._run_sync(self._prep_custom_voice(text, language or "Auto", "", instruct, non_streaming_mode))

    def generate_voice_clone(self, text: str, language: str | None = None, ref_audio=None, ref_text=None,
                             x_vector_only_mode: bool = False, voice_clone_prompt=None, non_streaming_mode: bool = True):
        self._need_front_end("generate_voice_clone")
        return self._run_sync(self._prep_voice_clone(text, language, ref_audio, ref_text, x_vector_only_mode,
                                                     voice_clone_prompt, non_streaming_mode))

    async def generate_voice_clone_async(self, text: str, language: str | None = None, ref_audio=None, ref_text=None,
                                         x_vector_only_mode: bool = False, voice_clone_prompt=None,
                                         non_streaming_mode: bool = True):
        if not self._zmq_tasks_started:
            raise RuntimeError("generate_voice_clone_async requires start_zmq_tasks() to be called first")
        self._need_front_end("generate_voice_clone_async")
        embeds, trailing, pad, mask = self._prep_voice_clone(text, language, ref_audio, ref_text, x_vector_only_mode,
                                                             voice_clone_prompt, non_streaming_mode)
        async for frame in self.generate_async(embeds, trailing, pad, mask):
            yield frame

    # ---------------------------------------------------------------- reference audio -> voice-clone prompt
    def _load_speaker_encoder(self):
        """The speaker encoder is a module of the third-party `qwen_tts` model class (the reference builds the whole
        model on the CPU to take it out, P/interface.py:442-470).  Callers that hold it already assign
        `interface.speaker_encoder` (any module: mel [1, frames, 128] -> [1, D])."""
        if self.speaker_encoder is not None:
            return self.speaker_encoder
        try:
            from qwen_tts.core.models import Qwen3TTSForConditionalGeneration
        except ImportError as e:
            raise RuntimeError("Speaker encoder not available for this model: the `qwen_tts` package is not installed "
                               "and no `interface.speaker_encoder` was assigned") from e
        whole = Qwen3TTSForConditionalGeneration.from_pretrained(self.model_path, device_map="cpu")
        enc = getattr(whole, "speaker_encoder", None)
        if enc is None:
            raise RuntimeError("Model does not have speaker_encoder")
        enc = enc.to(self.device)
        if hasattr(whole, "dtype"):
            enc = enc.to(whole.dtype)
        self.speaker_encoder = enc
        return enc

    @torch.inference_mode()
    def extract_speaker_embedding(self, audio, sr: int) -> torch.Tensor:
        """24 kHz waveform -> speaker vector [D]: 128-band mel spectrogram (1024-point transform, hop 256, 0-12 kHz)
        through the speaker encoder in its own dtype (P/interface.py:558-589)."""
        from .utils.ref_audio import MEL_ARGS, SPEAKER_SR, mel_spectrogram
        assert sr == SPEAKER_SR, "Only support 24kHz audio"
        enc = self._load_speaker_encoder()
        mels = mel_spectrogram(torch.as_tensor(audio, dtype=torch.float32).unsqueeze(0), **MEL_ARGS).transpose(1, 2)
        dtype = next(enc.parameters()).dtype
        return enc(mels.to(self.device).to(dtype))[0]

    def create_voice_clone_prompt(self, ref_audio, ref_text=None, x_vector_only_mode: bool = False):
        """Reference clip (path / URL / base64 / (waveform, sample_rate)) -> the dict generate_voice_clone takes:
        ref_code [T,16] from the speech codec's encoder (None in x-vector-only mode), ref_spk_embedding [D],
        x_vector_only_mode, icl_mode, ref_text (P/interface.py:670-721)."""
        from .utils.ref_audio import SPEAKER_SR, normalize_audio_inputs, resample
        if self.speech_tokenizer is None:
            raise RuntimeError("speech_tokenizer not available. Cannot create voice clone prompt.")
        if not x_vector_only_mode and not ref_text:
            raise ValueError("ref_text is required when x_vector_only_mode=False (ICL mode).")
        wav, sr = normalize_audio_inputs([ref_audio])[0]
        ref_code = self.speech_tokenizer.tokenizer.encode(wav, sr=sr).audio_codes[0].cpu()       # [time, 16]
        spk = self.extract_speaker_embedding(resample(wav, sr, SPEAKER_SR), SPEAKER_SR)
        return {"ref_code": None if x_vector_only_mode else ref_code, "ref_spk_embedding": spk,
                "x_vector_only_mode": bool(x_vector_only_mode), "icl_mode": not x_vector_only_mode, "ref_text": ref_text}
