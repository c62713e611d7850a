"""Data-parallel serving: N independent engine replicas, one process per GPU, and a router that
places every new stream on the least-loaded replica.

A stream's frame loop is sequential and the whole model fits one GPU, so the only way eight GPUs
help is by running different streams — there is no collective, no RCCL and no traffic between the
replicas; the router lives on the host.  The reference has no counterpart: it spawns its two engine
processes for ONE GPU (P/workers/client_bridge.py:182-271, `mp.get_context("spawn")`, one command and
one result channel per process — the pattern kept here) and points at `uvicorn --workers N` for more
(examples/server.py:6).

    pool = ReplicaPool(n_replicas=8, factory="nano_qwen3tts_vllm.replicas:frame_engine_factory",
                       factory_kwargs=dict(model_dir=..., max_streams=8))
    for frame in pool.generate(prompt_embeds, text_adds, max_frames=125): ...
    async for frame in pool.generate_async(...): ...
    pool.close()

A replica process pins itself to its GPU (`HIP_VISIBLE_DEVICES=<index>`, set before anything touches
a GPU, so device 0 inside the process is that GPU — which also keeps code that hard-wires device 0
working), builds its engine with `factory(**factory_kwargs)` and serves the engine protocol of
`engine.frame_engine.FrameEngine`: `add_stream / step / remove_stream / streams`.
"""
from __future__ import annotations

import asyncio
import importlib
import inspect
import multiprocessing as mp
import os
import queue
import threading
import time
import traceback
import uuid

import numpy as np

_CMD_ADD, _CMD_CANCEL, _CMD_SHUTDOWN = "add", "cancel", "shutdown"
_END = None                      # a stream's last item in its sink: (None, reason)
# What a replica sends: ONE message per frame for all its streams,
#     ("frames", replica index, [request ids], int32 array [n, 16] of their codes, [(request id, reason) of streams that ended])
# (the codes of a frame as one array: a message per stream and frame is 8 x 8 x 390 = 25 k pickled tuples a second
# through the parent's one result thread at 8 replicas of 8 streams, and 200 k at 64 streams each), and
#     ("__replica__", index, "ready" | "failed" | "stopped", info)
_MSG_FRAMES = "frames"


def frame_engine_factory(model_dir: str, max_streams: int = 8, capture: bool = True, **kw):
    """The production engine of a replica: a FrameEngine on the process's (only visible) GPU with every frame graph
    captured before the first request."""
    from .engine.frame_engine import FrameEngine
    fe = FrameEngine(model_dir, max_streams=max_streams, **kw)
    if capture:
        fe.capture_all()
    return fe


def _resolve(spec: str):
    mod, _, name = spec.partition(":")
    return getattr(importlib.import_module(mod), name)


def _to_tensor(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


def replica_main(index: int, device_index: int | None, factory: str, factory_kwargs: dict, cmd_q, res_q):
    """Body of one replica process: pin the GPU, build the engine, then loop — take commands (blocking only while
    idle), admit what fits, run one frame for every active stream, send the frames."""
    if device_index is not None:
        os.environ["HIP_VISIBLE_DEVICES"] = str(device_index)      # before any GPU call of this process
    try:
        engine = _resolve(factory)(**factory_kwargs)
    except Exception:
        res_q.put(("__replica__", index, "failed", traceback.format_exc()))
        return
    res_q.put(("__replica__", index, "ready", None))
    from .sampling_params import SamplingParams

    def takes(fn, name):
        try:
            return name in inspect.signature(fn).parameters
        except (TypeError, ValueError):
            return False
    # QTTS_ADMISSION: "join" (default; with an engine that can prefill on a second stream: requests are admitted from
    # the moment they arrive, also while a frame is in flight, and join with the next frame), "serial" (prefill
    # between two frames), "beside" (as join, but the running streams never wait for a prefill; QTTS_OVERLAP_PREFILL=1)
    admission = os.environ.get("QTTS_ADMISSION") or ("beside" if os.environ.get("QTTS_OVERLAP_PREFILL", "0") == "1" else "join")
    detached = admission != "serial" and getattr(engine, "overlapped_admission", False)
    packed = hasattr(engine, "add_streams")
    in_flight = detached and packed and takes(engine.step, "poll") and takes(engine.add_streams, "wait")
    if hasattr(engine, "join_prefills"):
        engine.join_prefills = admission != "beside"
    waiting: list = []
    state = {"running": True}

    def take_commands(block: bool):
        while True:
            try:
                cmd = cmd_q.get(block=block, timeout=0.5 if block else None)
            except queue.Empty:
                return
            block = False
            if cmd[0] == _CMD_SHUTDOWN:
                state["running"] = False
                waiting.clear()
                for sid in list(engine.streams):
                    engine.remove_stream(sid)
            elif cmd[0] == _CMD_CANCEL:
                waiting[:] = [w for w in waiting if w[1] != cmd[1]]
                engine.remove_stream(cmd[1])
            elif cmd[0] == _CMD_ADD and state["running"]:
                waiting.append(cmd)

    def admit():
        """What fits goes to the engine in one call (prompts that arrived together: one packed prefill)."""
        if not waiting:
            return
        room = max(getattr(engine, "max_streams", len(waiting)) - len(engine.streams), 0)
        batch, later = waiting[:room], waiting[room:]
        if not batch:
            return
        reqs = [dict(prompt_embeds=_to_tensor(c[2]), text_adds=_to_tensor(c[3]), max_frames=c[4],
                     talker_sampling=SamplingParams(**c[5]), predictor_sampling=SamplingParams(**c[6]), sid=c[1]) for c in batch]
        if packed and (not detached or takes(engine.add_streams, "wait")):
            results = engine.add_streams(reqs, **({"wait": False} if detached else {}))
        else:
            results = []
            for r in reqs:
                try:
                    results.append(engine.add_stream(**r, **({"wait": False} if detached else {})))
                except Exception as e:
                    results.append(e)
        still = []
        for cmd, r in zip(batch, results):
            if not isinstance(r, Exception):
                continue
            if type(r).__name__ == "NoCapacity" and engine.streams:
                still.append(cmd)                          # a row / KV blocks come back when a running stream ends
            else:
                res_q.put((_MSG_FRAMES, index, [], None, [(cmd[1], f"rejected: {r}")]))
        waiting[:] = still + later

    def poll():
        take_commands(False)
        admit()

    while state["running"] or engine.streams:
        take_commands(block=not engine.streams and not waiting)      # block only when there is nothing to run
        admit()
        if not engine.streams:
            continue
        active = list(engine.streams)
        try:
            frames = engine.step(poll=poll) if in_flight else engine.step()
        except Exception:
            err = traceback.format_exc()
            for rid in active:
                engine.remove_stream(rid)
            res_q.put((_MSG_FRAMES, index, [], None, [(rid, "engine error: " + err) for rid in active]))
            continue
        rids = [rid for rid in active + [r for r in frames if r not in active] if rid in frames]
        # codec EOS or the stream's frame budget (or cancelled meanwhile): the engine has dropped the stream
        ended = [(rid, "finished") for rid in active + [r for r in frames if r not in active] if rid not in engine.streams]
        if rids or ended:
            codes = np.asarray([frames[rid] for rid in rids], dtype=np.int32).reshape(len(rids), -1) if rids else None
            res_q.put((_MSG_FRAMES, index, rids, codes, ended))
    if hasattr(engine, "exit"):
        engine.exit()
    res_q.put(("__replica__", index, "stopped", None))


class _AsyncSink:
    """Where an async consumer's frames go: an asyncio.Queue of the consumer's event loop, filled from the result
    thread with call_soon_threadsafe (no executor thread parked per stream)."""

    def __init__(self, loop):
        self.loop, self.q = loop, asyncio.Queue()

    def put(self, item):
        try:
            self.loop.call_soon_threadsafe(self.q.put_nowait, item)
        except RuntimeError:                               # the consumer's loop has been closed
            pass


def _put_all(batch):
    for q, item in batch:
        q.put_nowait(item)


class ReplicaPool:
    """N replica processes and the host-side router: a new stream goes to the replica with the fewest streams in
    flight (lowest index on ties) and stays there; a replica at `max_streams_per_replica` takes no more until one of
    its streams ends, further requests wait in the router (first come, first served)."""

    def __init__(self, n_replicas: int, factory: str, factory_kwargs: dict | None = None, devices: list | None = None,
                 max_streams_per_replica: int = 8, start_timeout: float = 600.0):
        assert n_replicas >= 1
        self.n = n_replicas
        self.cap = max_streams_per_replica
        self._ctx = mp.get_context("spawn")               # a fresh interpreter per replica: no inherited GPU state
        self._res_q = self._ctx.Queue()
        self._cmd_qs = [self._ctx.Queue() for _ in range(n_replicas)]
        devices = devices if devices is not None else list(range(n_replicas))
        assert len(devices) == n_replicas
        self._procs = [self._ctx.Process(target=replica_main, name=f"qtts-replica-{i}", daemon=True,
                                         args=(i, devices[i], factory, dict(factory_kwargs or {}), self._cmd_qs[i], self._res_q))
                       for i in range(n_replicas)]
        self._lock = threading.Lock()
        self._load = [0] * n_replicas
        self._where: dict[str, int] = {}                   # request id -> replica
        self._sinks: dict[str, "queue.Queue"] = {}       # per request: where its frames go, until the consumer is done
        self._live: set[str] = set()                       # requests waiting in the router or running on a replica
        self._backlog: list = []
        self._dead: set[int] = set()                       # replicas whose process has gone away
        self._closed = False
        self.placements: list[tuple[str, int]] = []        # (request id, replica) in admission order
        self.frame_messages = 0                            # frame messages read from the replicas so far
        self.router_cpu_seconds = 0.0                      # CPU time the result thread has used
        for p in self._procs:
            p.start()
        ready = 0
        while ready < n_replicas:
            try:
                tag, idx, state, info = self._res_q.get(timeout=start_timeout)
            except queue.Empty:
                self.close()
                raise RuntimeError("replicas did not come up in time")
            if tag != "__replica__" or state != "ready":
                self.close()
                raise RuntimeError(f"replica {idx} failed to start:\n{info}")
            ready += 1
        self._pump = threading.Thread(target=self._pump_results, name="qtts-replica-results", daemon=True)
        self._pump.start()

    # ---------------------------------------------------------------- routing
    def _pick(self) -> int | None:
        alive = [i for i in range(self.n) if i not in self._dead]
        if not alive:
            return None
        best = min(alive, key=lambda i: (self._load[i], i))
        return best if self._load[best] < self.cap else None

    def _place_locked(self):
        while self._backlog:
            i = self._pick()
            if i is None:
                return
            msg = self._backlog.pop(0)
            rid = msg[1]
            if rid not in self._live:                      # cancelled while waiting
                continue
            self._load[i] += 1
            self._where[rid] = i
            self.placements.append((rid, i))
            self._cmd_qs[i].put(msg)

    def submit(self, prompt_embeds, text_adds, max_frames: int, talker_sampling=None, predictor_sampling=None,
               request_id: str | None = None, _sink=None) -> str:
        """Queue a stream; its frames arrive through `frames(request_id)`.  prompt_embeds [L,H], text_adds fp32 [F,H]
        (torch tensors or numpy arrays; they travel to the replica as numpy)."""
        from .sampling_params import SamplingParams
        if self._closed:
            raise RuntimeError("the replica pool is closed")
        rid = request_id or str(uuid.uuid4())
        tsp = talker_sampling or SamplingParams(temperature=1.0, max_tokens=1)
        psp = predictor_sampling or SamplingParams(temperature=0.9, max_tokens=17)
        as_np = lambda t: np.asarray(t.detach().float().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
        msg = (_CMD_ADD, rid, as_np(prompt_embeds), as_np(text_adds), int(max_frames), _sp_dict(tsp), _sp_dict(psp))
        with self._lock:
            if rid in self._sinks:
                raise ValueError(f"request id {rid!r} is already in use")
            self._sinks[rid] = _sink if _sink is not None else queue.Queue()
            self._live.add(rid)
            self._backlog.append(msg)
            self._place_locked()
        return rid

    def cancel(self, request_id: str):
        with self._lock:
            sink = self._sinks.get(request_id)
            was_live = request_id in self._live
            self._live.discard(request_id)
            i = self._where.pop(request_id, None)
            if i is not None:
                self._load[i] -= 1
                self._cmd_qs[i].put((_CMD_CANCEL, request_id))
            self._place_locked()
        if sink is not None and was_live:
            sink.put((_END, "cancelled"))

    def _pump_results(self):
        """The one thread that reads the replicas' messages: a frame message carries the codes of all streams of a
        replica; they are handed to the streams' sinks here (async consumers of one event loop in one callback), and
        the places of streams that ended are given to what waits in the router."""
        while True:
            try:
                item = self._res_q.get(timeout=0.5)
            except queue.Empty:
                if self._closed:
                    return
                self._reap()
                continue
            if item[0] != _MSG_FRAMES:
                continue
            _, _idx, rids, codes, ended = item
            self.frame_messages += 1
            self.router_cpu_seconds = time.thread_time()       # CPU time of this thread so far (load reports)
            out = []
            with self._lock:
                for k, rid in enumerate(rids):
                    if rid in self._live:                      # (cancelled: the replica's late frames are dropped)
                        sink = self._sinks.get(rid)
                        if sink is not None:
                            out.append((sink, (codes[k].tolist(), None)))
                for rid, reason in ended:
                    if rid not in self._live:
                        continue
                    self._live.discard(rid)
                    i = self._where.pop(rid, None)
                    if i is not None:
                        self._load[i] -= 1
                    sink = self._sinks.get(rid)
                    if sink is not None:
                        out.append((sink, (_END, reason)))
                if ended:
                    self._place_locked()
            by_loop: dict = {}
            for sink, payload in out:
                if isinstance(sink, _AsyncSink):
                    by_loop.setdefault(sink.loop, []).append((sink.q, payload))
                else:
                    sink.put(payload)
            for loop, batch in by_loop.items():
                try:
                    loop.call_soon_threadsafe(_put_all, batch)
                except RuntimeError:                           # that loop has been closed
                    pass

    def _reap(self):
        """A replica process that has gone away (crash, kill) takes its streams with it: they end with an error
        instead of leaving their consumers waiting, the router stops placing on it, and what waits in the router
        fails too once no replica is left."""
        ended = []
        with self._lock:
            for i, p in enumerate(self._procs):
                if i in self._dead or p.is_alive():
                    continue
                self._dead.add(i)
                for rid in [r for r, w in self._where.items() if w == i]:
                    self._where.pop(rid)
                    self._live.discard(rid)
                    ended.append((rid, f"replica {i} died (exit code {p.exitcode})"))
                self._load[i] = 0
            if len(self._dead) == self.n:
                for msg in self._backlog:
                    if msg[1] in self._live:
                        self._live.discard(msg[1])
                        ended.append((msg[1], "no replica left"))
                self._backlog.clear()
            else:
                self._place_locked()
            sinks = [(self._sinks.get(rid), reason) for rid, reason in ended]
        for sink, reason in sinks:
            if sink is not None:
                sink.put((_END, reason))

    # ---------------------------------------------------------------- consumers
    def frames(self, request_id: str, timeout: float | None = 120.0):
        """Generator over the stream's 16-id frames, in order; ends when the replica finishes the stream."""
        sink = self._sinks.get(request_id)
        if sink is None:
            return
        try:
            while True:
                try:
                    frame, reason = sink.get(timeout=timeout)
                except queue.Empty:
                    raise TimeoutError(f"stream {request_id}: no frame for {timeout} s") from None
                if frame is _END:
                    if reason and reason not in ("finished", "cancelled"):
                        raise RuntimeError(f"stream {request_id}: {reason}")
                    return
                yield frame
        finally:
            if request_id in self._live:
                self.cancel(request_id)
            self._sinks.pop(request_id, None)

    def generate(self, prompt_embeds, text_adds, max_frames: int, **kw):
        return self.frames(self.submit(prompt_embeds, text_adds, max_frames, **kw))

    async def generate_async(self, prompt_embeds, text_adds, max_frames: int, **kw):
        """Async generator of frames.  The stream's frames arrive in an asyncio.Queue of the caller's event loop, put
        there by the result thread: any number of streams may wait at once (no executor thread per stream), and a
        replica that goes away ends the stream with an error (the reaper puts the end marker)."""
        sink = _AsyncSink(asyncio.get_running_loop())
        rid = self.submit(prompt_embeds, text_adds, max_frames, _sink=sink, **kw)
        try:
            while True:
                frame, reason = await sink.q.get()
                if frame is _END:
                    if reason and reason not in ("finished", "cancelled"):
                        raise RuntimeError(f"stream {rid}: {reason}")
                    return
                yield frame
        finally:
            if rid in self._live:
                self.cancel(rid)
            self._sinks.pop(rid, None)

    def load(self) -> list[int]:
        with self._lock:
            return list(self._load)

    def close(self, timeout: float = 10.0):
        if self._closed:
            return
        self._closed = True
        for q_ in self._cmd_qs:
            try:
                q_.put((_CMD_SHUTDOWN,))
            except Exception:
                pass
        for p in self._procs:
            p.join(timeout=timeout)
            if p.is_alive():
                p.terminate()                              # the exact processes this pool started


def _sp_dict(sp) -> dict:
    return dict(temperature=sp.temperature, max_tokens=sp.max_tokens, ignore_eos=sp.ignore_eos,
                do_sample=sp.do_sample, top_k=sp.top_k, top_p=sp.top_p)
