"""API-process side of the engine workers (reference: P/workers/client_bridge.py:35-309).

Three interchangeable back ends behind the same client methods
(`send_add_request / send_clear_request / run_step_async / send_shutdown`):
  * ZMQ: PUSH/PULL on 127.0.0.1 ports base..base+3, two spawned worker processes and a
    result-bridge thread that completes asyncio Futures — the reference's topology;
  * pipe (QTTS_WORKER_TRANSPORT=pipe): the same topology - four addresses bound here, two spawned
    worker processes that connect to theirs, a result-bridge thread - over Unix-socket
    connections of the standard library (workers/transport.py), for hosts without pyzmq;
  * in-process: the engines live in this process and a single worker thread runs the steps;
    messages still pass through the wire format, so what the API loop sees (fp32 numpy hidden
    states, plain id lists) is identical.  Used when pyzmq is absent or QTTS_INPROCESS=1.
"""
from __future__ import annotations

import asyncio
import logging
import multiprocessing as mp
import os
import threading
import uuid
from concurrent.futures import ThreadPoolExecutor

from ..zmq.utils import find_available_port
from . import protocol as P

logger = logging.getLogger(__name__)
_sampling_params_to_dict = P.sampling_params_to_dict


def _have_zmq() -> bool:
    try:
        import zmq  # noqa: F401
        return True
    except ImportError:
        return False


# ------------------------------------------------------------------ ZMQ clients
class _ZmqClient:
    def __init__(self, command_bind_addr: str, pending: dict, ready: set, loop, add_ser, step_ser):
        import zmq
        self._ctx = zmq.Context()
        self._push = self._ctx.socket(zmq.PUSH)
        self._push.setsockopt(zmq.LINGER, 0)
        self._push.bind(command_bind_addr)
        self._pending, self._ready, self._loop = pending, ready, loop
        self._add_ser, self._step_ser = add_ser, step_ser

    def send_add_request(self, request_id: str, inputs_embeds, sampling_params) -> None:
        self._push.send(self._add_ser(request_id, inputs_embeds, P.sampling_params_to_dict(sampling_params)))
        self._ready.add(request_id)

    def send_clear_request(self, request_id: str) -> None:
        self._push.send(P.serialize_clear_request(request_id))

    def run_step_async(self) -> asyncio.Future:
        step_id = str(uuid.uuid4())
        fut = self._loop.create_future()
        self._pending[step_id] = fut
        self._push.send(self._step_ser(step_id))
        return fut

    def send_shutdown(self) -> None:
        self._push.send(P.serialize_shutdown())

    def close(self) -> None:
        try:
            self._push.close()
            self._ctx.term()
        except Exception:
            pass


class TalkerWorkerClient(_ZmqClient):
    def __init__(self, command_bind_addr, pending_talker_futures, talker_ready, loop):
        super().__init__(command_bind_addr, pending_talker_futures, talker_ready, loop,
                         P.serialize_talker_add_request, P.serialize_talker_run_step)


class PredictorWorkerClient(_ZmqClient):
    def __init__(self, command_bind_addr, pending_predictor_futures, predictor_ready, loop):
        super().__init__(command_bind_addr, pending_predictor_futures, predictor_ready, loop,
                         P.serialize_predictor_add_request, P.serialize_predictor_run_step)


def _run_result_bridge_thread(talker_pull, predictor_pull, pending_talker, pending_predictor, loop, stop_event):
    import zmq
    poller = zmq.Poller()
    poller.register(talker_pull, zmq.POLLIN)
    poller.register(predictor_pull, zmq.POLLIN)
    routes = ((talker_pull, P.deserialize_talker_result, pending_talker, "talker"),
              (predictor_pull, P.deserialize_predictor_result, pending_predictor, "predictor"))
    while not stop_event.is_set():
        try:
            evts = dict(poller.poll(timeout=100))
        except Exception:
            if stop_event.is_set():
                break
            continue
        for sock, deser, pending, tag in routes:
            if sock in evts:
                try:
                    step_id, outputs_all = deser(sock.recv())
                    fut = pending.pop(step_id, None)
                    if fut is not None and not fut.done():
                        loop.call_soon_threadsafe(fut.set_result, (tag, outputs_all))
                except Exception as e:
                    logger.warning("[result_bridge] %s result: %s", tag, e)


# ------------------------------------------------------------------ pipe clients
class _PipeClient:
    """The ZMQ client's surface over a transport.PipeListener: the command channel is bound here and the worker
    connects to it (`attach` waits for that), commands are the same protocol bytes."""

    def __init__(self, command_bind_addr: str, pending: dict, ready: set, loop, add_ser, step_ser):
        from .transport import PipeListener
        self._listener = PipeListener(command_bind_addr)
        self._conn = None
        self._pending, self._ready, self._loop = pending, ready, loop
        self._add_ser, self._step_ser = add_ser, step_ser
        self._lock = threading.Lock()

    def attach(self, timeout: float):
        self._conn = self._listener.accept(timeout)

    def _send(self, data: bytes):
        with self._lock:
            self._conn.send_bytes(data)

    def send_add_request(self, request_id: str, inputs_embeds, sampling_params) -> None:
        self._send(self._add_ser(request_id, inputs_embeds, P.sampling_params_to_dict(sampling_params)))
        self._ready.add(request_id)

    def send_clear_request(self, request_id: str) -> None:
        self._send(P.serialize_clear_request(request_id))

    def run_step_async(self) -> asyncio.Future:
        step_id = str(uuid.uuid4())
        fut = self._loop.create_future()
        self._pending[step_id] = fut
        self._send(self._step_ser(step_id))
        return fut

    def send_shutdown(self) -> None:
        try:
            self._send(P.serialize_shutdown())
        except (OSError, ValueError):                       # the worker has gone already
            pass

    def close(self) -> None:
        self._listener.close()


def _run_pipe_result_bridge(talker_conn, predictor_conn, pending_talker, pending_predictor, loop, stop_event):
    """What _run_result_bridge_thread does for the ZMQ sockets: results of either worker complete the step's Future."""
    from .transport import wait
    routes = {talker_conn: (P.deserialize_talker_result, pending_talker, "talker"),
              predictor_conn: (P.deserialize_predictor_result, pending_predictor, "predictor")}
    live = list(routes)
    while not stop_event.is_set() and live:
        for conn in wait(live, timeout=0.1):
            deser, pending, tag = routes[conn]
            try:
                step_id, outputs_all = deser(conn.recv_bytes())
            except (EOFError, OSError):                     # that worker has closed its end
                live.remove(conn)
                continue
            except Exception as e:
                logger.warning("[result_bridge] %s result: %s", tag, e)
                continue
            fut = pending.pop(step_id, None)
            if fut is not None:
                loop.call_soon_threadsafe(lambda f=fut, r=(tag, outputs_all): f.done() or f.set_result(r))


def start_pipe_engines(model_path: str, request_queues: dict, queues_lock, *, gpu_memory_utilization: float = 0.9,
                       enforce_eager: bool = False, tensor_parallel_size: int = 1, engine_factories: dict | None = None,
                       start_timeout: float = 600.0):
    """start_multiprocess_engines over pipe:// addresses: bind the four channels, spawn both workers (which build
    their engines and then connect), wait for the connections, start the result bridge.
    engine_factories: {"talker": "module:callable", "predictor": ...} for stand-in engines."""
    from .predictor_worker import run_predictor_worker
    from .talker_worker import run_talker_worker
    from .transport import PipeListener, fresh_pipe_addresses
    loop = asyncio.get_event_loop()
    addr = fresh_pipe_addresses(("talker_command", "talker_result", "predictor_command", "predictor_result"))
    pending_t, pending_p, talker_ready, predictor_ready = {}, {}, set(), set()
    results = {k: PipeListener(addr[k]) for k in ("talker_result", "predictor_result")}
    tc = _PipeClient(addr["talker_command"], pending_t, talker_ready, loop, P.serialize_talker_add_request, P.serialize_talker_run_step)
    pc = _PipeClient(addr["predictor_command"], pending_p, predictor_ready, loop, P.serialize_predictor_add_request,
                     P.serialize_predictor_run_step)
    spawn = mp.get_context("spawn")
    kw = dict(gpu_memory_utilization=gpu_memory_utilization / 2.0, enforce_eager=enforce_eager,
              tensor_parallel_size=tensor_parallel_size)
    # (QTTS_TALKER_ENGINE_FACTORY / QTTS_PREDICTOR_ENGINE_FACTORY: the same through the environment, for callers that
    # reach this through Qwen3TTSInterface.start_zmq_tasks)
    ef = engine_factories or {k: os.environ.get(f"QTTS_{k.upper()}_ENGINE_FACTORY") for k in ("talker", "predictor")}
    procs = [spawn.Process(target=run_talker_worker, args=(addr["talker_command"], addr["talker_result"], model_path),
                           kwargs=dict(kw, engine_factory=ef.get("talker")), name="qtts-talker-worker"),
             spawn.Process(target=run_predictor_worker, args=(addr["predictor_command"], addr["predictor_result"], model_path),
                           kwargs=dict(kw, engine_factory=ef.get("predictor")), name="qtts-predictor-worker")]
    for p in procs:
        p.start()
    stop_event = threading.Event()

    def stop():
        stop_event.set()
        tc.send_shutdown()
        pc.send_shutdown()
        for p in procs:
            p.join(timeout=10.0)
            if p.is_alive():
                p.terminate()                               # the exact processes started above
                p.join(timeout=5.0)
        for c in (tc, pc, *results.values()):
            c.close()

    try:
        # a worker connects to its command address first, then to its result address (transport.open_worker_channels)
        tc.attach(start_timeout)
        t_res = results["talker_result"].accept(start_timeout)
        pc.attach(start_timeout)
        p_res = results["predictor_result"].accept(start_timeout)
    except Exception:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for c in (tc, pc, *results.values()):
            c.close()
        raise RuntimeError("the engine workers did not connect (see their output above)")
    bridge = threading.Thread(target=_run_pipe_result_bridge, daemon=True, name="qtts-result-bridge",
                              args=(t_res, p_res, pending_t, pending_p, loop, stop_event))
    bridge.start()
    holder = _Holder(tc, pc, talker_ready, predictor_ready, stop)
    holder.processes = procs
    return holder


# ------------------------------------------------------------------ in-process clients
class _LocalClient:
    """Same surface as the ZMQ clients; commands are executed by `serve_one` on one worker thread."""

    def __init__(self, engine, ready: set, loop, executor, tag: str):
        self._engine, self._ready, self._loop, self._exec, self._tag = engine, ready, loop, executor, tag

    def send_add_request(self, request_id: str, inputs_embeds, sampling_params) -> None:
        import torch
        cmd = P.deserialize_command(P.serialize_talker_add_request(request_id, inputs_embeds,
                                                                   P.sampling_params_to_dict(sampling_params)))
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        embeds = [torch.from_numpy(a).to(dev) for a in cmd["inputs_embeds"]]
        sp = P.sampling_params_from_dict(cmd["sampling_params"])
        self._exec.submit(self._engine.add_request, embeds, sp, request_id).result()
        self._ready.add(request_id)

    def send_clear_request(self, request_id: str) -> None:
        self._exec.submit(self._engine.clear_request, request_id).result()

    def _step(self):
        if self._tag == "talker":
            _, _, outputs_all = self._engine.step_with_outputs()
            return P.deserialize_talker_result(P.serialize_talker_result("local", outputs_all))[1]
        finished = []
        while self._engine.scheduler.waiting or self._engine.scheduler.running:
            outs, _ = self._engine.step()
            finished.extend(outs)
        return P.deserialize_predictor_result(P.serialize_predictor_result("local", finished))[1]

    def run_step_async(self) -> asyncio.Future:
        fut = self._loop.create_future()

        def work():
            try:
                res = (self._tag, self._step())
                self._loop.call_soon_threadsafe(lambda: fut.done() or fut.set_result(res))
            except Exception as e:          # surfaced to the awaiting loop, which logs and carries on
                self._loop.call_soon_threadsafe(lambda: fut.done() or fut.set_exception(e))

        self._exec.submit(work)
        return fut

    def send_shutdown(self) -> None:
        pass

    def close(self) -> None:
        pass


class _Holder:
    def __init__(self, talker_client, predictor_client, talker_ready, predictor_ready, stop):
        self.talker_client, self.predictor_client = talker_client, predictor_client
        self.talker_ready, self.predictor_ready = talker_ready, predictor_ready
        self._stop = stop

    async def stop_async(self):
        self._stop()


def start_inprocess_engines(model_path: str, request_queues: dict, queues_lock, *, gpu_memory_utilization=0.9,
                            enforce_eager=False, tensor_parallel_size=1, engine_kwargs: dict | None = None):
    from ..llm import PredictorLLM, TalkerLLM
    loop = asyncio.get_event_loop()
    kw = dict(enforce_eager=enforce_eager, tensor_parallel_size=tensor_parallel_size,
              gpu_memory_utilization=gpu_memory_utilization / 2.0)
    kw.update(engine_kwargs or {})
    executor = ThreadPoolExecutor(max_workers=1, thread_name_prefix="qtts-engine")
    talker = executor.submit(lambda: TalkerLLM(model_path, **kw)).result()
    predictor = executor.submit(lambda: PredictorLLM(model_path, **kw)).result()
    talker_ready, predictor_ready = set(), set()

    def stop():
        executor.submit(talker.exit).result()
        executor.submit(predictor.exit).result()
        executor.shutdown(wait=True)

    return _Holder(_LocalClient(talker, talker_ready, loop, executor, "talker"),
                   _LocalClient(predictor, predictor_ready, loop, executor, "predictor"),
                   talker_ready, predictor_ready, stop)


def start_multiprocess_engines(model_path: str, request_queues: dict, queues_lock, *, gpu_memory_utilization: float = 0.9,
                               enforce_eager: bool = False, tensor_parallel_size: int = 1):
    """Bind the four sockets, start the result bridge, spawn both workers (reference :182-309)."""
    transport = os.environ.get("QTTS_WORKER_TRANSPORT", "")
    if transport == "pipe" and os.environ.get("QTTS_INPROCESS") != "1":
        return start_pipe_engines(model_path, request_queues, queues_lock, gpu_memory_utilization=gpu_memory_utilization,
                                  enforce_eager=enforce_eager, tensor_parallel_size=tensor_parallel_size)
    if not _have_zmq() or os.environ.get("QTTS_INPROCESS") == "1" or transport == "inprocess":
        return start_inprocess_engines(model_path, request_queues, queues_lock,
                                       gpu_memory_utilization=gpu_memory_utilization, enforce_eager=enforce_eager,
                                       tensor_parallel_size=tensor_parallel_size)
    import zmq
    from .predictor_worker import run_predictor_worker
    from .talker_worker import run_talker_worker
    loop = asyncio.get_event_loop()
    base = find_available_port(9600, max_attempts=100)
    addr = {k: f"tcp://127.0.0.1:{base + i}" for i, k in
            enumerate(("talker_command", "talker_result", "predictor_command", "predictor_result"))}
    ctx = zmq.Context()
    pulls = {}
    for k in ("talker_result", "predictor_result"):
        s = ctx.socket(zmq.PULL)
        s.setsockopt(zmq.LINGER, 0)
        s.bind(addr[k])
        pulls[k] = s
    pending_t, pending_p, talker_ready, predictor_ready = {}, {}, set(), set()
    stop_event = threading.Event()
    threading.Thread(target=_run_result_bridge_thread, daemon=True,
                     args=(pulls["talker_result"], pulls["predictor_result"], pending_t, pending_p, loop, stop_event)).start()
    tc = TalkerWorkerClient(addr["talker_command"], pending_t, talker_ready, loop)
    pc = PredictorWorkerClient(addr["predictor_command"], pending_p, predictor_ready, loop)
    spawn = mp.get_context("spawn")
    kw = dict(gpu_memory_utilization=gpu_memory_utilization / 2.0, enforce_eager=enforce_eager,
              tensor_parallel_size=tensor_parallel_size)
    procs = [spawn.Process(target=run_talker_worker, args=(addr["talker_command"], addr["talker_result"], model_path), kwargs=kw),
             spawn.Process(target=run_predictor_worker, args=(addr["predictor_command"], addr["predictor_result"], model_path), kwargs=kw)]
    for p in procs:
        p.start()

    def stop():
        stop_event.set()
        tc.send_shutdown()
        pc.send_shutdown()
        for p in procs:
            p.join(timeout=10.0)
            if p.is_alive():
                p.terminate()
                p.join(timeout=5.0)
        tc.close()
        pc.close()
        try:
            for s in pulls.values():
                s.close()
            ctx.term()
        except Exception:
            pass

    return _Holder(tc, pc, talker_ready, predictor_ready, stop)
