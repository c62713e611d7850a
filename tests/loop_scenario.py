"""A scripted run of the two orchestrator loops (`run_talker_loop_mp`, `run_predictor_loop_mp`) against fake engine
clients: which `(engine_type, msg_type, payload)` tuples land in which request's queue, what is left in the ready
sets, how often the client was asked for a step.  `drive(module)` takes any module with those two coroutines — the
fixture generator hands in the reference's, the test this package's."""
import asyncio


class ScriptedClient:
    """`run_step_async()` returns an awaitable of `(kind, outputs)`; the outputs (or an exception to raise) come
    from the script, one entry per call; once the script is used up every call returns no outputs."""

    def __init__(self, script):
        self.script, self.calls = list(script), 0
        self.called = asyncio.Event()

    def run_step_async(self):
        i, self.calls = self.calls, self.calls + 1
        fut = asyncio.get_running_loop().create_future()
        item = self.script[i] if i < len(self.script) else []
        if isinstance(item, Exception):
            fut.set_exception(item)
        else:
            fut.set_result(("step_done", item))
        self.called.set()
        return fut


async def _until_calls(client, n, timeout=5.0):
    async def wait():
        while client.calls < n:
            client.called.clear()
            await client.called.wait()
    await asyncio.wait_for(wait(), timeout)
    await asyncio.sleep(0.01)                     # let the loop fan the step's outputs out


def _drain(queues):
    out = {}
    for rid, q in queues.items():
        msgs = []
        while not q.empty():
            msgs.append(list(q.get_nowait()))
        out[rid] = msgs
    return out


TALKER_SCRIPT = [
    [("a", 0, [5], [[0.25, -0.5]], False), ("b", 1, [7], [[1.0, 2.0]], True)],
    [("c", 2, [9], [[0.5, 0.5]], False), ("a", 0, [6], [[-1.0, 0.0]], False), ("gone", 3, [1], [[0.0, 0.0]], True)],
    RuntimeError("worker hiccup"),
    [],
    [("c", 2, [2150], [[0.0, 1.0]], True)],
]
PREDICTOR_SCRIPT = [
    [("a", 0, list(range(15))), ("b", 1, list(range(100, 115)))],
    RuntimeError("worker hiccup"),
    [("c", 2, list(range(200, 215))), ("gone", 3, list(range(15)))],
]


async def _talker(module):
    queues = {r: asyncio.Queue() for r in "abc"}
    lock, ready = asyncio.Lock(), set()
    client = ScriptedClient(TALKER_SCRIPT)
    task = asyncio.create_task(module.run_talker_loop_mp(client, queues, lock, ready))
    await asyncio.sleep(0.02)
    idle_calls = client.calls                     # nothing is ready: the worker must not be asked
    ready.update("ab")                            # c is active but has not fed the engine: the loop lingers, then runs
    await _until_calls(client, 1)
    after_first = sorted(ready)
    ready.update(["c", "a", "gone"])              # "gone" has no queue (request closed): its outputs are dropped
    await _until_calls(client, 2)
    after_second = sorted(ready)
    ready.add("c")                                # step 3 raises, step 4 returns nothing: c stays ready, step 5 serves it
    await _until_calls(client, 5)
    task.cancel()
    await asyncio.gather(task, return_exceptions=True)
    return {"idle_calls": idle_calls, "ready_after_first": after_first, "ready_after_second": after_second,
            "ready_at_end": sorted(ready), "client_calls": client.calls, "queues": _drain(queues)}


async def _predictor(module):
    queues = {r: asyncio.Queue() for r in "abc"}
    lock, ready = asyncio.Lock(), set()
    client = ScriptedClient(PREDICTOR_SCRIPT)
    task = asyncio.create_task(module.run_predictor_loop_mp(client, queues, lock, ready))
    await asyncio.sleep(0.02)
    idle_calls = client.calls
    ready.update("ab")
    await _until_calls(client, 1)
    after_first = sorted(ready)
    ready.update(["c", "gone"])
    await _until_calls(client, 3)                 # burst 2 raises, burst 3 serves c and the closed request
    task.cancel()
    await asyncio.gather(task, return_exceptions=True)
    return {"idle_calls": idle_calls, "ready_after_first": after_first, "ready_at_end": sorted(ready),
            "client_calls": client.calls, "queues": _drain(queues)}


def drive(module) -> dict:
    async def both():
        return {"talker": await _talker(module), "predictor": await _predictor(module)}
    return asyncio.run(both())
