"""CPU, real processes: the talker and predictor workers as separate OS processes (the reference's topology,
P/workers/client_bridge.py:182-271 and P/workers/talker_worker.py:106-145) over the pipe:// carrier of
workers/transport.py, with stand-in engines.  The bytes on the wire are the reference's own
(tests/golden/protocol.npz, written by oracle/gen_golden.py from the reference's protocol module)."""
import asyncio
import multiprocessing as mp

import numpy as np
import pytest
import torch

from fake_worker_engines import FakePredictor, FakeTalker
from helpers import load_npz
from nano_qwen3tts_vllm.sampling_params import SamplingParams
from nano_qwen3tts_vllm.utils import synthetic
from nano_qwen3tts_vllm.workers import protocol as P
from nano_qwen3tts_vllm.workers.predictor_worker import run_predictor_worker
from nano_qwen3tts_vllm.workers.talker_worker import run_talker_worker
from nano_qwen3tts_vllm.workers.transport import PipeListener, fresh_pipe_addresses


def _spawn(target, addr, factory):
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=target, args=(addr["command"], addr["result"], "/no/model/needed"), kwargs=dict(engine_factory=factory))
    p.start()
    return p


def test_worker_processes_answer_the_reference_bytes_and_shut_down_cleanly():
    g = load_npz("protocol.npz")
    emb = torch.from_numpy(g["emb"]).bfloat16()
    # ---- talker worker: add_request / run_step / clear_request / run_step / shutdown, byte for byte the reference's messages
    addr = fresh_pipe_addresses(("command", "result"))
    cmd, res = PipeListener(addr["command"]), PipeListener(addr["result"])
    proc = _spawn(run_talker_worker, addr, "fake_worker_engines:FakeTalker")
    c, r = cmd.accept(120), res.accept(120)
    c.send_bytes(g["talker_add"].tobytes())
    c.send_bytes(g["run_step"].tobytes())
    assert r.poll(60)
    step_id, outs = P.deserialize_talker_result(r.recv_bytes())
    ref = FakeTalker()
    ref.add_request([emb.float()], None, "req-1")                # (the wire carries the embedding widened to fp32)
    want = ref.step_with_outputs()[2][0]
    assert step_id == "step-7" and len(outs) == 1
    assert outs[0][0] == "req-1" and outs[0][2] == want[2] and outs[0][4] is False
    assert outs[0][3].dtype == np.float32 and np.array_equal(outs[0][3], want[3].float().numpy())
    c.send_bytes(g["clear"].tobytes())
    c.send_bytes(g["run_step"].tobytes())
    assert r.poll(60) and P.deserialize_talker_result(r.recv_bytes()) == ("step-7", [])       # nothing left to step
    c.send_bytes(g["shutdown"].tobytes())
    proc.join(30)
    assert proc.exitcode == 0
    cmd.close(), res.close()
    # ---- predictor worker: two requests, one burst
    addr = fresh_pipe_addresses(("command", "result"))
    cmd, res = PipeListener(addr["command"]), PipeListener(addr["result"])
    proc = _spawn(run_predictor_worker, addr, "fake_worker_engines:FakePredictor")
    c, r = cmd.accept(120), res.accept(120)
    c.send_bytes(g["predictor_add"].tobytes())
    sp = P.sampling_params_to_dict(SamplingParams(temperature=0.9, max_tokens=17))
    c.send_bytes(P.serialize_predictor_add_request("req-2", [torch.ones(1, 2, emb.shape[-1])], sp))
    c.send_bytes(P.serialize_predictor_run_step("burst-1"))
    assert r.poll(60)
    step_id, outs = P.deserialize_predictor_result(r.recv_bytes())
    ref = FakePredictor()
    ref.add_request([emb.float()], None, "req-1")
    ref.add_request([torch.ones(1, 2, emb.shape[-1])], None, "req-2")
    assert step_id == "burst-1" and [tuple(o) for o in outs] == [tuple(o) for o in ref.step()[0]]
    c.send_bytes(g["shutdown"].tobytes())
    proc.join(30)
    assert proc.exitcode == 0
    cmd.close(), res.close()


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("tiny_model"))
    synthetic.write_model_dir(d, "tiny", seed=3, embed_std=0.05)
    return d


def test_interface_over_worker_processes(model_dir, monkeypatch):
    """Qwen3TTSInterface.start_zmq_tasks with QTTS_WORKER_TRANSPORT=pipe: two spawned worker processes (stand-in
    engines), the API process's clients, result bridge and orchestrator loops in front of them; concurrent requests get
    the frames the in-process path gives, a consumer that leaves early releases its request on both workers, and
    stop_zmq_tasks ends both processes with exit code 0."""
    from test_api_surface import expected_frames
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    monkeypatch.delenv("QTTS_INPROCESS", raising=False)
    monkeypatch.setenv("QTTS_WORKER_TRANSPORT", "pipe")
    monkeypatch.setenv("QTTS_ASYNC_ENGINE", "workers")
    monkeypatch.setenv("QTTS_TALKER_ENGINE_FACTORY", "fake_worker_engines:FakeTalker")
    monkeypatch.setenv("QTTS_PREDICTOR_ENGINE_FACTORY", "fake_worker_engines:FakePredictor")
    iface = Qwen3TTSInterface.from_pretrained(model_dir)
    iface.max_frames = 5
    H = iface.model_config.talker_config.hidden_size
    gen = torch.Generator().manual_seed(1)
    prompts = [torch.randn(1, L, H, generator=gen) for L in (6, 9, 4)]
    trailing = torch.randn(1, 3, H, generator=gen)
    pad = torch.randn(1, 1, H, generator=gen)

    async def run():
        await iface.start_zmq_tasks()
        procs = iface._mp_holder.processes
        assert len(procs) == 2 and all(p.is_alive() for p in procs)

        async def one(pr):
            return [f async for f in iface.generate_async(pr, trailing, pad, None)]
        res = await asyncio.gather(*[one(pr) for pr in prompts])
        agen = iface.generate_async(prompts[0], trailing, pad, None, request_id="early")
        first = await agen.__anext__()
        await agen.aclose()
        await iface.stop_zmq_tasks()
        return res, first, procs

    res, first, procs = asyncio.run(run())
    for pr, got in zip(prompts, res):
        assert got == expected_frames(iface, pr, trailing, pad, 5)
    assert first == res[0][0]
    assert [p.exitcode for p in procs] == [0, 0]
    assert not iface._request_queues
