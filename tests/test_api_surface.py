"""CPU: wire protocol against the reference's own bytes, worker command loops, the asyncio
orchestrator loops and Qwen3TTSInterface.generate_async — driven with deterministic fake
engines (no GPU, no pyzmq)."""
import asyncio
import os
import queue

import numpy as np
import pytest
import torch

from helpers import load_npz
from nano_qwen3tts_vllm.sampling_params import SamplingParams
from nano_qwen3tts_vllm.utils import synthetic
from nano_qwen3tts_vllm.workers import predictor_worker, protocol as P, talker_worker


# ------------------------------------------------------------------ protocol
def test_protocol_matches_reference_bytes():
    g = load_npz("protocol.npz")
    emb = torch.from_numpy(g["emb"]).bfloat16()
    hid = torch.from_numpy(g["hid"]).bfloat16()
    sp = {"temperature": 0.9, "max_tokens": 17, "ignore_eos": False, "do_sample": True, "top_k": 50, "top_p": 1.0}
    mine = {
        "talker_add": P.serialize_talker_add_request("req-1", [emb], sp),
        "predictor_add": P.serialize_predictor_add_request("req-1", [emb.float()], sp),
        "run_step": P.serialize_talker_run_step("step-7"),
        "clear": P.serialize_clear_request("req-1"),
        "shutdown": P.serialize_shutdown(),
        "talker_result": P.serialize_talker_result("step-7", [("req-1", 3, [11, 12], hid, False)]),
        "predictor_result": P.serialize_predictor_result("step-8", [("req-1", 4, list(range(15)))]),
    }
    for k, b in mine.items():
        assert b == g[k].tobytes(), k          # byte-identical to what the reference emits
    cmd = P.deserialize_command(g["talker_add"].tobytes())
    assert cmd["cmd"] == "add_request" and cmd["request_id"] == "req-1" and cmd["sampling_params"] == sp
    assert cmd["inputs_embeds"][0].dtype == np.float32 and np.array_equal(cmd["inputs_embeds"][0], g["emb"])
    step_id, outs = P.deserialize_talker_result(g["talker_result"].tobytes())
    assert step_id == "step-7" and outs[0][:3] == ("req-1", 3, [11, 12]) and np.array_equal(outs[0][3], g["hid"])
    assert P.sampling_params_to_dict(P.sampling_params_from_dict(sp)) == sp


# ------------------------------------------------------------------ fake engines
from fake_worker_engines import FakePredictor, FakeTalker  # noqa: E402


def test_worker_command_loops():
    """serve(): add_request / run_step / clear_request / shutdown over an in-memory transport."""
    inbox, outbox = queue.Queue(), queue.Queue()
    t = FakeTalker()
    sp = P.sampling_params_to_dict(SamplingParams(temperature=1.0, max_tokens=1))
    e = torch.ones(1, 4, 8)
    for msg in (P.serialize_talker_add_request("a", [e], sp), P.serialize_talker_run_step("s1"),
                P.serialize_talker_add_request("a", [e[:, :1] * 2], sp), P.serialize_talker_run_step("s2"),
                P.serialize_clear_request("a"), P.serialize_talker_run_step("s3"), P.serialize_shutdown()):
        inbox.put(msg)
    assert talker_worker.serve(t, inbox.get, outbox.put, device="cpu") == 3
    r1, r2, r3 = (P.deserialize_talker_result(outbox.get()) for _ in range(3))
    assert r1[0] == "s1" and r1[1][0][2] == [32000 % 500] and r1[1][0][3].dtype == np.float32
    assert r2[1][0][2] == [32000 % 500, 16000 % 500]
    assert r3 == ("s3", []) and t.cleared == ["a"]

    p = FakePredictor()
    inbox.put(P.serialize_predictor_add_request("a", [torch.ones(1, 2, 8)], sp))
    inbox.put(P.serialize_predictor_add_request("b", [torch.ones(1, 2, 8) * 2], sp))
    inbox.put(P.serialize_predictor_run_step("b1"))
    inbox.put(P.serialize_shutdown())
    assert predictor_worker.serve(p, inbox.get, outbox.put, device="cpu") == 1
    sid, outs = P.deserialize_predictor_result(outbox.get())
    assert sid == "b1" and [o[0] for o in outs] == ["a", "b"] and all(len(o[2]) == 15 for o in outs)

    class Boom(FakeTalker):
        def step_with_outputs(self):
            raise RuntimeError("boom")
    inbox.put(P.serialize_talker_run_step("s9"))
    with pytest.raises(RuntimeError):
        talker_worker.serve(Boom(), inbox.get, outbox.put, device="cpu")
    assert P.deserialize_talker_result(outbox.get()) == ("s9", [])       # an empty answer precedes the crash


# ------------------------------------------------------------------ interface + orchestrator loops
@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("tiny_model"))
    synthetic.write_model_dir(d, "tiny", seed=3, embed_std=0.05)
    return d


def expected_frames(iface, prompt, trailing, pad, n):
    """The loop of P/interface.py:1210-1299 written out against the fake engines' arithmetic."""
    t, p = FakeTalker(), FakePredictor()
    frames = []
    x = prompt
    for step in range(n):
        t.add_request([x], None, "r")
        _, _, outs = t.step_with_outputs()
        tok, hid = outs[0][2][-1], outs[0][3].float()
        emb = iface.input_embedding_weight[tok].cpu()
        p.add_request([torch.stack([hid, emb]).unsqueeze(0)], None, "r")
        ptoks = p.step()[0][0][2]
        frames.append([tok] + ptoks)
        s = torch.stack([emb] + [iface.predictor_embedding_weights[i][ptoks[i]].cpu() for i in range(15)]).unsqueeze(0).sum(1, keepdim=True)
        x = s + (trailing[:, step].unsqueeze(1) if step < trailing.shape[1] else pad)
    return frames


def test_interface_generate_async_with_inprocess_clients(model_dir, monkeypatch):
    import nano_qwen3tts_vllm.llm as llm
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    monkeypatch.setenv("QTTS_INPROCESS", "1")
    monkeypatch.setenv("QTTS_ASYNC_ENGINE", "workers")     # the reference's two-engine orchestration (the fused loop needs a GPU)
    monkeypatch.setattr(llm, "TalkerLLM", FakeTalker)
    monkeypatch.setattr(llm, "PredictorLLM", FakePredictor)
    iface = Qwen3TTSInterface.from_pretrained(model_dir)
    iface.max_frames = 5
    H = iface.model_config.talker_config.hidden_size
    gen = torch.Generator().manual_seed(1)
    prompts = [torch.randn(1, L, H, generator=gen) for L in (6, 9, 4)]
    trailing = torch.randn(1, 3, H, generator=gen)
    pad = torch.randn(1, 1, H, generator=gen)

    async def run():
        with pytest.raises(RuntimeError):
            async for _ in iface.generate_async(prompts[0], trailing, pad, None):
                pass
        await iface.start_zmq_tasks()
        await iface.start_zmq_tasks()                      # idempotent

        async def one(pr):
            return [f async for f in iface.generate_async(pr, trailing, pad, None)]
        res = await asyncio.gather(*[one(pr) for pr in prompts])      # three concurrent requests
        # a consumer that stops early must release its request on both engines
        agen = iface.generate_async(prompts[0], trailing, pad, None, request_id="early")
        first = await agen.__anext__()
        await agen.aclose()
        holder = iface._mp_holder
        await iface.stop_zmq_tasks()
        return res, first, holder

    res, first, holder = asyncio.run(run())
    for pr, got in zip(prompts, res):
        assert got == expected_frames(iface, pr, trailing, pad, 5)
        assert all(len(f) == 16 for f in got)
    assert first == res[0][0]
    assert "early" in holder.talker_client._engine.cleared and "early" in holder.predictor_client._engine.cleared
    assert holder.talker_client._engine.exited and holder.predictor_client._engine.exited
    assert not iface._request_queues


def test_interface_rejects_missing_path_and_front_end(model_dir):
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    with pytest.raises(ValueError):
        Qwen3TTSInterface.from_pretrained("/nonexistent/model/dir")
    iface = Qwen3TTSInterface(model_dir)
    with pytest.raises(RuntimeError, match="no tokenizer"):
        iface.generate_voice_design("hi", "a calm voice")          # the synthetic directory has no tokenizer files
    with pytest.raises(RuntimeError, match="speech_tokenizer not available"):
        iface.create_voice_clone_prompt(ref_audio="ref.wav", ref_text="hello")      # no speech codec package here
    with pytest.raises(AssertionError):
        Qwen3TTSInterface(model_dir, tensor_parallel_size=2)


def test_orchestrator_loops_put_the_reference_messages_on_the_request_queues():
    """tests/golden/loop_messages.json holds what the reference's run_talker_loop_mp / run_predictor_loop_mp
    (zmq/engine_loop_mp.py:27-158) put on the per-request queues in the scripted scenario of tests/loop_scenario.py
    (written by oracle/ref_callers.py): ("talker", "token", {token_ids, hidden_states}), ("talker", "done", {}),
    ("predictor", "token", {token_ids}); outputs of a closed request dropped; a failing step retried; the worker not
    asked while nobody is ready.  This package's loops must produce the same, message for message."""
    import json
    import logging
    import loop_scenario
    from nano_qwen3tts_vllm.zmq import engine_loop_mp
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "loop_messages.json")))
    logging.disable(logging.CRITICAL)
    try:
        got = loop_scenario.drive(engine_loop_mp)
    finally:
        logging.disable(logging.NOTSET)
    assert got == want
    assert want["talker"]["queues"]["b"][-1] == ["talker", "done", {}]
    assert want["talker"]["idle_calls"] == 0 and want["talker"]["client_calls"] == 5


def test_fused_loop_admits_requests_while_a_frame_is_in_flight(model_dir, monkeypatch):
    """The fused loop hands the engine a hook (step(poll=...)): requests that arrive while the engine's thread is busy
    with a frame are admitted from inside that call (add_streams(wait=False)) instead of after it.  With a stand-in
    engine on the CPU: every request gets all its frames in order, nothing is lost or duplicated when a stream is
    admitted mid-frame, a consumer that leaves early releases its row, and some admissions did come through the hook."""
    import fake_frame_engine
    from nano_qwen3tts_vllm.engine import frame_engine as FE
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    monkeypatch.setenv("QTTS_ASYNC_ENGINE", "fused")
    monkeypatch.delenv("QTTS_ADMISSION", raising=False)
    monkeypatch.setattr(FE, "FrameEngine", fake_frame_engine.FakePollingFrameEngine)
    iface = Qwen3TTSInterface.from_pretrained(model_dir)
    iface.max_streams = 3
    H = iface.model_config.talker_config.hidden_size
    gen = torch.Generator().manual_seed(5)
    prompts = [torch.randn(1, 4 + i, H, generator=gen) for i in range(7)]
    trailing = torch.randn(1, 3, H, generator=gen)
    pad = torch.randn(1, 1, H, generator=gen)

    def want(pr):
        key = int(pr.reshape(-1, H).float().abs().sum().item() * 7) % 1000
        return [[(key + n * 17 + c) % 256 for c in range(16)] for n in range(fake_frame_engine.FRAMES)]

    async def run():
        await iface.start_zmq_tasks()
        fe = iface._frame_engine

        async def one(i):
            await asyncio.sleep(0.003 * i)              # arrivals spread over the frames of the earlier requests
            return [f async for f in iface.generate_async(prompts[i], trailing, pad, None, request_id=f"r{i}")]
        res = await asyncio.gather(*[one(i) for i in range(6)])
        agen = iface.generate_async(prompts[6], trailing, pad, None, request_id="early")
        first = await agen.__anext__()
        await agen.aclose()
        await asyncio.sleep(0.05)
        left = dict(fe.streams)
        n = fe.admitted_in_flight
        await iface.stop_zmq_tasks()
        return res, first, left, n

    res, first, left, n = asyncio.run(run())
    for i, got in enumerate(res):
        assert got == want(prompts[i]), i
    assert first == want(prompts[6])[0] and not left
    assert n > 0, "no request was admitted from the engine's hook"
    assert not iface._request_queues


@pytest.mark.parametrize("admission", ["join", "beside"])
def test_fused_loop_keeps_streams_whose_prefill_has_not_joined_yet(model_dir, monkeypatch, admission):
    """A stream the engine has admitted without waiting is in `engine.streams` before it takes part in a frame (its
    prefill runs beside the frames, or waits behind another group's).  A frame that brings nothing for it is not its
    end: with a stand-in engine whose admissions join two frames late, every request still gets all its frames."""
    import fake_frame_engine
    from nano_qwen3tts_vllm.engine import frame_engine as FE
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    monkeypatch.setenv("QTTS_ASYNC_ENGINE", "fused")
    monkeypatch.setenv("QTTS_ADMISSION", admission)
    monkeypatch.setattr(FE, "FrameEngine", fake_frame_engine.FakeLateJoinFrameEngine)
    iface = Qwen3TTSInterface.from_pretrained(model_dir)
    iface.max_streams = 4
    H = iface.model_config.talker_config.hidden_size
    gen = torch.Generator().manual_seed(9)
    prompts = [torch.randn(1, 3 + i, H, generator=gen) for i in range(6)]
    trailing = torch.randn(1, 3, H, generator=gen)
    pad = torch.randn(1, 1, H, generator=gen)

    def want(pr):
        key = int(pr.reshape(-1, H).float().abs().sum().item() * 7) % 1000
        return [[(key + n * 17 + c) % 256 for c in range(16)] for n in range(fake_frame_engine.FRAMES)]

    async def run():
        await iface.start_zmq_tasks()

        async def one(i):
            await asyncio.sleep(0.004 * i)
            return [f async for f in iface.generate_async(prompts[i], trailing, pad, None, request_id=f"r{i}")]
        res = await asyncio.gather(*[one(i) for i in range(6)])
        fe = iface._frame_engine
        late, left = fe.late_frames, dict(fe.streams)
        await iface.stop_zmq_tasks()
        return res, late, left

    res, late, left = asyncio.run(run())
    assert late > 0, "no admission was held back"
    for i, got in enumerate(res):
        assert got == want(prompts[i]), (i, len(got))
    assert not left and not iface._request_queues
