"""GPU: the API object end to end — Qwen3TTSInterface.generate_async through the orchestrator
loops, in-process worker clients and the real HIP engines, and the synchronous generate()
through the FrameEngine — against the reference fixture's frames (greedy)."""
import asyncio
import json
import os

import numpy as np
import pytest
import torch

import qtts_oracle as O
from helpers import check_frames, fixture_model, golden_weights, oracle_cfgs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    """The `sharp` fixture family (clear winners: id parity binds) as a model directory."""
    from safetensors.torch import save_file
    d = str(tmp_path_factory.mktemp("iface_model"))
    cfg, sd, tw, pw, g = fixture_model("sharp")
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    save_file(sd, os.path.join(d, "model.safetensors"))
    prompts = [torch.from_numpy(g[f"f_prompt{i}"]) for i in range(2)]
    text = torch.from_numpy(g["f_text_adds"])

    def check(got_by_stream, streams, what):
        """got_by_stream: one list of 6 frames per entry of `streams` (indices into the fixture's two prompts): every
        pick judged against the oracle's logits of the same history, at least 90 % of them binding."""
        tc, pc = oracle_cfgs(cfg)
        got = np.array(got_by_stream, dtype=np.int64)
        check_frames(got, tw, tc, pw, pc, [prompts[i] for i in streams], text[list(streams)], what, 2.0, 0.9,
                     fixture_frames=g["f_frames"][list(streams)], fixture_tag="sharp")
    return d, prompts, text, check


@pytest.mark.parametrize("mode", ["fused", "workers"])
def test_generate_async_matches_fixture(model, monkeypatch, mode):
    """fused: one loop over the device-resident frame engine; workers: the reference's talker / predictor
    loops over the two engines (in-process clients)."""
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    monkeypatch.setenv("QTTS_INPROCESS", "1")
    monkeypatch.setenv("QTTS_ASYNC_ENGINE", mode)
    d, prompts, text, check = model
    iface = Qwen3TTSInterface.from_pretrained(d, gpu_memory_utilization=0.05)
    iface.do_sample = False
    iface.max_frames = 6
    H = prompts[0].shape[-1]
    pad = torch.zeros(1, 1, H)

    async def run():
        await iface.start_zmq_tasks()

        async def one(i):
            return [f async for f in iface.generate_async(prompts[i].unsqueeze(0), text[i].unsqueeze(0), pad, None)]
        res = await asyncio.gather(one(0), one(1))          # two concurrent streams, batched by the loops
        await iface.stop_zmq_tasks()
        return res

    res = asyncio.run(run())
    for i in range(2):
        assert len(res[i]) == 6 and all(len(f) == 16 for f in res[i])
    check(res, (0, 1), f"generate_async, {mode}")
    iface.shutdown()


def test_fused_async_continuous_batching(model, monkeypatch):
    """More requests than rows, a consumer that leaves early, staggered arrival: every stream still gets its
    own frames, and rows / KV blocks are all returned."""
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    monkeypatch.setenv("QTTS_ASYNC_ENGINE", "fused")
    monkeypatch.setenv("QTTS_MAX_STREAMS", "2")
    d, prompts, text, check = model
    iface = Qwen3TTSInterface.from_pretrained(d, gpu_memory_utilization=0.05)
    iface.do_sample = False
    iface.max_frames = 6
    pad = torch.zeros(1, 1, prompts[0].shape[-1])

    async def run():
        await iface.start_zmq_tasks()
        total_blocks = len(iface._frame_engine.talker.scheduler.block_manager.free_block_ids)

        async def one(i, delay):
            await asyncio.sleep(delay)
            return [f async for f in iface.generate_async(prompts[i].unsqueeze(0), text[i].unsqueeze(0), pad, None)]

        async def early():
            agen = iface.generate_async(prompts[0].unsqueeze(0), text[0].unsqueeze(0), pad, None, request_id="early")
            first = await agen.__anext__()
            await agen.aclose()
            return first
        res = await asyncio.gather(one(0, 0), one(1, 0), one(0, 0.002), one(1, 0.01), early())
        await asyncio.sleep(0.01)
        fe = iface._frame_engine
        state = (len(fe.streams), len(fe.talker.scheduler.block_manager.free_block_ids) - total_blocks, dict(iface._request_queues))
        await iface.stop_zmq_tasks()
        return res, state

    res, (n_streams, missing_blocks, queues) = asyncio.run(run())
    assert all(len(got) == 6 for got in res[:4])
    check(res[:4], (0, 1, 0, 1), "generate_async, more requests than rows")
    assert list(res[4]) == list(res[0][0])
    assert n_streams == 0 and not queues
    iface.shutdown()
    assert missing_blocks == 0                # every KV block went back to the free list


def test_sync_generate_matches_fixture(model):
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    d, prompts, text, check = model
    iface = Qwen3TTSInterface(d, gpu_memory_utilization=0.05)
    iface.do_sample = False
    H = prompts[0].shape[-1]
    res = []
    for i in range(2):
        res.append(list(iface.generate(prompts[i].unsqueeze(0), text[i].unsqueeze(0), torch.zeros(1, 1, H), None, max_frames=6)))
        assert len(res[-1]) == 6
    check(res, (0, 1), "sync generate")
    iface.shutdown()


def test_voice_design_text_entry_point(tmp_path):
    """Text in, frames out: generate_voice_design / generate_custom_voice_async build the prompt with the
    front-end and run it on the engine; same frames as handing the prepared embeddings to generate()."""
    from test_frontend import CharProcessor
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    from nano_qwen3tts_vllm.utils import synthetic
    d = str(tmp_path)
    synthetic.write_model_dir(d, "tiny", seed=21, with_text=True, embed_std=0.05, std=0.2)
    iface = Qwen3TTSInterface.from_pretrained(d, gpu_memory_utilization=0.05)
    iface.processor = CharProcessor(iface.model_config.talker_config.text_vocab_size)
    iface.do_sample = False
    iface.max_frames = 5
    got = list(iface.generate_voice_design("Hello there, world.", "A calm low voice", language="English"))
    emb, tail, pad, _ = iface._prep_custom_voice("Hello there, world.", "English", "", "A calm low voice", True)
    want = list(iface.generate(emb, tail, pad, None, max_frames=5))
    assert len(got) == 5 and all(len(f) == 16 for f in got) and got == want

    async def run():
        await iface.start_zmq_tasks()
        a = [f async for f in iface.generate_custom_voice_async("Good morning.", language="English", speaker="Vivian")]
        await iface.stop_zmq_tasks()
        return a
    a = asyncio.run(run())
    b = list(iface.generate_custom_voice("Good morning.", language="English", speaker="Vivian"))
    assert len(a) == 5 and a == b
    iface.shutdown()
