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
from helpers import golden_weights, load_npz, oracle_cfgs

pytestmark = pytest.mark.gpu
MARGIN_TOL = 0.0625


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    from safetensors.torch import save_file
    d = str(tmp_path_factory.mktemp("iface_model"))
    cfg, sd, tw, pw = golden_weights("tiny", 11)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    save_file(sd, os.path.join(d, "model.safetensors"))
    g = load_npz("model_tiny.npz")
    tc, pc = oracle_cfgs(cfg)
    prompts = [torch.from_numpy(g[f"f_prompt{i}"]) for i in range(2)]
    text = torch.from_numpy(g["f_text_adds"])
    want, _, margins = O.frame_loop(tw, tc, pw, pc, prompts, text, 6)
    assert np.array_equal(want.numpy(), g["f_frames"])
    return d, prompts, text, want.numpy(), margins.numpy()


def check(got, want, margins):
    for f in range(want.shape[0]):
        if list(got[f]) != list(want[f]):
            assert margins[:f + 1].min() <= MARGIN_TOL, (f, got[f], want[f].tolist())
            return


def test_generate_async_matches_fixture(model, monkeypatch):
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    monkeypatch.setenv("QTTS_INPROCESS", "1")
    d, prompts, text, want, margins = model
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
        check(res[i], want[i], margins[i])


def test_sync_generate_matches_fixture(model):
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    d, prompts, text, want, margins = model
    iface = Qwen3TTSInterface(d, gpu_memory_utilization=0.05)
    iface.do_sample = False
    H = prompts[0].shape[-1]
    for i in range(2):
        got = list(iface.generate(prompts[i].unsqueeze(0), text[i].unsqueeze(0), torch.zeros(1, 1, H), None, max_frames=6))
        assert len(got) == 6
        check(got, want[i], margins[i])
    iface.shutdown()
