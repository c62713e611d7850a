"""The reference's own example scripts run unmodified on this repository's package, and the calls they make are the
committed recording (tests/golden/api_calls.json, written by oracle/ref_callers.py).

CPU (build container only — the reference tree does not travel): `examples/quick_benchmark.py::benchmark_nano_vllm`,
`examples/batch_async_example.py` as `__main__` and `examples/server.py` served in-process (start-up with its 17
warm-up requests, the GET routes, a speech request), with a stand-in engine and a stand-in codec (no GPU here, the
product has no CPU path).  GPU: the recorded call sequences replayed on the real HIP engines."""
import asyncio
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "api_calls.json")


@pytest.mark.skipif(not os.path.isdir("/root/reference/examples"), reason="the reference tree exists in the build container only")
def test_reference_example_scripts_run_unmodified(tmp_path):
    import ref_callers
    d = ref_callers.make_model_dir(str(tmp_path / "model"))
    rec = ref_callers.run_reference_callers(d, str(tmp_path / "out"))
    want = json.load(open(GOLDEN))
    assert rec == {k: want[k] for k in rec}
    # what the scripts rely on: from_pretrained / the constructor with exactly these keywords, sync and async
    # generators of 16-id frames, the codec wrapper behind `speech_tokenizer`
    names = [c["call"] for c in rec["quick_benchmark"]]
    assert names[:3] == ["Qwen3TTSInterface.from_pretrained", "Qwen3TTSInterface", "generate_voice_design"]
    assert names.count("speech_tokenizer.decode") == 2
    names = [c["call"] for c in rec["batch_async_example"]]
    assert names == ["Qwen3TTSInterface", "start_zmq_tasks"] + ["generate_custom_voice_async"] * 4 + ["stop_zmq_tasks"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/examples"), reason="the reference tree exists in the build container only")
def test_reference_server_runs_unmodified(tmp_path):
    """examples/server.py: module import, lifespan start-up (interface + codec wrapper construction, start_zmq_tasks,
    1 + 16 warm-up streams), GET /health /voices /, POST /v1/audio/speech streaming PCM16 (50 ms of leading silence +
    12 frames x 1920 samples from the stand-ins), request validation, shutdown (stop_zmq_tasks)."""
    import ref_callers
    d = ref_callers.make_model_dir(str(tmp_path / "model"))
    rec = ref_callers.run_reference_server(d, str(tmp_path / "scratch"))
    want = json.load(open(GOLDEN))["server"]
    assert rec == want
    names = [c["call"] for c in rec["interface"]]
    assert names == ["Qwen3TTSInterface", "start_zmq_tasks"] + ["generate_custom_voice_async"] * 18 + ["stop_zmq_tasks"]
    assert rec["codec"]["constructor"]["kwargs"] == {"device": "cuda:0", "num_graph_lengths": 50}
    assert rec["codec"]["methods"] == ["decode_window_batched"]
    speech = [h for h in rec["http"] if h["request"] == "POST /v1/audio/speech"][0]
    assert speech == {"request": "POST /v1/audio/speech", "status": 200, "content_type": "audio/L16", "sample_rate": "24000",
                      "pcm16_bytes": 2 * (24000 * 50 // 1000 + 12 * 1920)}
    assert not os.path.exists("/root/reference/examples/debug")


@pytest.mark.gpu
def test_recorded_server_calls_on_the_real_engine(tmp_path):
    """What the reference's server does with the two objects, replayed from the recording on the real engine: the
    constructor keywords, start_zmq_tasks, then the 16 concurrent warm-up streams (twice the engine's 8 stream rows:
    half of them wait for a row) each decoded the way the server streams them — every 4 frames the window of up to 8
    frames of context + the new frames through `decode_window_batched` of the codec wrapper (the codec network is a
    stand-in; graphs captured for windows up to 12 frames) — then stop_zmq_tasks."""
    from test_codec_wrapper import UP, VOCAB, StubTokenizer
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    from nano_qwen3tts_vllm.utils import synthetic
    from nano_qwen3tts_vllm.utils.speech_tokenizer_cudagraph import SpeechTokenizerCUDAGraph
    d = str(tmp_path / "model")
    synthetic.write_model_dir(d, "tiny", seed=21, with_text=True, embed_std=0.05, std=0.2)
    synthetic.write_tokenizer(d)
    rec = json.load(open(GOLDEN))["server"]
    ctor = {k: (d if v == "<model dir>" else v) for k, v in rec["interface"][0]["kwargs"].items()}
    iface = Qwen3TTSInterface(**ctor)
    iface.max_frames = 10
    ck = rec["codec"]["constructor"]["kwargs"]
    tok = StubTokenizer()
    tok.model.to(ck["device"])
    codec = SpeechTokenizerCUDAGraph(*rec["codec"]["constructor"]["args"], device=ck["device"], num_graph_lengths=12,
                                     dtype=torch.float32, tokenizer=tok)
    gens = [c for c in rec["interface"] if c["call"] == "generate_custom_voice_async"][1:17]
    assert len(gens) == 16

    async def one(c):
        frames, pos, samples = [], 0, 0
        async for f in iface.generate_custom_voice_async(**c["kwargs"]):
            frames.append(f)
            if len(frames) % 4 == 0:
                start = max(0, pos - 8)
                window = [[i % VOCAB for i in fr] for fr in frames[start:]]      # the stand-in decoder's tables have VOCAB rows
                wavs, sr = codec.decode_window_batched([{"audio_codes": window, "left_context_frames": pos - start}])
                assert sr == 24000 and len(wavs[0]) == (len(frames) - pos) * UP
                samples += len(wavs[0])
                pos = len(frames)
        return len(frames), samples

    async def run():
        await iface.start_zmq_tasks()
        res = await asyncio.gather(*(one(c) for c in gens))
        await iface.stop_zmq_tasks()
        return res
    res = asyncio.run(run())
    assert res == [(10, 8 * UP)] * 16
    iface.shutdown()


@pytest.mark.gpu
def test_recorded_call_sequences_on_the_real_engine(tmp_path):
    """Every call of the recording, in order and with its recorded keyword arguments, on the HIP engines (tiny
    synthetic checkpoint with text tables and an offline tokenizer; the codec package is absent, so
    `speech_tokenizer` is None as in the reference without qwen_tts and the decode calls are skipped)."""
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    from nano_qwen3tts_vllm.utils import synthetic
    d = str(tmp_path / "model")
    synthetic.write_model_dir(d, "tiny", seed=21, with_text=True, embed_std=0.05, std=0.2)
    synthetic.write_tokenizer(d)
    rec = json.load(open(GOLDEN))

    def kwargs_of(c):
        return {k: (d if v == "<model dir>" else v) for k, v in c["kwargs"].items()}

    # quick_benchmark: from_pretrained -> generate_voice_design x3 (sync generators of 16-id frames)
    calls = rec["quick_benchmark"]
    assert calls[0]["call"] == "Qwen3TTSInterface.from_pretrained"
    iface = Qwen3TTSInterface.from_pretrained(**kwargs_of(calls[0]))
    iface.max_frames, iface.do_sample = 10, True          # random weights never emit the codec EOS: bound the run
    assert iface.speech_tokenizer is None
    n = 0
    for c in calls[2:]:
        if c["call"] != "generate_voice_design":
            continue
        frames = list(iface.generate_voice_design(**kwargs_of(c)))
        assert len(frames) == 10 and all(isinstance(f, list) and len(f) == 16 for f in frames)
        n += 1
    assert n == 3
    iface.shutdown()
    del iface

    # batch_async_example: constructor -> start_zmq_tasks -> 4 concurrent generate_custom_voice_async -> stop_zmq_tasks
    calls = rec["batch_async_example"]
    iface = Qwen3TTSInterface(**kwargs_of(calls[0]))
    iface.max_frames = 10
    gens = [c for c in calls if c["call"] == "generate_custom_voice_async"]

    async def run():
        await iface.start_zmq_tasks()

        async def one(c):
            return [f async for f in iface.generate_custom_voice_async(**kwargs_of(c))]
        res = await asyncio.gather(*(one(c) for c in gens))
        await iface.stop_zmq_tasks()
        return res
    res = asyncio.run(run())
    assert len(res) == 4 and all(len(r) == 10 and all(len(f) == 16 for f in r) for r in res)
    iface.shutdown()
