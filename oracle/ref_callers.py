"""TEST INFRASTRUCTURE — runs the REFERENCE's own example scripts, unmodified, against this repository's
`nano_qwen3tts_vllm` package and records every call they make on the API object.

    python oracle/ref_callers.py            # (build container only) rewrites tests/golden/api_calls.json

What runs: `/root/reference/examples/quick_benchmark.py::benchmark_nano_vllm` (from_pretrained -> warm-up ->
N x list(generate_voice_design(...)) -> speech_tokenizer.decode, :25-84) and
`/root/reference/examples/batch_async_example.py` as `__main__` (Qwen3TTSInterface(...) -> start_zmq_tasks ->
asyncio.gather of generate_custom_voice_async streams -> stop_zmq_tasks, :47-61,99-147).  There is no GPU in the build
container and the product has no CPU path, so the engine under the API object is a deterministic stand-in
(tests/fake_frame_engine.py) and the third-party codec package `qwen_tts` is a stand-in module that returns silence
of the right length: what this pins is the API surface — names, keyword arguments, call order, the shapes that
come back — which the GPU tests then replay on the real engine from the committed recording.
"""
from __future__ import annotations

import asyncio
import contextlib
import io
import json
import os
import runpy
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_EXAMPLES = "/root/reference/examples"
GOLDEN = os.path.join(ROOT, "tests", "golden", "api_calls.json")


def make_model_dir(path: str) -> str:
    """A tiny synthetic checkpoint with the text tables and an offline tokenizer: a directory the scripts can be
    pointed at with --model-path."""
    from nano_qwen3tts_vllm.utils import synthetic
    synthetic.write_model_dir(path, "tiny", seed=21, with_text=True, embed_std=0.05, std=0.2)
    synthetic.write_tokenizer(path, synthetic.make_config("tiny")["talker_config"]["text_vocab_size"])
    return path


def load_reference_script(name: str, as_dir: str, module_name: str):
    """The reference's examples/<name>, compiled from its text as it stands and executed as a module whose
    `__file__` lies in `as_dir`: a script that creates directories next to itself (server.py: voices/, debug/) then
    does so in a scratch directory, and no bytecode cache is written beside the source — nothing under
    /root/reference is touched."""
    path = os.path.join(REF_EXAMPLES, name)
    with open(path) as f:
        code = compile(f.read(), os.path.join(as_dir, name), "exec", dont_inherit=True)
    mod = types.ModuleType(module_name)
    mod.__file__ = os.path.join(as_dir, name)
    sys.modules[module_name] = mod
    exec(code, mod.__dict__)
    return mod


def codec_stand_in() -> types.ModuleType:
    """`qwen_tts` as far as the scripts and the wrappers touch it: a tokenizer whose decode returns 1920 samples of
    silence per frame at 24 kHz (12.5 Hz frames), with a `.model.decoder` module of the same behaviour."""
    import numpy as np
    import torch

    class Decoder(torch.nn.Module):
        total_upsample = 1920

        def forward(self, codes):                                  # [B, 16, T] ids -> [B, 1, T * 1920]
            return torch.zeros(codes.shape[0], 1, codes.shape[-1] * self.total_upsample)

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.decoder = Decoder()

        def get_output_sample_rate(self):
            return 24000

    class Qwen3TTSTokenizer:
        sample_rate = 24000

        def __init__(self):
            self.model = Model()

        @classmethod
        def from_pretrained(cls, name, **kw):
            return cls()

        def decode(self, inputs):
            return [np.zeros(1920 * int(len(i["audio_codes"])), dtype=np.float32) for i in inputs], self.sample_rate

    mod = types.ModuleType("qwen_tts")
    mod.Qwen3TTSTokenizer = Qwen3TTSTokenizer
    inf = types.ModuleType("qwen_tts.inference")
    tk = types.ModuleType("qwen_tts.inference.qwen3_tts_tokenizer")
    tk.Qwen3TTSTokenizer = Qwen3TTSTokenizer
    mod.inference, inf.qwen3_tts_tokenizer = inf, tk
    return {"qwen_tts": mod, "qwen_tts.inference": inf, "qwen_tts.inference.qwen3_tts_tokenizer": tk}


@contextlib.contextmanager
def recording(calls: list):
    """Wraps the API object's public entry points so that every call is appended to `calls` as
    {"call": name, "kwargs": {...}, "returns": shape summary}."""
    from nano_qwen3tts_vllm import interface as I
    cls = I.Qwen3TTSInterface
    saved = {}

    def note(name, kwargs, returns=None):
        calls.append({"call": name, "kwargs": {k: (v if isinstance(v, (str, int, float, bool, type(None))) else type(v).__name__)
                                                for k, v in kwargs.items()}, "returns": returns})
        return calls[-1]

    def wrap_init(orig):
        def __init__(self, *a, **k):
            note("Qwen3TTSInterface", dict(zip(("model_path", "enforce_eager", "tensor_parallel_size", "gpu_memory_utilization"), a), **k)
                 | {"model_path": "<model dir>"})
            orig(self, *a, **k)
            if self.speech_tokenizer is not None:
                dec = self.speech_tokenizer.decode

                def decode(arg):
                    wavs, sr = dec(arg)
                    note("speech_tokenizer.decode", {"arg": "[{'audio_codes': frames}]" if isinstance(arg, list) and arg and isinstance(arg[0], dict) else type(arg).__name__},
                         {"wavs": len(wavs), "samples_per_frame": len(wavs[0]) // max(len(arg[0]["audio_codes"]), 1), "sample_rate": sr})
                    return wavs, sr
                self.speech_tokenizer.decode = decode
        return __init__

    def wrap_sync_gen(name, orig):
        def f(self, *a, **k):
            rec = note(name, dict(k, **{f"arg{i}": v for i, v in enumerate(a)}))
            n = 0
            for frame in orig(self, *a, **k):
                n += 1
                assert isinstance(frame, list) and len(frame) == 16
                yield frame
            rec["returns"] = {"frames_of_16_ids": n > 0}
        return f

    def wrap_async_gen(name, orig):
        async def f(self, *a, **k):
            rec = note(name, dict(k, **{f"arg{i}": v for i, v in enumerate(a)}))
            n = 0
            async for frame in orig(self, *a, **k):
                n += 1
                assert isinstance(frame, list) and len(frame) == 16
                yield frame
            rec["returns"] = {"frames_of_16_ids": n > 0}
        return f

    def wrap_coro(name, orig):
        async def f(self, *a, **k):
            note(name, k)
            return await orig(self, *a, **k)
        return f

    try:
        saved["__init__"] = cls.__init__
        cls.__init__ = wrap_init(cls.__init__)
        orig_fp = cls.from_pretrained.__func__
        saved["from_pretrained"] = cls.__dict__["from_pretrained"]

        def from_pretrained(c, *a, **k):
            note("Qwen3TTSInterface.from_pretrained", dict(k, **{f"arg{i}": "<model dir>" for i, _ in enumerate(a)})
                 | ({"pretrained_model_name_or_path": "<model dir>"} if "pretrained_model_name_or_path" in k else {}))
            return orig_fp(c, *a, **k)
        cls.from_pretrained = classmethod(from_pretrained)
        for name in ("generate_voice_design", "generate_custom_voice", "generate_voice_clone"):
            saved[name] = getattr(cls, name)
            setattr(cls, name, wrap_sync_gen(name, saved[name]))
        for name in ("generate_custom_voice_async", "generate_voice_clone_async"):
            saved[name] = getattr(cls, name)
            setattr(cls, name, wrap_async_gen(name, saved[name]))
        for name in ("start_zmq_tasks", "stop_zmq_tasks"):
            saved[name] = getattr(cls, name)
            setattr(cls, name, wrap_coro(name, saved[name]))
        yield
    finally:
        for name, orig in saved.items():
            setattr(cls, name, orig)


def run_reference_callers(model_dir: str, out_dir: str) -> dict:
    """Runs both scripts; returns {"quick_benchmark": [calls...], "batch_async_example": [calls...]}."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_frame_engine
    from nano_qwen3tts_vllm.engine import frame_engine as FE
    saved_modules = {k: sys.modules.get(k) for k in codec_stand_in()}
    saved_engine, saved_argv = FE.FrameEngine, list(sys.argv)
    sys.modules.update(codec_stand_in())
    FE.FrameEngine = fake_frame_engine.FakeFrameEngine
    result = {}
    try:
        calls: list = []
        with recording(calls), contextlib.redirect_stdout(io.StringIO()) as out:
            qb = load_reference_script("quick_benchmark.py", out_dir, "ref_quick_benchmark")
            avg_time, avg_rtf = qb.benchmark_nano_vllm(model_dir, num_runs=2)
        assert avg_time > 0 and avg_rtf > 0 and "Avg RTF" in out.getvalue()
        result["quick_benchmark"] = calls
        calls = []
        sys.argv = ["batch_async_example.py", "--model-path", model_dir, "--batch-size", "4", "--output-dir", out_dir]
        with recording(calls), contextlib.redirect_stdout(io.StringIO()) as out:
            runpy.run_path(os.path.join(REF_EXAMPLES, "batch_async_example.py"), run_name="__main__")
        text = out.getvalue()
        assert "All streams finished" in text and "Error:" not in text and "Batch async example finished." in text, text
        result["batch_async_example"] = calls
    finally:
        sys.argv = saved_argv
        FE.FrameEngine = saved_engine
        for k, v in saved_modules.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return result


def run_reference_server(model_dir: str, scratch: str) -> dict:
    """`/root/reference/examples/server.py` as it stands, served in-process (starlette's TestClient drives the ASGI
    app: lifespan start-up with its 1 + 16 warm-up requests, GET /health, /voices, /, POST /v1/audio/speech, lifespan
    shutdown).  Returns {"interface": [calls...], "codec": {constructor keywords, methods used}, "http": [...]}.
    Stand-ins as in run_reference_callers; the codec wrapper is the product's class with the device the server
    hard-wires ("cuda:0") replaced by "cpu", since there is no GPU where this runs."""
    from starlette.testclient import TestClient
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_frame_engine
    from nano_qwen3tts_vllm.engine import frame_engine as FE
    os.makedirs(scratch, exist_ok=True)
    stand_ins = codec_stand_in()
    saved_modules = {k: sys.modules.get(k) for k in stand_ins}
    sys.modules.update(stand_ins)
    import importlib
    import nano_qwen3tts_vllm.utils.speech_tokenizer_cudagraph as W
    W = importlib.reload(W)                                 # picks the stand-in codec package up
    saved_engine, real_wrapper = FE.FrameEngine, W.SpeechTokenizerCUDAGraph
    saved_env = {k: os.environ.get(k) for k in ("QWEN3_TTS_MODEL_PATH", "DEBUG_SAVE_AUDIO")}
    codec = {"constructor": None, "methods": set()}

    class RecordedWrapper(real_wrapper):
        def __init__(self, *a, **k):
            codec["constructor"] = {"args": list(a), "kwargs": dict(k)}
            k["device"] = "cpu"
            super().__init__(*a, **k)

        def __getattribute__(self, name):
            if name in ("decode", "chunked_decode", "decode_window_batched", "decode_codec_ids"):
                codec["methods"].add(name)
            return super().__getattribute__(name)

    W.SpeechTokenizerCUDAGraph = RecordedWrapper
    FE.FrameEngine = fake_frame_engine.FakeFrameEngine
    os.environ["QWEN3_TTS_MODEL_PATH"] = model_dir
    os.environ["DEBUG_SAVE_AUDIO"] = "0"
    calls: list = []
    http = []
    try:
        with recording(calls):
            srv = load_reference_script("server.py", scratch, "ref_server")
            with TestClient(srv.app) as client:
                for path in ("/health", "/voices", "/"):
                    r = client.get(path)
                    http.append({"request": "GET " + path, "status": r.status_code, "json": r.json()})
                r = client.post("/v1/audio/speech", json={"text": "Hello there.", "language": "English", "speaker": "Vivian"})
                http.append({"request": "POST /v1/audio/speech", "status": r.status_code,
                             "content_type": r.headers["content-type"].split(";")[0], "sample_rate": r.headers.get("sample-rate"),
                             "pcm16_bytes": len(r.content)})
                r = client.post("/v1/audio/speech", json={"text": ""})
                http.append({"request": "POST /v1/audio/speech (empty text)", "status": r.status_code})
    finally:
        FE.FrameEngine = saved_engine
        W.SpeechTokenizerCUDAGraph = real_wrapper
        sys.modules.pop("ref_server", None)
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        for k, v in saved_modules.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        importlib.reload(W)
    codec["methods"] = sorted(codec["methods"])
    return {"interface": calls, "codec": codec, "http": http}


def run_reference_loops() -> dict:
    """The reference's orchestrator loops (/root/reference/nano-qwen3tts-vllm/zmq/engine_loop_mp.py:27-158, stdlib
    only) driven by the scripted scenario of tests/loop_scenario.py: the messages they put on the per-request queues."""
    import logging
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import loop_scenario
    path = "/root/reference/nano-qwen3tts-vllm/zmq/engine_loop_mp.py"
    with open(path) as f:
        code = compile(f.read(), "<reference zmq/engine_loop_mp.py>", "exec", dont_inherit=True)
    mod = types.ModuleType("ref_engine_loop_mp")
    exec(code, mod.__dict__)
    logging.disable(logging.CRITICAL)
    try:
        return loop_scenario.drive(mod)
    finally:
        logging.disable(logging.NOTSET)


if __name__ == "__main__":
    import tempfile
    sys.path.insert(0, ROOT)
    d = make_model_dir(tempfile.mkdtemp(prefix="qtts_callers_"))
    rec = run_reference_callers(d, tempfile.mkdtemp())
    rec["server"] = run_reference_server(d, tempfile.mkdtemp())
    with open(GOLDEN, "w") as f:
        json.dump(rec, f, indent=1)
    print(GOLDEN, {k: len(v) for k, v in rec.items()})
    with open(os.path.join(ROOT, "tests", "golden", "loop_messages.json"), "w") as f:
        json.dump(run_reference_loops(), f, indent=1)
