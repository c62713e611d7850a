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
import importlib.util
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


def codec_stand_in() -> types.ModuleType:
    """`qwen_tts` as far as the scripts and the wrapper touch it: a tokenizer whose decode returns 1920 samples of
    silence per frame at 24 kHz (12.5 Hz frames)."""
    import numpy as np

    class Qwen3TTSTokenizer:
        sample_rate = 24000

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
            spec = importlib.util.spec_from_file_location("ref_quick_benchmark", os.path.join(REF_EXAMPLES, "quick_benchmark.py"))
            qb = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(qb)
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


if __name__ == "__main__":
    import tempfile
    sys.path.insert(0, ROOT)
    d = make_model_dir(tempfile.mkdtemp(prefix="qtts_callers_"))
    rec = run_reference_callers(d, tempfile.mkdtemp())
    with open(GOLDEN, "w") as f:
        json.dump(rec, f, indent=1)
    print(GOLDEN, {k: len(v) for k, v in rec.items()})
