"""Host time of an admission that does not wait for the GPU (FrameEngine.add_streams(wait=False)): what the serving
loop spends before the prompt's first kernel is queued, per prompt length, split by cProfile.

    python tools/prof_add_stream.py
"""
import cProfile
import json
import os
import pstats
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine  # noqa: E402
from nano_qwen3tts_vllm.sampling_params import SamplingParams  # noqa: E402
from nano_qwen3tts_vllm.utils import synthetic  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = synthetic.make_config("1.7B")
    tw, pw = bench.make_weights(cfg, dev)
    d = tempfile.mkdtemp()
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    fe = FrameEngine(d, max_streams=8, max_text_frames=160, talker_weights=tw, predictor_weights=pw,
                     num_kvcache_blocks=32, max_num_batched_tokens=4096, codec_eos=None)
    sp = (SamplingParams(temperature=1.0, max_tokens=1, do_sample=False), SamplingParams(temperature=0.9, max_tokens=17, do_sample=False))
    prof = cProfile.Profile()
    for L in (16, 16, 48, 200, 200, 48, 16):
        x = torch.randn(L, 2048, device=dev) * 0.5
        t = torch.randn(126, 2048, device=dev) * 0.1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first = not getattr(main, 'warm', False)
        main.warm = True
        if not first:
            prof.enable()
        (sid,) = fe.add_streams([dict(prompt_embeds=x, text_adds=t, max_frames=125, talker_sampling=sp[0], predictor_sampling=sp[1])], wait=False)
        prof.disable()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"prompt {L:4d}: host {1e3 * (t1 - t0):.3f} ms until the call returns, GPU done {1e3 * (t2 - t0):.3f} ms after the call began", flush=True)
        fe.step()
        fe.remove_stream(sid)
        torch.cuda.synchronize()
    pstats.Stats(prof).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
