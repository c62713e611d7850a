"""What the boundaries inside a frame cost: the frame as the engine runs it (codes graph, D2H copy of the codes, talker
graph) against the same launches as ONE graph, as two graphs without the copy, and with the copy on a side stream.

    python tools/graph_boundary_probe.py      # on a GPU box; prints us per frame for each form
"""
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nano_qwen3tts_vllm import hip  # noqa: E402
from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine  # noqa: E402
from nano_qwen3tts_vllm.sampling_params import SamplingParams  # noqa: E402
from nano_qwen3tts_vllm.utils import synthetic  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = synthetic.make_config("1.7B")
    tw, pw = bench.make_weights(cfg, dev)
    d = tempfile.mkdtemp()
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    B, N = int(os.environ.get("QTTS_STREAMS", "8")), 40
    fe = FrameEngine(d, max_streams=B, max_text_frames=700, talker_weights=tw, predictor_weights=pw, num_kvcache_blocks=4 * B,
                     max_num_batched_tokens=4096, codec_eos=None)
    sp = (SamplingParams(temperature=1.0, max_tokens=1, do_sample=False), SamplingParams(temperature=0.9, max_tokens=17, do_sample=False))
    g = torch.Generator().manual_seed(1)
    for i in range(B):
        L = (16, 48, 128, 200)[i % 4]
        fe.add_stream((torch.randn(L, 2048, generator=g) * 0.5).to(dev), (torch.randn(600, 2048, generator=g) * 0.1).to(dev), 600, *sp)
    fe.run_frames(5)
    bs, ctx = fe._bucket(), 896                     # one context bucket for every form (covers all frames generated here)
    host = torch.empty(N, bs, 16, dtype=torch.int64).pin_memory()
    side = torch.cuda.Stream()
    ev = torch.cuda.Event()
    with fe.tr.on_stream():
        fe._setup_predictor_meta(bs)
        gc = hip.Graph.capture(lambda: fe._codes_part(bs))
        gt = hip.Graph.capture(lambda: fe._talker_part(bs, ctx))
        gw = hip.Graph.capture(lambda: (fe._codes_part(bs), fe._talker_part(bs, ctx)))

        def as_engine():
            for f in range(N):
                gc.launch()
                host[f].copy_(fe.codes[:bs], non_blocking=True)
                gt.launch()

        def two_graphs_no_copy():
            for f in range(N):
                gc.launch()
                gt.launch()

        def one_graph():
            for f in range(N):
                gw.launch()

        def copy_on_side_stream():
            for f in range(N):
                gc.launch()
                ev.record(fe.tr.stream)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    host[f].copy_(fe.codes[:bs], non_blocking=True)
                gt.launch()
        # (a side copy of frame f may read the codes while frame f + 1 rewrites them: fine for a timing probe only)
        for rep in range(2):
            for name, fn in (("as the engine runs it", as_engine), ("two graphs, no copy", two_graphs_no_copy), ("one graph", one_graph),
                             ("copy on a side stream", copy_on_side_stream)):
                fe.tr.stream.synchronize()
                t0 = time.perf_counter()
                fn()
                fe.tr.stream.synchronize()
                side.synchronize()
                print(f"{name:24s} {(time.perf_counter() - t0) / N * 1e6:8.1f} us per frame", flush=True)
    for s in fe.streams.values():
        s.frames.extend([[0] * 16] * (8 * N))
    fe.exit()


if __name__ == "__main__":
    main()
