"""K concurrent streams through the async API: the counterpart of the reference's examples/batch_async_example.py
(E/batch_async_example.py:47-61,123-147) and of the streaming loop of its server (E/server.py:599-637):
start_zmq_tasks() -> asyncio.gather of K generate_async streams -> first-chunk and inter-chunk latency per stream,
aggregate codec tokens/s and RTF.  Requests are synthetic embeddings (see quick_benchmark.py).

    python examples/batch_async_example.py [--streams 8] [--frames 125] [--stagger-ms 50] [--engine fused|workers]
"""
import argparse
import asyncio
import json
import os
import statistics
import time

from _common import FRAME_SECONDS, model_dir, request


async def run(args):
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    d = model_dir(args.size, args.model_dir)
    iface = Qwen3TTSInterface.from_pretrained(d, enforce_eager=False, tensor_parallel_size=1)
    iface.do_sample = args.sample
    iface.ignore_eos = True
    iface.max_frames = args.frames
    H = iface.model_config.talker_config.hidden_size
    lens = [16, 48, 128, 200]
    reqs = [request(H, lens[i % 4], 20, seed=1000 + i) for i in range(args.streams)]
    await iface.start_zmq_tasks()

    async def stream(i, delay, stats):
        await asyncio.sleep(delay)
        t0 = time.perf_counter()
        stamps = []
        async for frame in iface.generate_async(*reqs[i], None):
            assert len(frame) == 16
            stamps.append(time.perf_counter())
        stats.append({"first": stamps[0] - t0, "gaps": [b - a for a, b in zip(stamps, stamps[1:])], "frames": len(stamps),
                      "total": stamps[-1] - t0})

    warm = []
    old, iface.max_frames = iface.max_frames, 4
    await asyncio.gather(*[stream(i, 0.0, warm) for i in range(args.streams)])          # warm-up: every graph bucket
    iface.max_frames = old
    stats = []
    t0 = time.perf_counter()
    await asyncio.gather(*[stream(i, i * args.stagger_ms / 1e3, stats) for i in range(args.streams)])
    wall = time.perf_counter() - t0
    await iface.stop_zmq_tasks()
    frames = sum(s["frames"] for s in stats)
    gaps = [g for s in stats for g in s["gaps"]]
    print(json.dumps({"example": "batch_async_example", "engine": iface._async_engine, "model": args.size,
                      "streams": args.streams, "frames_per_stream": args.frames, "stagger_ms": args.stagger_ms,
                      "first_chunk_ms_p50": statistics.median(s["first"] for s in stats) * 1e3,
                      "first_chunk_ms_max": max(s["first"] for s in stats) * 1e3,
                      "inter_chunk_ms_p50": statistics.median(gaps) * 1e3,
                      "inter_chunk_ms_p99": sorted(gaps)[int(0.99 * (len(gaps) - 1))] * 1e3,
                      "codec_tokens_per_s": 16 * frames / wall,
                      "rtf_per_stream_p50": statistics.median(s["total"] / (s["frames"] * FRAME_SECONDS) for s in stats),
                      "sampling": "reference" if args.sample else "greedy",
                      "data": "synthetic embeddings, random-init weights; a chunk is one 16-code frame"}))
    iface.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1.7B")
    ap.add_argument("--model-dir", default=None)
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--frames", type=int, default=125)
    ap.add_argument("--stagger-ms", type=float, default=50.0, help="one new stream every so many ms")
    ap.add_argument("--engine", default="fused", choices=["fused", "workers"])
    ap.add_argument("--sample", action="store_true")
    args = ap.parse_args()
    os.environ["QTTS_ASYNC_ENGINE"] = args.engine
    os.environ.setdefault("QTTS_INPROCESS", "1")
    os.environ["QTTS_MAX_STREAMS"] = str(max(args.streams, 1))
    asyncio.run(run(args))


if __name__ == "__main__":
    main()
