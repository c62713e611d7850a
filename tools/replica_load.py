"""Host-side load test of the replica router (no GPU): N stand-in replica processes, each producing a frame for every
one of its streams every `step` seconds (the real engine's 2.5 ms), and the parent's one result thread fanning the
frames out.  Reports delivered against offered frames per second and the CPU share of the router thread.

    python tools/replica_load.py [replicas=8] [streams_per_replica=8] [frames=300] [step_ms=2.5] [async|sinks]

async: every stream is consumed by a coroutine of one event loop (generate_async), as an API server does.
sinks: the frames are left in the streams' queues and counted at the end - the router thread by itself (hundreds of
streams at 400 frames a second each are more than one Python event loop can wake coroutines for; an API process per
few replicas, the reference's own `uvicorn --workers N` hint (examples/server.py:6), is the answer there, not the router).
"""
import asyncio
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(n_replicas=8, streams=8, frames=300, step_s=0.0025, consumers="async"):
    import torch
    from nano_qwen3tts_vllm.replicas import ReplicaPool
    os.environ["PYTHONPATH"] = os.path.join(ROOT, "tests") + os.pathsep + ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")
    pool = ReplicaPool(n_replicas, "fake_replica_engine:make", dict(max_streams=streams, step_seconds=step_s, paced=True),
                       devices=[None] * n_replicas, max_streams_per_replica=streams, start_timeout=300)
    total = n_replicas * streams
    text = torch.zeros(1, 8)
    prompt = torch.zeros(2, 8)
    stamps = {}
    try:
        if consumers == "async":
            async def main():
                async def one(i):
                    n, first = 0, None
                    async for _ in pool.generate_async(prompt, text, frames, request_id=f"s{i}"):
                        if first is None:
                            first = time.perf_counter()
                        n += 1
                    return n, first, time.perf_counter()
                return await asyncio.gather(*(one(i) for i in range(total)))
            res = asyncio.run(main())
            got = sum(r[0] for r in res)
            t_first, t_last = min(r[1] for r in res), max(r[2] for r in res)
        else:
            rids = [pool.submit(prompt, text, frames, request_id=f"s{i}") for i in range(total)]
            sinks = [pool._sinks[r] for r in rids]
            while sinks[0].empty():
                time.sleep(0.0005)
            t_first = time.perf_counter()
            while pool.load() != [0] * n_replicas:           # every stream has ended (its end marker went through the router)
                time.sleep(0.002)
            t_last = time.perf_counter()
            got = sum(sk.qsize() - 1 for sk in sinks)
        cpu = pool.router_cpu_seconds
        msgs = pool.frame_messages
    finally:
        pool.close()
    wall = t_last - t_first
    offered = total / step_s
    # the first frame's own step is not inside [first frame, last frame]
    delivered = (got - total) / wall if wall > 0 else 0.0
    return {"replicas": n_replicas, "streams_per_replica": streams, "frames_per_stream": frames, "consumers": consumers,
            "frames_delivered": got, "frames_expected": total * frames, "offered_frames_per_s": round(offered),
            "delivered_frames_per_s": round(delivered), "ratio": round(delivered / offered, 4),
            "router_messages": msgs, "router_cpu_share": round(cpu / wall, 4) if wall > 0 else None,
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    a = sys.argv[1:]
    out = run(int(a[0]) if len(a) > 0 else 8, int(a[1]) if len(a) > 1 else 8, int(a[2]) if len(a) > 2 else 300,
              (float(a[3]) if len(a) > 3 else 2.5) * 1e-3, a[4] if len(a) > 4 else "async")
    print(json.dumps(out))
