"""Host-side load test of the replica router (no GPU): N stand-in replica processes, each producing a frame for every
one of its streams every `step` seconds (the real engine's 2.5 ms), and the parent's one result thread fanning the
frames out.  Reports delivered against offered frames per second and the CPU share of the router thread.

    python tools/replica_load.py [replicas=8] [streams_per_replica=8] [frames=300] [step_ms=2.5] [async|sinks]

async: every stream is consumed by a coroutine of one event loop (generate_async), as an API server does.
sinks: every stream's consumer only counts and timestamps its frames inside the router thread - the router by itself (hundreds of
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


class _CountingSink:
    """A stream's consumer reduced to what the measurement needs: when its first and last frame came, how many."""

    def __init__(self):
        self.n, self.first, self.last, self.reason = 0, None, None, None

    def put(self, item):
        frame, reason = item
        if frame is None:
            self.reason = reason
            return
        self.last = time.perf_counter()
        if self.first is None:
            self.first = self.last
        self.n += 1


def run(n_replicas=8, streams=8, frames=300, step_s=0.0025, consumers="async"):
    import torch
    from nano_qwen3tts_vllm.replicas import ReplicaPool
    os.environ["PYTHONPATH"] = os.path.join(ROOT, "tests") + os.pathsep + ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")
    pool = ReplicaPool(n_replicas, "fake_replica_engine:make", dict(max_streams=streams, step_seconds=step_s, paced=True),
                       devices=[None] * n_replicas, max_streams_per_replica=streams, start_timeout=300)
    total = n_replicas * streams
    text = torch.zeros(1, 8)
    prompt = torch.zeros(2, 8)
    try:
        # every replica serves one short stream first (its first admission imports torch: seconds, not part of the load)
        for rid in [pool.submit(prompt, text, 3, request_id=f"warm{i}") for i in range(n_replicas)]:
            list(pool.frames(rid))
        cpu0, msg0 = pool.router_cpu_seconds, pool.frame_messages
        if consumers == "async":
            async def main():
                async def one(i):
                    n, first, last = 0, None, None
                    async for _ in pool.generate_async(prompt, text, frames, request_id=f"s{i}"):
                        last = time.perf_counter()
                        if first is None:
                            first = last
                        n += 1
                    return n, first, last
                return await asyncio.gather(*(one(i) for i in range(total)))
            res = asyncio.run(main())
        else:
            sinks = [_CountingSink() for _ in range(total)]
            for i, sk in enumerate(sinks):
                pool.submit(prompt, text, frames, request_id=f"s{i}", _sink=sk)
            deadline = time.perf_counter() + 20 * frames * step_s + 30
            while any(sk.reason is None for sk in sinks) and time.perf_counter() < deadline:
                time.sleep(0.005)
            res = [(sk.n, sk.first, sk.last) for sk in sinks]
        cpu, msgs = pool.router_cpu_seconds - cpu0, pool.frame_messages - msg0
    finally:
        pool.close()
    got = sum(r[0] for r in res)
    wall = max(r[2] for r in res) - min(r[1] for r in res)
    offered = total / step_s
    # a stream's own rate between its first and its last frame, summed over the streams: what the consumers were
    # delivered per second while they were being served (the streams do not all start in the same frame)
    delivered = sum((r[0] - 1) / (r[2] - r[1]) for r in res if r[0] > 1)
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
