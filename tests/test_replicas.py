"""CPU, real processes: the data-parallel replica pool (nano_qwen3tts_vllm/replicas.py) with stand-in engines —
least-loaded placement, per-replica GPU pinning, ordered frames, waiting when every replica is full, cancellation,
engine failures, start-up failures."""
import asyncio

import pytest
import torch

from nano_qwen3tts_vllm.replicas import ReplicaPool

FACTORY = "fake_replica_engine:make"


def _prompt(key, L=4):
    p = torch.zeros(L, 8)
    p[0, 0] = key
    return p


@pytest.fixture()
def pool():
    p = ReplicaPool(2, FACTORY, dict(max_streams=2, step_seconds=0.005), max_streams_per_replica=2, start_timeout=120)
    yield p
    p.close()


def test_least_loaded_placement_and_ordered_frames(pool):
    text = torch.zeros(3, 8)
    rids = [pool.submit(_prompt(10 + i), text, max_frames=20 + 5 * i, request_id=f"r{i}") for i in range(4)]
    # four streams over two replicas with room for two each: alternating, lowest index first
    assert pool.placements == [("r0", 0), ("r1", 1), ("r2", 0), ("r3", 1)]
    assert pool.load() == [2, 2]
    with pytest.raises(ValueError, match="already in use"):
        pool.submit(_prompt(99), text, max_frames=3, request_id="r0")
    for i, rid in enumerate(rids):
        frames = list(pool.frames(rid))
        assert len(frames) == 20 + 5 * i and all(len(f) == 16 for f in frames)
        assert [f[1] for f in frames] == list(range(len(frames)))            # in order
        assert all(f[0] == 10 + i for f in frames)
        # the replica it was placed on served it, pinned to its own GPU index
        assert {f[2] for f in frames} == {i % 2}
    assert pool.load() == [0, 0]


def test_requests_wait_for_a_free_replica_and_cancel(pool):
    text = torch.zeros(2, 8)
    long_ids = [pool.submit(_prompt(i), text, max_frames=400, request_id=f"long{i}") for i in range(4)]
    waiting = pool.submit(_prompt(77), text, max_frames=3, request_id="late")
    assert pool.load() == [2, 2] and ("late", 0) not in pool.placements and ("late", 1) not in pool.placements
    it = pool.frames("long1")
    assert next(it)[0] == 1
    it.close()                                        # the consumer goes away: the stream is cancelled on its replica
    got = list(pool.frames(waiting))                  # ... and the waiting request takes the freed place
    assert [f[1] for f in got] == [0, 1, 2]
    assert dict(pool.placements)["late"] == 1
    for rid in long_ids:
        pool.cancel(rid)
    assert pool.load() == [0, 0]


def test_generate_async_streams_concurrently(pool):
    text = torch.zeros(2, 8)

    async def run():
        async def one(i):
            return [f async for f in pool.generate_async(_prompt(30 + i), text, max_frames=6, request_id=f"a{i}")]
        return await asyncio.gather(*(one(i) for i in range(4)))

    res = asyncio.run(run())
    for i, frames in enumerate(res):
        assert [f[:2] for f in frames] == [[30 + i, n] for n in range(6)]
    assert sorted(r for _, r in pool.placements) == [0, 0, 1, 1]


def test_rejected_and_failing_streams_end_with_an_error():
    pool = ReplicaPool(1, FACTORY, dict(max_streams=2, fail_on=5), devices=[None], max_streams_per_replica=2, start_timeout=120)
    try:
        with pytest.raises(RuntimeError, match="rejected"):
            list(pool.generate(_prompt(1, L=1000), torch.zeros(1, 8), max_frames=100))
        with pytest.raises(RuntimeError, match="injected engine failure"):
            list(pool.generate(_prompt(5), torch.zeros(1, 8), max_frames=4))
        ok = list(pool.generate(_prompt(6), torch.zeros(1, 8), max_frames=4))          # the replica keeps serving
        assert len(ok) == 4 and ok[0][2] == -1                                         # devices=[None]: no pinning
    finally:
        pool.close()


def test_a_replica_process_that_dies_ends_its_streams_and_is_routed_around():
    """Replica 0's process exits in the middle of a stream (key 7 kills whichever replica serves it): that stream ends
    with an error naming the replica, the other replica keeps serving, and later requests go to the survivor only."""
    pool = ReplicaPool(2, FACTORY, dict(max_streams=2, die_on=7, step_seconds=0.01), devices=[None, None],
                       max_streams_per_replica=2, start_timeout=120)
    try:
        other = pool.submit(_prompt(3), torch.zeros(1, 8), max_frames=400, request_id="long")   # on replica 0, dies with it
        survivor = pool.submit(_prompt(4), torch.zeros(1, 8), max_frames=30, request_id="ok")   # on replica 1
        with pytest.raises(RuntimeError, match="replica 0 died"):
            list(pool.generate(_prompt(7), torch.zeros(1, 8), max_frames=50, request_id="killer"))      # least loaded: replica 0
        with pytest.raises(RuntimeError, match="replica 0 died"):
            list(pool.frames(other))
        assert len(list(pool.frames(survivor))) == 30
        assert dict(pool.placements)["killer"] == 0 and dict(pool.placements)["ok"] == 1
        for i in range(3):                                                             # only replica 1 is left
            assert len(list(pool.generate(_prompt(20 + i), torch.zeros(1, 8), max_frames=3, request_id=f"after{i}"))) == 3
            assert dict(pool.placements)[f"after{i}"] == 1
        assert pool.load() == [0, 0]
    finally:
        pool.close()


def test_start_up_failure_is_reported():
    with pytest.raises(RuntimeError, match="no engine today"):
        ReplicaPool(1, "fake_replica_engine:broken", {}, start_timeout=120)


@pytest.mark.parametrize("streams,consumers", [(8, "async"), (64, "sinks")])
def test_router_keeps_up_with_eight_replicas(streams, consumers):
    """Eight stand-in replica processes, each finishing a frame for all of its streams every 2.5 ms (the real engine's
    frame time): the parent's one result thread must hand the consumers at least 95 % of what the replicas offer -
    8 streams per replica with every stream consumed by a coroutine of one event loop (BASELINE's 64 streams over 8
    GPUs, as an API process serves them), and 64 streams per replica (204 800 frames a second) with consumers that only
    count, i.e. the router thread by itself.  tools/replica_load.py is the same measurement as a script and also
    reports the router thread's CPU share."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import replica_load
    best = None
    for _ in range(3):                                  # (further tries on a machine that is busy with something else)
        r = replica_load.run(8, streams, 200, 0.0025, consumers)
        assert r["frames_delivered"] == r["frames_expected"], r
        best = r if best is None or r["ratio"] > best["ratio"] else best
        if best["ratio"] >= 0.95:
            break
    print(f"[replicas] {best}")
    assert best["ratio"] >= 0.95, best
    # one message per replica and frame whatever the number of streams: far fewer messages than frames (how many
    # frames a replica runs beyond the 200 of a stream depends on how far apart its streams were admitted, i.e. on the
    # machine's load, so the bound is on frames per message: at least four on average)
    assert best["router_messages"] * 4 <= best["frames_delivered"], best


@pytest.mark.gpu
def test_two_replica_processes_with_the_real_engine(tmp_path):
    """Two replica processes, each with its own FrameEngine on the HIP path (both pinned to GPU 0 here: the test box
    has one), fed four greedy streams through the router: every stream's frames equal a run of that stream on an
    engine in this process, whichever replica served it."""
    import json
    import os
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    from nano_qwen3tts_vllm.sampling_params import SamplingParams
    from nano_qwen3tts_vllm.utils import synthetic
    d = str(tmp_path / "model")
    synthetic.write_model_dir(d, "tiny", seed=11, norm_jitter=0.1, embed_std=0.05, head_sharp=1.0, head_std=0.05)
    H = synthetic.make_config("tiny")["talker_config"]["hidden_size"]
    gen = torch.Generator().manual_seed(2)
    prompts = [torch.randn(L, H, generator=gen) * 0.5 for L in (7, 30, 12, 19)]
    texts = [torch.randn(8, H, generator=gen) * 0.3 for _ in prompts]
    tsp = SamplingParams(temperature=1.0, max_tokens=1, do_sample=False)
    psp = SamplingParams(temperature=0.9, max_tokens=17, do_sample=False)
    kw = dict(max_streams=2, max_text_frames=16, num_kvcache_blocks=4, max_num_batched_tokens=1024, codec_eos=None)
    fe = FrameEngine(d, **kw)
    want = []
    for p, t in zip(prompts, texts):
        sid = fe.add_stream(p, t, max_frames=6, talker_sampling=tsp, predictor_sampling=psp)
        want.append(fe.run_frames(6)[:, 0].tolist())
        fe.remove_stream(sid)
    fe.exit()
    pool = ReplicaPool(2, "nano_qwen3tts_vllm.replicas:frame_engine_factory", dict(model_dir=d, capture=False, **kw),
                       devices=[0, 0], max_streams_per_replica=2, start_timeout=300)
    try:
        rids = [pool.submit(p, t, 6, talker_sampling=tsp, predictor_sampling=psp, request_id=f"s{i}")
                for i, (p, t) in enumerate(zip(prompts, texts))]
        assert sorted(r for _, r in pool.placements) == [0, 0, 1, 1]
        got = [list(pool.frames(rid, timeout=120)) for rid in rids]
    finally:
        pool.close()
    assert got == want


def test_replica_admits_requests_while_a_frame_is_in_flight():
    """A replica whose engine offers the serving hooks (step(poll=...), add_streams(wait=False)) takes commands and
    admits what fits from inside the engine's step: requests submitted while the first streams run are admitted
    mid-frame (slot 4 of their frames says so), every stream still gets exactly its frames in order, a cancellation
    that arrives mid-frame takes effect, and the loads return to zero."""
    p = ReplicaPool(1, "fake_replica_engine:make_polling", dict(max_streams=4, step_seconds=0.006),
                    max_streams_per_replica=4, start_timeout=120)
    try:
        text = torch.zeros(2, 8)
        first = p.submit(_prompt(5), text, max_frames=60, request_id="first")
        it = p.frames(first)
        assert next(it)[1] == 0                              # the replica is stepping now
        rids = [p.submit(_prompt(20 + i), text, max_frames=10 + i, request_id=f"mid{i}") for i in range(2)]
        doomed = p.submit(_prompt(99), text, max_frames=500, request_id="doomed")
        got = {rid: list(p.frames(rid)) for rid in rids}
        for i, rid in enumerate(rids):
            assert [f[1] for f in got[rid]] == list(range(10 + i)) and all(f[0] == 20 + i for f in got[rid])
        assert any(f[4] == 1 for rid in rids for f in got[rid]), "nothing was admitted from inside a step"
        p.cancel(doomed)
        rest = list(it)
        assert [f[1] for f in rest] == list(range(1, 60))
        assert p.load() == [0]
    finally:
        p.close()
