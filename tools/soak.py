"""Soak run of the frame engine at 1.7B shapes: random arrivals (single, packed, prefilled beside the frames singly and
packed, admitted from the engine's hook while a frame is in flight),
prompt lengths from 4 tokens to most of the model's 1024 positions, random frame budgets, greedy and sampling
requests, random cancellations.

Phase 1 (QTTS_SOAK_STEPS engine steps, default 4000, every kind of admission).  Checked while it runs and at the end:
  * every frame holds 16 ids inside the two vocabularies,
  * a stream ends exactly at its frame budget (codec EOS is switched off) unless it was cancelled,
  * every KV block is back in the free list when the last stream has gone.
Phase 2 (QTTS_SOAK_REPLAY_STEPS steps, default 1500, twice, on fresh engines, fixed sampling seed, admissions that do
not depend on GPU timing): both runs must produce the same ids for every stream, sampled ones included - whatever
ran beside a stream, in whatever order the hardware retired it, nothing uninitialised or racing reaches a result.
(Not checked: a stream alone against the same stream in a mix.  The decode attention picks its token split from the
batch's longest context, and a packed prompt goes through other GEMM / attention forms than a lone one: same
arithmetic, other summation order, and on flat random-weight logits the last bit decides ids.)

    python tools/soak.py            # on a GPU box; prints one summary line, exit code 1 on a violation
"""
import json
import os
import random
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine, NoCapacity  # noqa: E402
from nano_qwen3tts_vllm.sampling_params import SamplingParams  # noqa: E402
from nano_qwen3tts_vllm.utils import synthetic  # noqa: E402


def run_schedule(seed, steps, d, H, tv, pv, tw, pw, dev, timing_free):
    rnd = random.Random(seed)
    B = int(os.environ.get("QTTS_SOAK_STREAMS", "8"))      # stream rows of the engine (more than 16: the steps run the prompt-form GEMMs)
    fe = FrameEngine(d, max_streams=B, max_text_frames=320, talker_weights=tw, predictor_weights=pw, num_kvcache_blocks=4 * B,
                     max_num_batched_tokens=4096, codec_eos=None)
    fe.capture_all()
    free0 = len(fe.talker.scheduler.block_manager.free_block_ids)
    gen = torch.Generator().manual_seed(seed + 1)
    greedy = (SamplingParams(temperature=1.0, max_tokens=1, do_sample=False), SamplingParams(temperature=0.9, max_tokens=17, do_sample=False))
    sample = (SamplingParams(temperature=1.0, max_tokens=1, do_sample=True), SamplingParams(temperature=0.9, max_tokens=17, do_sample=True))
    live, finished = {}, {}
    counts = dict(admitted=0, packed=0, overlapped=0, in_flight=0, cancelled=0, no_capacity=0, frames=0, sampled=0)
    n_req = 0

    def new_request():
        nonlocal n_req
        L = rnd.choice([4, 9, 16, 17, 31, 33, 48, 64, 100, 127, 128, 129, 200, 255, 256, 257, 300, 400, 520, 700])
        frames = min(rnd.choice([1, 2, 5, 17, 40, 90, 200, 300]), 1024 - L - 2)
        sp = sample if rnd.random() < 0.35 else greedy
        n_req += 1
        return dict(prompt_embeds=(torch.randn(L, H, generator=gen) * 0.5).to(dev), text_adds=(torch.randn(min(frames, 60) + 1, H, generator=gen) * 0.1).to(dev),
                    max_frames=frames, talker_sampling=sp[0], predictor_sampling=sp[1], sid=f"r{n_req}")

    def track(req):
        live[req["sid"]] = dict(req=req, frames=[])
        counts["admitted"] += 1
        counts["sampled"] += int(req["talker_sampling"].do_sample)

    # detached prefills join with the next frame (join_prefills): which frame that is does not depend on GPU timing, so
    # they are part of the replayed schedule too; joining once the prefill has finished (the other policy) and
    # admissions from the engine's hook while a frame is in flight are timing-dependent and stay in phase 1
    fe.join_prefills = timing_free or rnd.random() < 0.5
    hook = {"reqs": None}

    def poll():
        reqs, hook["reqs"] = hook["reqs"], None
        if reqs:
            for req, res in zip(reqs, fe.add_streams(reqs, wait=False)):
                if isinstance(res, NoCapacity):
                    counts["no_capacity"] += 1
                elif isinstance(res, Exception):
                    raise res
                else:
                    track(req)
                    counts["in_flight"] += 1

    step = 0
    while step < steps or fe.streams:
        step += 1
        if step < steps and len(fe.streams) < B and rnd.random() < (0.5 if B <= 8 else 0.85):
            room = B - len(fe.streams)
            how = rnd.random()
            if how < 0.4 and room >= 2:                       # several at once: one packed prefill
                reqs = [new_request() for _ in range(rnd.randint(2, room))]
                detached = rnd.random() < 0.4                 # ... on the second stream, beside the frames
                for req, res in zip(reqs, fe.add_streams(reqs, wait=not detached)):
                    if isinstance(res, NoCapacity):
                        counts["no_capacity"] += 1
                    elif isinstance(res, Exception):
                        raise res
                    else:
                        track(req)
                        counts["packed"] += 1
                        counts["overlapped"] += int(detached)
            elif how >= 0.9 and not timing_free and fe.streams:
                hook["reqs"] = [new_request() for _ in range(rnd.randint(1, min(2, room)))]     # found while the frame is in flight
            else:
                req = new_request()
                overlap = how >= 0.7
                try:
                    fe.add_stream(**req, wait=not overlap)
                    track(req)
                    counts["overlapped"] += int(overlap)
                except NoCapacity:
                    counts["no_capacity"] += 1
        if live and rnd.random() < 0.03:                      # a consumer goes away
            sid = rnd.choice(sorted(live))
            fe.remove_stream(sid)
            finished[sid] = live.pop(sid)["frames"]
            counts["cancelled"] += 1
        for sid, frame in fe.step(poll=poll if hook["reqs"] else None).items():
            assert sid in live, f"frame for unknown stream {sid}"
            assert len(frame) == 16 and 0 <= frame[0] < tv and all(0 <= c < pv for c in frame[1:]), (sid, frame)
            live[sid]["frames"].append(frame)
            counts["frames"] += 1
        for sid in [s for s in live if s not in fe.streams]:
            rec = live.pop(sid)
            assert len(rec["frames"]) == rec["req"]["max_frames"], (sid, len(rec["frames"]), rec["req"]["max_frames"])
            finished[sid] = rec["frames"]
    assert not live and not fe.streams
    free1 = len(fe.talker.scheduler.block_manager.free_block_ids)
    assert free1 == free0, f"KV blocks leaked: {free0} free at the start, {free1} at the end"
    fe.exit()
    del fe
    torch.cuda.synchronize()
    return finished, counts


def main():
    os.environ.setdefault("QTTS_SEED", "1234")               # the sampler's counter RNG: the same draws in both replay runs
    steps = int(os.environ.get("QTTS_SOAK_STEPS", "4000"))
    replay_steps = int(os.environ.get("QTTS_SOAK_REPLAY_STEPS", "1500"))
    seed = int(os.environ.get("QTTS_SOAK_SEED", "5"))
    dev = torch.device("cuda", 0)
    cfg = synthetic.make_config("1.7B")
    tw, pw = bench.make_weights(cfg, dev)
    d = tempfile.mkdtemp(prefix="qtts_soak_")
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    H = cfg["talker_config"]["hidden_size"]
    tv = cfg["talker_config"]["vocab_size"]
    pv = cfg["talker_config"]["code_predictor_config"]["vocab_size"]
    t0 = time.time()
    finished, counts = run_schedule(seed, steps, d, H, tv, pv, tw, pw, dev, timing_free=False)
    a, ca = run_schedule(seed + 100, replay_steps, d, H, tv, pv, tw, pw, dev, timing_free=True)
    b, _ = run_schedule(seed + 100, replay_steps, d, H, tv, pv, tw, pw, dev, timing_free=True)
    assert sorted(a) == sorted(b), "the two replay runs served different streams"
    differ = [sid for sid in a if a[sid] != b[sid]]
    assert not differ, f"{len(differ)} of {len(a)} streams differ between two runs of the same schedule: {differ[:5]}"
    print(json.dumps(dict(seconds=round(time.time() - t0, 1), stream_rows=int(os.environ.get("QTTS_SOAK_STREAMS", "8")), streams=len(finished), replayed_streams_identical=len(a),
                          replayed_sampled=ca["sampled"], **counts)))


if __name__ == "__main__":
    main()
