"""GPU parity of the device-resident frame loop (FrameEngine: the predictor burst + glue and the talker step as
hipGraphs, ids and hidden states never leaving the device) against the reference fixtures and the CPU oracle.
Every greedy pick is judged against the oracle's logits of the same history (helpers.check_frames): it must be the
arg-max or within TOL_ULPS bf16 steps of it; the tests print how many picks were binding."""
import json
import os

import numpy as np
import pytest
import torch

import qtts_oracle as O
from helpers import FIXTURES, check_frames, fixture_model, golden_weights, oracle_cfgs, ulp_distance

pytestmark = pytest.mark.gpu
TOL_ULPS = 2.0
LOGIT_ULPS = 4.0
MIN_BINDING = {"tiny": 0.7, "tiny_id": 0.7, "sharp": 0.9, "sharp_id": 0.9}


def greedy(max_tokens, temperature=1.0):
    from nano_qwen3tts_vllm.sampling_params import SamplingParams
    return SamplingParams(temperature=temperature, max_tokens=max_tokens, do_sample=False)


@pytest.mark.parametrize("tag", list(FIXTURES))
@pytest.mark.parametrize("eager", [False, True])
def test_frame_engine_matches_fixture(tmp_path, tag, eager):
    """2 streams x 6 frames of the reference fixture, frame by frame: the 192 picks, the logits behind each of them
    (talker head and the 15 predictor heads) and the hidden state that opens each frame."""
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    cfg, sd, tw, pw, g = fixture_model(tag)
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    S, NF = 2, 6
    prompts = [torch.from_numpy(g[f"f_prompt{i}"]) for i in range(S)]
    text_adds = torch.from_numpy(g["f_text_adds"])
    fe = FrameEngine(str(tmp_path), max_streams=4, max_text_frames=8, talker_weights=tw, predictor_weights=pw,
                     enforce_eager=eager, num_kvcache_blocks=8, max_num_batched_tokens=1024, codec_eos=None)
    t_logits, hidden, p_logits, got = [[] for _ in range(S)], [[] for _ in range(S)], [], []
    for i in range(S):
        fe.add_stream(prompts[i], text_adds[i], max_frames=NF, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
        torch.cuda.synchronize()
        t_logits[i].append(fe.tr.logits[0].float().cpu())
        hidden[i].append(fe.cur_hidden[i].float().cpu())
    for f in range(NF):
        got.append(fe.run_frames(1)[0, :S])
        p_logits.append(fe.pr.burst_logits[:, :S].float().cpu().permute(1, 0, 2))
        for i in range(S):
            t_logits[i].append(fe.tr.logits[i].float().cpu())
            hidden[i].append(fe.cur_hidden[i].float().cpu())
    got = torch.stack(got, 1)
    fe.exit()
    tc, pc = oracle_cfgs(cfg)
    r = check_frames(got, tw, tc, pw, pc, prompts, text_adds, f"frame engine {tag} eager={eager}", TOL_ULPS, MIN_BINDING[tag],
                     fixture_frames=g["f_frames"], fixture_tag=tag)
    gt = torch.stack([torch.stack(t[:NF]) for t in t_logits])
    gh = torch.stack([torch.stack(h[:NF]) for h in hidden])
    gp = torch.stack(p_logits, 1)
    wt, wp = float(ulp_distance(gt, r["t_logits"]).max()), float(ulp_distance(gp, r["p_logits"]).max())
    print(f"[logits] frame engine {tag} eager={eager}: talker within {wt:.2f}, predictor within {wp:.2f} bf16 steps")
    assert wt <= LOGIT_ULPS and wp <= LOGIT_ULPS
    assert torch.allclose(gh, r["hidden"], atol=4e-2, rtol=4e-2), float((gh - r["hidden"]).abs().max())
    if np.array_equal(got.numpy(), g["f_frames"]):                # same history as the reference's own run
        assert float(ulp_distance(gt, g["f_t_logits"]).max()) <= LOGIT_ULPS
        assert float(ulp_distance(gp, g["f_p_logits"]).max()) <= LOGIT_ULPS


def test_frame_engine_step_eos_and_late_join(tmp_path):
    """step(): a stream ends when the talker emits the codec EOS id; a stream admitted later
    joins the running batch without disturbing the others (ids equal a solo run)."""
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    cfg, sd, tw, pw = golden_weights("tiny", 11)
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    gen = torch.Generator().manual_seed(5)
    H = cfg["talker_config"]["hidden_size"]
    pa, pb = torch.randn(9, H, generator=gen) * 0.5, torch.randn(300, H, generator=gen) * 0.5
    ta, tb = torch.randn(12, H, generator=gen) * 0.3, torch.randn(12, H, generator=gen) * 0.3
    kw = dict(max_streams=4, max_text_frames=16, talker_weights=tw, predictor_weights=pw, num_kvcache_blocks=12,
              max_num_batched_tokens=1024)

    def solo(prompt, text, n):
        fe = FrameEngine(str(tmp_path), codec_eos=None, **kw)
        fe.add_stream(prompt, text, max_frames=n, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
        out = fe.run_frames(n)[:, 0].tolist()
        fe.exit()
        return out

    solo_a, solo_b = solo(pa, ta, 8), solo(pb, tb, 5)
    eos = solo_a[5][0]                       # make stream a end at its 6th frame
    fe = FrameEngine(str(tmp_path), codec_eos=eos, **kw)
    a = fe.add_stream(pa, ta, max_frames=8, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
    got_a, got_b, b = [], [], None
    for f in range(8):
        if f == 3:
            b = fe.add_stream(pb, tb, max_frames=5, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
        res = fe.step()
        if a in res:
            got_a.append(res[a])
        if b is not None and b in res:
            got_b.append(res[b])
    first_eos = next(i for i, fr in enumerate(solo_a) if fr[0] == eos)
    assert got_a == solo_a[:first_eos]        # frames before the EOS frame, EOS frame not yielded
    assert got_b == solo_b[:len(got_b)] and len(got_b) == 5
    assert a not in fe.streams
    fe.exit()


def test_capture_all_then_no_capture_while_serving(tmp_path):
    """Every (batch bucket, context bucket) graph is captured up front; serving a stream across a context
    bucket boundary then reuses them (no new graph) and yields the ids of an engine that captures lazily."""
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    cfg, sd, tw, pw = golden_weights("tiny", 11)
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    gen = torch.Generator().manual_seed(9)
    H = cfg["talker_config"]["hidden_size"]
    prompt, text = torch.randn(120, H, generator=gen) * 0.5, torch.randn(4, H, generator=gen) * 0.3
    kw = dict(max_streams=2, max_text_frames=16, talker_weights=tw, predictor_weights=pw, num_kvcache_blocks=8,
              max_num_batched_tokens=1024, codec_eos=None)
    outs = []
    for eager_capture in (True, False):
        fe = FrameEngine(str(tmp_path), **kw)
        if eager_capture:
            n = fe.capture_all()
            assert n == 2 * 8                 # batch buckets {1, 2} x context buckets {128, ..., 1024}
        fe.add_stream(prompt, text, max_frames=12, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
        frames = [fe.step()["stream-0"] for _ in range(12)]          # context 121 -> 133: crosses the 128 boundary
        if eager_capture:
            assert len(fe.graphs) == n
        outs.append(frames)
        fe.exit()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("max_streams", [3, 5, 12, 24])
def test_stream_counts_between_the_power_of_two_buckets(tmp_path, max_streams):
    """An engine sized for 3, 5, 12 or 24 streams (not one of the batch buckets 1, 2, 4, 8, 16, 32 ...) runs its full
    batch, and every stream's ids equal a solo run of the same prompt (12 streams: the 16-row steps hand the attention
    and SwiGLU outputs to the next GEMM in fragment order, a solo run row-major)."""
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    cfg, sd, tw, pw = golden_weights("tiny", 11)
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    gen = torch.Generator().manual_seed(max_streams)
    H = cfg["talker_config"]["hidden_size"]
    prompts = [torch.randn(5 + (i % 7), H, generator=gen) * 0.5 for i in range(max_streams)]
    texts = [torch.randn(6, H, generator=gen) * 0.3 for _ in range(max_streams)]
    kw = dict(max_text_frames=8, talker_weights=tw, predictor_weights=pw, max_num_batched_tokens=1024, codec_eos=None)
    fe = FrameEngine(str(tmp_path), max_streams=max_streams, num_kvcache_blocks=max_streams + 1, **kw)
    for p, t in zip(prompts, texts):
        fe.add_stream(p, t, max_frames=4, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
    got = fe.run_frames(4).permute(1, 0, 2)[:max_streams].numpy()
    fe.exit()
    solo = FrameEngine(str(tmp_path), max_streams=1, num_kvcache_blocks=2, **kw)
    for i in (0, max_streams - 1):
        sid = solo.add_stream(prompts[i], texts[i], max_frames=4, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
        want = solo.run_frames(4).permute(1, 0, 2)[0].numpy()
        solo.remove_stream(sid)
        assert np.array_equal(got[i], want), i
    solo.exit()


@pytest.mark.parametrize("B,weights", [(8, "bf16"), (12, "bf16"), (12, "fp8")])
def test_full_size_batch_invariance_and_graph_equals_eager(tmp_path, monkeypatch, B, weights):
    """Qwen3-TTS-1.7B shapes (28 + 5 layers, the bench's random-init weights), 8 or 12 streams, 3 frames.  Two properties
    that hold at any size because every row of a step is computed independently of the others and a graph replays
    the launches it captured: (1) the frame graphs return exactly the ids of eager launches; (2) a stream run alone
    returns exactly the ids it gets inside the batch (equal prompt lengths, so the attention split is the same).
    With 12 streams the steps are 16 rows wide: no shared prologue lanes, activations between the kernels of a layer
    in fragment order - other kernel forms than the solo run's, the same bits."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    from nano_qwen3tts_vllm.utils import synthetic
    monkeypatch.setenv("QTTS_WEIGHT_DTYPE", weights)
    cfg = synthetic.make_config("1.7B")
    dev = torch.device("cuda")
    tw, pw = bench.make_weights(cfg, dev)
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    H, NF, L = cfg["talker_config"]["hidden_size"], 3, 100
    g = torch.Generator().manual_seed(77)
    prompts = [(torch.randn(L, H, generator=g) * 0.5) for _ in range(B)]
    texts = [(torch.randn(NF + 2, H, generator=g) * 0.1) for _ in range(B)]

    def run(which, eager):
        fe = FrameEngine(str(tmp_path), max_streams=B, max_text_frames=NF + 2, talker_weights=tw, predictor_weights=pw,
                         enforce_eager=eager, num_kvcache_blocks=4 * B, max_num_batched_tokens=4096, codec_eos=None)
        for i in which:
            fe.add_stream(prompts[i].to(dev), texts[i].to(dev), max_frames=NF + 1, talker_sampling=greedy(1),
                          predictor_sampling=greedy(17, 0.9))
        ids = fe.run_frames(NF).permute(1, 0, 2)[:len(which)].clone()
        assert fe.tr.decoder.weight_dtype == weights and fe.pr.decoder.weight_dtype == weights
        fe.exit()
        return ids

    batch = run(range(B), False)
    assert batch.shape == (B, NF, 16)
    assert torch.equal(batch, run(range(B), True)), "graph replay differs from eager launches"
    assert torch.equal(batch[3:4], run([3], False)), "a stream's ids depend on its batch"


def test_a_stream_pushed_past_its_frame_budget_cannot_touch_another_streams_kv(tmp_path):
    """Stream B holds physical KV block 0; stream A (250-token prompt, 4 frames: one KV block reserved, the rest of its
    block-table row zero) is then driven far past its budget through the engine's launch internals.  The device-side
    metadata advance freezes A at the context its block covers, so A never follows a table entry that is not its own
    and B's ids equal a run of B alone.  The public calls refuse to go past the budget in the first place."""
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    cfg, sd, tw, pw = golden_weights("tiny", 11)
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    gen = torch.Generator().manual_seed(23)
    H = cfg["talker_config"]["hidden_size"]
    pb, pa = torch.randn(12, H, generator=gen) * 0.5, torch.randn(250, H, generator=gen) * 0.5
    tb, ta = torch.randn(20, H, generator=gen) * 0.3, torch.randn(4, H, generator=gen) * 0.3
    kw = dict(max_streams=2, max_text_frames=24, talker_weights=tw, predictor_weights=pw, num_kvcache_blocks=4,
              max_num_batched_tokens=1024, codec_eos=None)
    solo = FrameEngine(str(tmp_path), **kw)
    solo.add_stream(pb, tb, max_frames=16, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
    want = solo.run_frames(16)[:, 0].tolist()
    solo.exit()

    fe = FrameEngine(str(tmp_path), **kw)
    b = fe.add_stream(pb, tb, max_frames=16, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
    a = fe.add_stream(pa, ta, max_frames=4, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
    assert fe.streams[b].seq.block_table == [0] and len(fe.streams[a].seq.block_table) == 1
    with pytest.raises(ValueError, match="max_frames"):
        fe.run_frames(5)                                   # more than A has left
    got = []
    with fe.tr.on_stream():                                # the launches themselves, 16 frames, no budget check
        fe._setup_predictor_meta(2)
        for f in range(16):
            fe._launch_codes(2)
            got.append(fe.codes[0].tolist())
            fe._launch_talker(2, 384)
        fe.tr.stream.synchronize()
    assert got == want
    st = fe.tr.dstage
    assert int(st.d("ctx")[1]) == 250 + 4                  # A stopped where its reserved block ends being covered
    fe.exit()


def test_step_finishes_a_stream_at_its_frame_budget(tmp_path):
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    cfg, sd, tw, pw = golden_weights("tiny", 11)
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    gen = torch.Generator().manual_seed(4)
    H = cfg["talker_config"]["hidden_size"]
    fe = FrameEngine(str(tmp_path), max_streams=2, max_text_frames=8, talker_weights=tw, predictor_weights=pw,
                     num_kvcache_blocks=4, max_num_batched_tokens=1024, codec_eos=None)
    free0 = len(fe.talker.scheduler.block_manager.free_block_ids)
    a = fe.add_stream(torch.randn(6, H, generator=gen) * 0.5, torch.zeros(1, H), max_frames=3, talker_sampling=greedy(1),
                      predictor_sampling=greedy(17))
    b = fe.add_stream(torch.randn(9, H, generator=gen) * 0.5, torch.zeros(1, H), max_frames=5, talker_sampling=greedy(1),
                      predictor_sampling=greedy(17))
    seen = {a: 0, b: 0}
    for _ in range(5):
        for sid in fe.step():
            seen[sid] += 1
    assert seen == {a: 3, b: 5} and not fe.streams and fe.step() == {}
    assert len(fe.talker.scheduler.block_manager.free_block_ids) == free0
    # a request's own top_k is accepted and ignored, as the reference's sampler ignores it (P/layers/sampler.py:15)
    from nano_qwen3tts_vllm.sampling_params import SamplingParams
    t = fe.add_stream(torch.zeros(4, H), torch.zeros(1, H), max_frames=2,
                      talker_sampling=SamplingParams(temperature=1.0, max_tokens=1, do_sample=True, top_k=20))
    assert len([f for _ in range(2) for f in fe.step().values()]) == 2 and t not in fe.streams
    # a stream id that is in use is refused (the second stream would orphan the first one's row and KV blocks)
    c = fe.add_stream(torch.zeros(4, H), torch.zeros(1, H), max_frames=2, sid="same")
    with pytest.raises(ValueError, match="already in use"):
        fe.add_stream(torch.zeros(4, H), torch.zeros(1, H), max_frames=2, sid="same")
    fe.remove_stream(c)
    assert len(fe.talker.scheduler.block_manager.free_block_ids) == free0 and fe.rows == [None, None]
    fe.exit()


def test_a_sampling_request_draws_the_same_codes_whichever_way_it_is_admitted(tmp_path, monkeypatch):
    """With a fixed sampler seed (QTTS_SEED) a sampling request's ids do not depend on the admission mode: prefilled
    between two frames (wait=True) or on the second stream beside them (wait=False), its first code is drawn at the same
    RNG offset, and the frames that follow draw from the same device-side counter."""
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    from nano_qwen3tts_vllm.sampling_params import SamplingParams
    monkeypatch.setenv("QTTS_SEED", "1234")
    cfg, sd, tw, pw = golden_weights("tiny", 11)
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    H = cfg["talker_config"]["hidden_size"]
    gen = torch.Generator().manual_seed(8)
    prompt, text = torch.randn(7, H, generator=gen) * 0.5, torch.randn(6, H, generator=gen) * 0.3
    sp = dict(talker_sampling=SamplingParams(temperature=1.0, max_tokens=1, do_sample=True, top_k=50),
              predictor_sampling=SamplingParams(temperature=0.9, max_tokens=17, do_sample=True, top_k=50))
    runs = []
    for wait in (True, False):
        fe = FrameEngine(str(tmp_path), max_streams=2, max_text_frames=8, talker_weights=tw, predictor_weights=pw,
                         num_kvcache_blocks=4, max_num_batched_tokens=1024, codec_eos=None)
        sid = fe.add_stream(prompt, text, max_frames=5, wait=wait, **sp)
        frames = []
        while fe.streams:
            frames += list(fe.step().values())
        fe.exit()
        runs.append(frames)
        assert len(frames) == 5
    assert runs[0] == runs[1]
    greedy_run = runs[0][0][0]
    assert any(fr != runs[0][0] for fr in runs[0][1:]) and 0 <= greedy_run < cfg["talker_config"]["vocab_size"]


@pytest.mark.parametrize("join", [True, False])
def test_prompts_prefilled_beside_the_running_frames_give_the_same_ids(tmp_path, join):
    """add_stream(wait=False): the prompt is prefilled on a second HIP stream while the running streams' frames go on,
    and joins at a frame boundary - the next one, which then waits for the rest of the prefill (join_prefills, the
    default), or the first one after the prefill has finished.  Three newcomers admitted at once (their prefills queue
    up), one of them cancelled while it waits: every stream's ids equal a run of that stream alone, the running stream
    is not disturbed, and all KV blocks come back."""
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    cfg, sd, tw, pw, g = fixture_model("sharp")
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    gen = torch.Generator().manual_seed(31)
    H = cfg["talker_config"]["hidden_size"]
    prompts = [torch.randn(L, H, generator=gen) * 0.5 for L in (9, 300, 40, 17)]
    texts = [torch.randn(12, H, generator=gen) * 0.3 for _ in prompts]
    kw = dict(max_streams=4, max_text_frames=16, talker_weights=tw, predictor_weights=pw, num_kvcache_blocks=12,
              max_num_batched_tokens=1024, codec_eos=None)

    def solo(i, n):
        fe = FrameEngine(str(tmp_path), **kw)
        fe.add_stream(prompts[i], texts[i], max_frames=n, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
        out = fe.run_frames(n)[:, 0].tolist()
        fe.exit()
        return out

    want = [solo(0, 12), solo(1, 6), solo(2, 6), solo(3, 6)]
    fe = FrameEngine(str(tmp_path), **kw)
    fe.join_prefills = join
    free0 = len(fe.talker.scheduler.block_manager.free_block_ids)
    sp = dict(talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
    a = fe.add_stream(prompts[0], texts[0], max_frames=12, **sp)
    got = {a: []}
    ids = {}
    for f in range(12):
        if f == 2:
            for i in (1, 2, 3):
                ids[i] = fe.add_stream(prompts[i], texts[i], max_frames=6, wait=False, **sp)
                got[ids[i]] = []
            assert not fe.streams[ids[1]].active or fe.streams[ids[1]].ready is not None
            fe.remove_stream(ids[3])                      # leaves while its prefill has not even started
        for sid, frame in fe.step().items():
            got[sid].append(frame)
    while fe.streams:
        for sid, frame in fe.step().items():
            got[sid].append(frame)
    assert got[a] == want[0]
    assert got[ids[1]] == want[1] and got[ids[2]] == want[2] and got[ids[3]] == []
    assert len(fe.talker.scheduler.block_manager.free_block_ids) == free0
    # with nothing running, step() waits for the first prefill instead of returning nothing
    b = fe.add_stream(prompts[3], texts[3], max_frames=6, wait=False, **sp)
    frames = []
    while fe.streams:
        frames += [fr for fr in fe.step().values()]
    assert frames == want[3]
    fe.exit()


def test_requests_admitted_while_a_frame_is_in_flight_give_the_same_ids(tmp_path):
    """step(poll=...): the serving loop's hook is called while the host waits for a frame's codes; requests admitted
    from it (two at once: one packed prefill on the second stream; later a third into the row a finished stream has
    left, i.e. with KV blocks that were in use a frame before; one of them sampling its first code) start beside the
    frame in flight and join with the next frame.  Ids of the greedy streams equal a run of each alone, the stream
    that was running is not disturbed, every KV block comes back."""
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    from nano_qwen3tts_vllm.sampling_params import SamplingParams
    cfg, sd, tw, pw, g = fixture_model("sharp")
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    gen = torch.Generator().manual_seed(77)
    H = cfg["talker_config"]["hidden_size"]
    prompts = [torch.randn(L, H, generator=gen) * 0.5 for L in (11, 270, 35, 20, 8)]
    texts = [torch.randn(12, H, generator=gen) * 0.3 for _ in prompts]
    kw = dict(max_streams=4, max_text_frames=16, talker_weights=tw, predictor_weights=pw, num_kvcache_blocks=8,
              max_num_batched_tokens=1024, codec_eos=None)
    sp = dict(talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))

    def solo(i, n):
        fe = FrameEngine(str(tmp_path), **kw)
        fe.add_stream(prompts[i], texts[i], max_frames=n, **sp)
        out = fe.run_frames(n)[:, 0].tolist()
        fe.exit()
        return out

    want = [solo(0, 10), solo(1, 3), solo(2, 5), solo(3, 4)]
    fe = FrameEngine(str(tmp_path), **kw)
    free0 = len(fe.talker.scheduler.block_manager.free_block_ids)
    a = fe.add_stream(prompts[0], texts[0], max_frames=10, **sp)
    got, ids, frame_no, polls = {a: []}, {}, [0], [0]

    def poll():
        polls[0] += 1
        if frame_no[0] == 1 and 1 not in ids:             # two requests found waiting during frame 1
            r = fe.add_streams([dict(prompt_embeds=prompts[i], text_adds=texts[i], max_frames=n, sid=f"s{i}", **sp)
                                for i, n in ((1, 3), (2, 5))], wait=False)
            assert r == ["s1", "s2"]
            ids[1], ids[2] = r
        if frame_no[0] == 5 and 3 not in ids:             # stream 1 has finished (3 frames): its row and blocks are free
            assert ids[1] not in fe.streams
            ids[3] = fe.add_stream(prompts[3], texts[3], max_frames=4, wait=False, sid="s3", **sp)
            ids[4] = fe.add_stream(prompts[4], texts[4], max_frames=2, wait=False, sid="s4", predictor_sampling=greedy(17, 0.9),
                                   talker_sampling=SamplingParams(temperature=1.0, max_tokens=1, do_sample=True, top_k=50))
    while fe.streams:
        for sid, frame in fe.step(poll=poll).items():
            got.setdefault(sid, []).append(frame)
        frame_no[0] += 1
    assert polls[0] > 0 and len(ids) == 4
    assert got[a] == want[0]
    assert got["s1"] == want[1] and got["s2"] == want[2] and got["s3"] == want[3]
    # newcomers found during frame 1 have their first frame two calls later at the latest (frame 2 waits for them)
    assert len(got["s4"]) == 2 and all(0 <= c < 512 for fr in got["s4"] for c in fr)
    assert len(fe.talker.scheduler.block_manager.free_block_ids) == free0
    # the polling form's commit margin follows the measured commit time between its bounds, unless an instance pins it
    assert fe.LAUNCH_MARGIN_MIN_S <= fe.launch_margin_s() <= fe.LAUNCH_MARGIN_MAX_S
    fe.LAUNCH_MARGIN_S = 4e-4
    assert fe.launch_margin_s() == 4e-4
    fe.exit()


def test_burst_steps_without_metadata_arrays_give_the_same_ids(tmp_path, monkeypatch):
    """In the frame engine row r of the predictor's burst is always sequence r in predictor KV block r at context
    step + 2, so the burst's decode steps are launched without positions / slots / context lengths / block tables
    (QTTS_UNIFORM_BURST=1, the default).  Same arithmetic: ids, every step's logits and the predictor's KV cache
    equal, bit for bit, the launches that read the metadata arrays."""
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    cfg, sd, tw, pw, g = fixture_model("sharp")
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    gen = torch.Generator().manual_seed(8)
    H = cfg["talker_config"]["hidden_size"]
    prompts = [torch.randn(L, H, generator=gen) * 0.5 for L in (5, 33, 12)]
    texts = [torch.randn(6, H, generator=gen) * 0.3 for _ in prompts]
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("QTTS_UNIFORM_BURST", flag)
        fe = FrameEngine(str(tmp_path), max_streams=4, max_text_frames=8, talker_weights=tw, predictor_weights=pw,
                         num_kvcache_blocks=8, max_num_batched_tokens=1024, codec_eos=None)
        for p, t in zip(prompts, texts):
            fe.add_stream(p, t, max_frames=5, talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))
        ids = fe.run_frames(5).clone()
        res[flag] = (ids, fe.pr.burst_logits[:, :3].clone(), fe.pr.decoder.kv_cache[:, :, :3, :17].clone())
        fe.exit()
    assert torch.equal(res["1"][0], res["0"][0])
    assert torch.equal(res["1"][1], res["0"][1]) and torch.equal(res["1"][2], res["0"][2])


def test_prompts_admitted_together_are_prefilled_packed_and_give_the_same_ids(tmp_path):
    """add_streams: the prompts of several requests go through ONE packed prefill (ragged lengths, one of them
    longer than a KV block).  Every stream's ids equal a run of that stream admitted alone; a request that can never
    be served and one that does not fit right now come back as their exceptions, in order, without disturbing the
    others; prompts that together exceed the runner's row limit (600 + 500 > 1024) are packed in two groups."""
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine, NoCapacity
    cfg, sd, tw, pw, g = fixture_model("sharp")
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    gen = torch.Generator().manual_seed(77)
    H = cfg["talker_config"]["hidden_size"]
    prompts = [torch.randn(L, H, generator=gen) * 0.5 for L in (9, 300, 40, 130, 600, 500)]
    texts = [torch.randn(8, H, generator=gen) * 0.3 for _ in prompts]
    kw = dict(max_streams=4, max_text_frames=16, talker_weights=tw, predictor_weights=pw, num_kvcache_blocks=12,
              max_num_batched_tokens=1024, codec_eos=None)
    sp = dict(talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))

    def solo(i):
        fe = FrameEngine(str(tmp_path), **kw)
        fe.add_stream(prompts[i], texts[i], max_frames=5, **sp)
        out = fe.run_frames(5)[:, 0].tolist()
        fe.exit()
        return out

    want = [solo(i) for i in range(6)]
    fe = FrameEngine(str(tmp_path), **kw)
    free0 = len(fe.talker.scheduler.block_manager.free_block_ids)
    reqs = [dict(prompt_embeds=prompts[0], text_adds=texts[0], max_frames=5, sid="a", **sp),
            dict(prompt_embeds=torch.zeros(1000, H), text_adds=texts[0], max_frames=100, sid="too long", **sp),
            dict(prompt_embeds=prompts[1], text_adds=texts[1], max_frames=5, sid="b", **sp),
            dict(prompt_embeds=prompts[2], text_adds=texts[2], max_frames=5, sid="c", **sp),
            dict(prompt_embeds=prompts[3], text_adds=texts[3], max_frames=5, sid="d", **sp),
            dict(prompt_embeds=prompts[0], text_adds=texts[0], max_frames=5, sid="no row left", **sp)]
    res = fe.add_streams(reqs)
    assert res[0] == "a" and res[2:5] == ["b", "c", "d"]
    assert isinstance(res[1], ValueError) and isinstance(res[5], NoCapacity)
    assert sorted(fe.streams) == ["a", "b", "c", "d"]
    rows = {sid: fe.streams[sid].row for sid in "abcd"}
    out = fe.run_frames(5)
    for i, sid in enumerate("abcd"):
        assert out[:, rows[sid]].tolist() == want[i], sid
    for sid in "abcd":
        fe.remove_stream(sid)
    assert len(fe.talker.scheduler.block_manager.free_block_ids) == free0
    # two prompts that do not fit one runner step together
    res = fe.add_streams([dict(prompt_embeds=prompts[i], text_adds=texts[i], max_frames=5, sid=f"long{i}", **sp) for i in (4, 5)])
    assert res == ["long4", "long5"]
    rows = {sid: fe.streams[sid].row for sid in res}
    out = fe.run_frames(5)
    assert out[:, rows["long4"]].tolist() == want[4] and out[:, rows["long5"]].tolist() == want[5]
    fe.exit()


def test_a_row_taken_over_while_its_last_talker_step_runs_starts_at_frame_zero(tmp_path):
    """1.7B shapes (a talker step takes a millisecond): stream A ends with its frame budget and, before the talker step
    that closed its last frame has finished on the GPU, request C takes over A's row (add_stream(wait=False), as a
    serving loop does from the engine's hook; frames go on beside C's prefill).  Handing the row back and switching C
    on both write the row's decode metadata behind the queued frames: the row must stay switched off in between (its
    frame index still 0 at C's first frame, i.e. C reads its own text rows from the first), whatever the moment the
    device sees those writes.  C's and the bystander B's ids must equal their runs alone."""
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    from nano_qwen3tts_vllm.utils import synthetic
    cfg = synthetic.make_config("1.7B")
    sd = synthetic.make_state_dict(cfg, seed=31, norm_jitter=0.1, embed_std=0.05, head_sharp=1.0, head_std=0.05)
    tw, pw = synthetic.split_state_dict(sd)
    del sd
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    H = cfg["talker_config"]["hidden_size"]
    gen = torch.Generator().manual_seed(9)
    prompts = [torch.randn(L, H, generator=gen) * 0.5 for L in (7, 12, 10)]
    texts = [torch.randn(8, H, generator=gen) * 0.6 for _ in prompts]        # a frame's text row matters to its ids
    kw = dict(max_streams=2, max_text_frames=8, talker_weights=tw, predictor_weights=pw, num_kvcache_blocks=4,
              max_num_batched_tokens=1024, codec_eos=None)
    sp = dict(talker_sampling=greedy(1), predictor_sampling=greedy(17, 0.9))

    def solo(i, n):
        fe = FrameEngine(str(tmp_path), **kw)
        fe.add_stream(prompts[i], texts[i], max_frames=n, **sp)
        out = fe.run_frames(n)[:, 0].tolist()
        fe.exit()
        return out

    want_b, want_c = solo(1, 7), solo(2, 4)
    for attempt in range(2):
        fe = FrameEngine(str(tmp_path), **kw)
        fe.join_prefills = False                              # frames go on while C's prompt is prefilled beside them
        a = fe.add_stream(prompts[0], texts[0], max_frames=2, **sp)
        b = fe.add_stream(prompts[1], texts[1], max_frames=7, **sp)
        got = {a: [], b: []}
        for _ in range(2):
            for sid, fr in fe.step().items():
                got[sid].append(fr)
        assert a not in fe.streams                            # A has ended; the talker step of its last frame is in flight
        c = fe.add_stream(prompts[2], texts[2], max_frames=4, wait=False, **sp)
        assert fe.streams[c].row == 0
        got[c] = []
        while fe.streams:
            for sid, fr in fe.step().items():
                got[sid].append(fr)
        fe.exit()
        assert got[b] == want_b, attempt
        assert got[c] == want_c, attempt
