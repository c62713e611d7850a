#!/usr/bin/env python3
"""Headline benchmark: codec tokens/s and RTF at 8 concurrent streams per GPU,
Qwen3-TTS-12Hz-1.7B shapes (BASELINE.json configs[2]), synthetic weights and prompts.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python bench.py --gpus N ...                      (starts its N ranks itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one 16-code frame for every stream of this rank's replica (15-step predictor burst +
frame glue as one hipGraph, talker decode as another).  Ranks are independent replicas (no
collective on the data path; the barrier around the timed region is host-side).  `value` is the
steady state with all streams admitted (prefill before the timed region, reported as first-chunk
latency); `arrivals` in the same line is the serving form of the same configuration: streams
arriving 50 ms apart, 125 frames each, prefill inside the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

FRAME_SECONDS = 0.08          # 12.5 Hz codec frames
HBM_PEAK = 8.0e12             # B/s, MI355X HBM3E (spec)


def make_weights(cfg, device, seed=1234):
    """N(0, 0.02^2) bf16 weights of the released 1.7B architecture, generated on the device."""
    from nano_qwen3tts_vllm.utils import synthetic
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for k, shp in synthetic.checkpoint_shapes(cfg).items():
        if k.endswith("norm.weight") or k.endswith("layernorm.weight"):
            sd[k] = torch.ones(shp, dtype=torch.bfloat16, device=device)
        else:
            std = 0.05 if "codec_embedding" in k else 0.02
            sd[k] = (torch.randn(shp, generator=g, device=device, dtype=torch.float32) * std).to(torch.bfloat16)
    return synthetic.split_state_dict(sd)


def algorithmic_bytes_per_frame(cfg, B, mean_ctx, weights="bf16", folded=False):
    """Bytes one frame must move: every talker weight once, the predictor's per-step weights
    15 times, plus the KV rows attention reads (bf16).  With fp8 weights the layers' projection
    matrices count one byte per element (+ 4 bytes of scale per row, neglected).  folded: the
    predictor's 14 decode steps read their projected input rows and layer 0's qkv rows from tables
    built at load, so the projection matrix and layer 0's q/k/v matrices are not streamed in
    those steps (the B table rows read instead are counted)."""
    from nano_qwen3tts_vllm.utils import synthetic
    shapes = synthetic.checkpoint_shapes(cfg)
    numel = lambda k: int(torch.tensor(shapes[k]).prod())
    t = cfg["talker_config"]
    p = t["code_predictor_config"]
    talker = sum(numel(k) for k in shapes if k.startswith("talker.model.layers.")) + numel("talker.model.norm.weight") \
        + numel("talker.codec_head.weight")
    pred_layers = sum(numel(k) for k in shapes if k.startswith("talker.code_predictor.model.layers.")) \
        + numel("talker.code_predictor.model.norm.weight")
    pred_step = pred_layers + numel("talker.code_predictor.lm_head.0.weight")
    if "talker.code_predictor.small_to_mtp_projection.weight" in shapes:
        pred_step += numel("talker.code_predictor.small_to_mtp_projection.weight")
    kv_t = 2 * t["num_hidden_layers"] * t["num_key_value_heads"] * t["head_dim"] * 2
    kv_p = 2 * p["num_hidden_layers"] * p["num_key_value_heads"] * p["head_dim"] * 2
    n_gen = p["num_code_groups"] - 1
    pred_ctx = sum(range(2, 2 + n_gen))          # 2..16 cached positions read over the burst
    if weights == "fp8":
        proj = lambda pre: sum(numel(k) for k in shapes if k.startswith(pre) and "_proj.weight" in k)
        tproj, pproj = proj("talker.model.layers."), proj("talker.code_predictor.model.layers.")
        return (2 * talker - tproj) + n_gen * (2 * pred_step - pproj) + B * mean_ctx * kv_t + B * pred_ctx * kv_p
    saved = 0
    if folded:
        l0 = "talker.code_predictor.model.layers.0."
        skipped = sum(numel(l0 + f"self_attn.{n}_proj.weight") for n in "qkv") + numel(l0 + "input_layernorm.weight")
        qkvw = (p["num_attention_heads"] + 2 * p["num_key_value_heads"]) * p["head_dim"]
        if "talker.code_predictor.small_to_mtp_projection.weight" in shapes:
            skipped += numel("talker.code_predictor.small_to_mtp_projection.weight")
        saved = (n_gen - 1) * (2 * skipped - 2 * B * (qkvw + p["hidden_size"]))
    return 2 * talker + 2 * n_gen * pred_step + B * mean_ctx * kv_t + B * pred_ctx * kv_p - saved


# The launches of a frame that the event hook can time in place (qtts_model_set_layer_events): family -> (model,
# which, what it is).  Algorithmic bytes: the launch's weights (+ rows in and out), or the K/V rows attention reads.
LAUNCH_FAMILIES = [
    ("talker norm + gate/up + SwiGLU", "talker", 3), ("talker down_proj + residual", "talker", 4),
    ("talker norm + qkv", "talker", 1), ("talker o_proj + residual", "talker", 2),
    ("talker q/k-norm + RoPE + store + attention", "talker", 5), ("talker final norm + codec head", "talker", 6),
    ("predictor norm + gate/up + SwiGLU", "predictor", 3), ("predictor down_proj + residual", "predictor", 4),
    ("predictor norm + qkv", "predictor", 1), ("predictor o_proj + residual", "predictor", 2),
    ("predictor q/k-norm + RoPE + store + attention", "predictor", 5), ("predictor final norm + head", "predictor", 6),
]


def launch_bytes_and_counts(fe, B, mean_ctx, folded):
    """Per launch family: (algorithmic bytes of one launch, launches per frame)."""
    fp8 = fe.tr.decoder.weight_dtype == "fp8"
    wb = 1 if fp8 else 2
    out = {}
    for tag, runner in (("talker", fe.tr), ("predictor", fe.pr)):
        sh = runner.shape
        H, I, L = sh.hidden_size, sh.intermediate_size, sh.num_layers
        A, KVW = sh.num_heads * sh.head_dim, sh.num_kv_heads * sh.head_dim
        steps = 1 if tag == "talker" else fe.pr.n_gen
        rows = B                                        # (the burst's first step has 2 B rows: counted as B)
        ctx = mean_ctx if tag == "talker" else (2 + fe.pr.n_gen) / 2.0
        n = L * steps
        out[f"{tag} norm + gate/up + SwiGLU"] = (2 * I * H * wb + rows * (H + I) * 2, n)
        out[f"{tag} down_proj + residual"] = (I * H * wb + rows * (I + 2 * H) * 2, n)
        out[f"{tag} norm + qkv"] = ((A + 2 * KVW) * H * wb + rows * (H + A + 2 * KVW) * 2,
                                    n - ((fe.pr.n_gen - 1) if (tag == "predictor" and folded) else 0))
        out[f"{tag} o_proj + residual"] = (A * H * wb + rows * (A + 2 * H) * 2, n)
        out[f"{tag} q/k-norm + RoPE + store + attention"] = (int(rows * ctx * 2 * KVW * 2), n)
        out[f"{tag} final norm + " + ("codec head" if tag == "talker" else "head")] = (runner.vocab * H * 2 + rows * (H + runner.vocab) * 2, steps)
    return out


def time_launches_in_frames(fe, frames=6):
    """Device time of every launch family of LAUNCH_FAMILIES where it runs: while whole frames are generated eagerly on
    the engine's stream, launch `which` of every layer of one model carries two HIP events as its start / stop events
    (qtts_model_set_layer_events -> hipExtLaunchKernelGGL), whose elapsed time is the kernel's own execution time in
    its real place in the frame, weights cold.  One family at a time.  Returns {family: (mean s, median s, samples)}.
    (For the predictor the events keep the last burst step's five launches of each frame.)"""
    from nano_qwen3tts_vllm import hip
    import ctypes
    tr = fe.tr
    bs = fe._bucket()
    out = {}
    generated = 0
    with tr.on_stream():
        fe._setup_predictor_meta(bs)
        for name, model, which in LAUNCH_FAMILIES:
            dec = (fe.tr if model == "talker" else fe.pr).decoder
            nl = dec.shape.num_layers
            n_ev = 1 if which == 6 else nl
            begin = [torch.cuda.Event(enable_timing=True) for _ in range(nl)]
            end = [torch.cuda.Event(enable_timing=True) for _ in range(nl)]
            for e in begin + end:                       # an event's handle exists once it has been recorded
                e.record()
            arr = ctypes.c_void_p * nl
            hip.check(hip.lib.qtts_model_set_layer_events(dec._h, which, arr(*[e.cuda_event for e in begin]),
                                                          arr(*[e.cuda_event for e in end])), "set_layer_events")
            samples = []
            try:
                for f in range(frames):
                    fe._codes_part(bs)                   # eager launches (the graphs' launches cannot carry events)
                    fe._talker_part(bs, fe._max_ctx_bucket(extra_frames=generated + 1))
                    generated += 1
                    tr.stream.synchronize()
                    if f >= 1:
                        samples += [b.elapsed_time(e) * 1e-3 for b, e in list(zip(begin, end))[:n_ev]]
            finally:
                hip.check(hip.lib.qtts_model_set_layer_events(dec._h, 0, None, None), "set_layer_events")
            out[name] = (statistics.fmean(samples), statistics.median(samples), len(samples))
    for s in fe.streams.values():                       # these frames advanced the streams' contexts like any others
        s.frames.extend([[0] * 16] * generated)
    return out, generated


def host_cost_per_frame(fe, frames=40):
    """What one frame costs the host through the serving call: wall time of FrameEngine.step, the part of it the host
    is busy (enqueueing the frame's two graphs, reading the codes, bookkeeping - not waiting), and the process's CPU
    time per frame with the admission hook polled (step(poll=...): the replica processes' form, sleeping between
    polls)."""
    res = {}
    for label, poll in (("step", None), ("step_polling", lambda: None)):
        fe.host_busy_s, fe.host_frames = 0.0, 0
        fe.step(poll=poll)
        fe.host_busy_s, fe.host_frames = 0.0, 0
        c0, t0 = time.process_time(), time.perf_counter()
        for _ in range(frames):
            fe.step(poll=poll)
        wall, cpu = time.perf_counter() - t0, time.process_time() - c0
        res[label] = {"ms_per_frame": wall / frames * 1e3, "host_enqueue_ms_per_frame": fe.host_busy_s / max(fe.host_frames, 1) * 1e3,
                      "host_cpu_ms_per_frame": cpu / frames * 1e3}
    return res


def cpu_baseline(cfg, tw, pw, B, frames, prompt_len):
    """The CPU oracle (oracle/qtts_oracle.py, torch eager bf16) on a bounded sample of the same
    workload, timed on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import qtts_oracle as O
    from helpers import oracle_cfgs
    tc, pc = oracle_cfgs(cfg)
    twc = {k: v.cpu() for k, v in tw.items()}
    pwc = {k: v.cpu() for k, v in pw.items()}
    g = torch.Generator().manual_seed(7)
    H = tc.hidden_size
    prompts = [torch.randn(prompt_len, H, generator=g) * 0.5 for _ in range(B)]
    text = torch.randn(B, frames, H, generator=g) * 0.1
    t0 = time.perf_counter()
    O.frame_loop(twc, tc, pwc, pc, prompts, text, frames, num_blocks=4 * B)
    dt = time.perf_counter() - t0
    return dict(value=B * frames * 16 / dt, unit="codec tokens/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle frame loop (torch eager bf16 CPU): {B} streams, prompt {prompt_len} tokens (prefill included), "
                       f"{frames} frames, {dt:.1f} s")


PROFILE_ROUNDS = ("r03", "r02", "r01")


def pmc_traffic(which="talker"):
    """HBM bytes per launch of a decode gate/up launch (which: talker / predictor) as stored in profiles/ by the
    rocprofv3 PMC passes over tools/microbench.py dominant (FETCH_SIZE and WRITE_SIZE collected separately,
    FETCH_SIZE doubled as gfx950 requires) - a stored figure: hardware counters cannot be read from inside this
    process.  (None, None) when absent."""
    for rnd in PROFILE_ROUNDS:
        name = f"{rnd}_pmc_traffic.json"
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if which == "predictor":
                d = d["predictor_gate_up"]
            return d["traffic_bytes_per_launch"], f"profiles/{name}"
        except Exception:
            continue
    return None, None


def profiler_duration(which="talker"):
    """(average us, file) of a decode gate/up launch (which: talker / predictor) in the committed rocprofv3 kernel
    trace of this same command (profiles/rNN_bench_summary_*.txt, the first grid line of that kernel's section) - a
    stored figure, shown beside the live one: the profiler's begin->end timestamps exclude what the command processor
    spends around a kernel, which the events include."""
    import glob
    import re
    section = {"talker": "linear2_kernel<8, 1, true, 1,", "predictor": "linear2_kernel<8, 1, false, 1,"}[which]
    for rnd in PROFILE_ROUNDS:
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{rnd}_bench_summary_*.txt")), reverse=True):
            try:
                text = open(path).read()
                at = text.find(section + " by grid")
                m = re.search(r"grid \(\d+, \d+\): calls\s+\d+ avg\s+([0-9.]+) us", text[at:]) if at >= 0 else None
                if m:
                    return float(m.group(1)), "profiles/" + os.path.basename(path)
            except Exception:
                continue
    return None, None


def dist_env():
    """(rank, world, local_rank) as torch.distributed.run exports them; a plain `python bench.py` is rank 0 of 1."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def max_over_ranks(elapsed: float, world: int) -> float:
    """Replicas are independent: the job's time is the slowest rank's (host-side reduction)."""
    if world <= 1:
        return elapsed
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_tokens(streams_per_gpu: int, steps: int, world: int) -> int:
    """16 codes per frame per stream, every rank runs `streams_per_gpu` streams for `steps` frames (weak scaling)."""
    return 16 * streams_per_gpu * steps * world


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_replicas(n: int) -> int:
    """`python bench.py --gpus N` without a launcher around it: start N copies of this script, one per GPU (rank i
    on device i), before this process has touched a GPU, and wait for them.  Rank 0's JSON line goes to this
    process's stdout.  A failed or killed child ends the others and the exit code is non-zero."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), QTTS_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for o in pending:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def percentile(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(round(q * (len(xs) - 1))))] if xs else None


ADMISSION_MODES = {
    "serial": "prefill between two frames",
    "beside": "prefill on a second stream beside the running frames, joined once finished",
    "join": "prefill started on a second stream as soon as the request arrives, beside the frame in flight; the next frame waits for it",
}


def run_arrivals(fe, device, H, streams, frames, gap_s, lens, samplers, seed, admission="join"):
    """BASELINE.json configs[2] as a serving workload: `streams` requests arrive `gap_s` apart (prompt lengths cycling
    through `lens`), each wants `frames` frames; admission and prefill happen inside the timed region, as in a
    server's loop.  admission "serial": what has arrived is admitted between two frames, its packed prefill runs there.
    "join" / "beside": arrivals are also looked for while the host waits for a frame's codes, and their packed prefill
    starts at once on a second stream (FrameEngine.add_streams(wait=False)); the newcomers join with the next frame
    ("join": it waits for the rest of the prefill) or with the first frame after the prefill has finished ("beside").
    Returns throughput and latency statistics; the clock is the host's (a frame is counted when its 16 codes are on the
    host)."""
    from nano_qwen3tts_vllm.engine.frame_engine import NoCapacity
    g = torch.Generator().manual_seed(seed)
    reqs = []
    for i in range(streams):
        L = lens[i % len(lens)]
        reqs.append((L, (torch.randn(L, H, generator=g) * 0.5).to(device), (torch.randn(frames + 1, H, generator=g) * 0.1).to(device)))
    torch.cuda.synchronize()
    arrival = [i * gap_s for i in range(streams)]
    first, last, gaps, done_frames, sid_of = {}, {}, [], {}, {}
    state = {"nxt": 0}
    detached = admission != "serial"
    fe.join_prefills = admission == "join"
    if os.environ.get("QTTS_LAUNCH_MARGIN_US"):      # tuning: how long before the talker step's end a frame is committed
        fe.LAUNCH_MARGIN_S = float(os.environ["QTTS_LAUNCH_MARGIN_US"]) * 1e-6
    t0 = time.perf_counter()
    now = lambda: time.perf_counter() - t0

    def admit():
        """Everything that has arrived by now goes to the engine in one call (one packed prefill), as a serving loop does."""
        nxt = state["nxt"]
        due = []
        while nxt + len(due) < streams and now() >= arrival[nxt + len(due)] and len(fe.streams) + len(due) < fe.max_streams:
            due.append(nxt + len(due))
        if not due:
            return
        res = fe.add_streams([dict(prompt_embeds=reqs[i][1], text_adds=reqs[i][2], max_frames=frames, talker_sampling=samplers[0],
                                   predictor_sampling=samplers[1], sid=f"arr-{i}") for i in due], wait=not detached)
        for i, r in zip(due, res):
            if isinstance(r, Exception):
                if isinstance(r, NoCapacity):
                    break                              # admitted as soon as a running stream ends
                raise r
            sid_of[r] = i
            done_frames[i] = 0
            state["nxt"] += 1

    while len(done_frames) < streams or fe.streams:
        admit()
        if not fe.streams:
            if state["nxt"] >= streams:
                break
            time.sleep(max(0.0, min(arrival[state["nxt"]] - now(), 0.001)))
            continue
        res = fe.step(poll=admit if detached else None)
        t = now()
        for sid in res:
            i = sid_of[sid]
            if done_frames[i] == 0:
                first[i] = t - arrival[i]
            else:
                gaps.append(t - last[i])
            last[i] = t
            done_frames[i] += 1
    elapsed = now()
    total = sum(done_frames.values())
    assert total == streams * frames, (done_frames, frames)
    ms = lambda v: None if v is None else v * 1e3
    return {"streams": streams, "arrival_gap_ms": gap_s * 1e3, "frames_per_stream": frames, "prompt_tokens": [r[0] for r in reqs],
            "elapsed_s": elapsed, "codec_tokens_per_s": 16 * total / elapsed,
            "rtf_per_stream_mean": statistics.fmean((last[i] - arrival[i]) / (frames * FRAME_SECONDS) for i in range(streams)),
            "first_chunk_ms_p50": ms(percentile(list(first.values()), 0.5)), "first_chunk_ms_p99": ms(percentile(list(first.values()), 0.99)),
            "inter_chunk_ms_p50": ms(percentile(gaps, 0.5)), "inter_chunk_ms_p99": ms(percentile(gaps, 0.99)),
            "admission": ADMISSION_MODES[admission],
            "first_chunk_ms": [ms(first[i]) for i in range(streams)],
            "note": "prefill and admission inside the timed region; first chunk = arrival -> the stream's first 16-code frame on the host"}


ROOFLINE_FRAMES = 6 * len(LAUNCH_FAMILIES) + 2 * 42      # frames generated after the timed region (launch timing, host cost)
ARRIVAL_FRAMES = 125          # 10 s of audio per stream
ARRIVAL_GAP_S = 0.050


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--size", default="1.7B")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: the layers' GEMM weights as e4m3 with power-of-two row scales (an extension; not the contract line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-arrivals", action="store_true", help="skip the staggered-arrival serving workload")
    ap.add_argument("--cpu-frames", type=int, default=40)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one (nothing has touched a GPU yet)
        sys.exit(launch_replicas(args.gpus))

    os.environ["QTTS_WEIGHT_DTYPE"] = args.weights
    rank, world, local = dist_env()
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    # Rehearsal on a box with fewer GPUs than ranks (QTTS_BENCH_REHEARSAL=1): every rank uses GPU 0.  Never used for
    # reported numbers.
    rehearsal = os.environ.get("QTTS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        # Replicas only: no collective touches the data path, and none touches a GPU - the barrier around the timed
        # region and the max over ranks are host-side (gloo).
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")

    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    from nano_qwen3tts_vllm.sampling_params import SamplingParams
    from nano_qwen3tts_vllm.utils import synthetic

    cfg = synthetic.make_config(args.size)
    tw, pw = make_weights(cfg, device)
    mdir = tempfile.mkdtemp(prefix="qtts_bench_")
    json.dump(cfg, open(os.path.join(mdir, "config.json"), "w"))
    B = args.streams
    n_frames = args.steps + args.warmup + 1 + ROOFLINE_FRAMES
    H = cfg["talker_config"]["hidden_size"]
    fe = FrameEngine(mdir, max_streams=B, max_text_frames=max(n_frames, ARRIVAL_FRAMES) + 2, talker_weights=tw, predictor_weights=pw,
                     num_kvcache_blocks=4 * B, max_num_batched_tokens=4096, codec_eos=None)
    lens = [16, 48, 128, 200]
    # a stream's prompt + frames must fit the model's 1024 positions: very long runs get shorter prompts
    room = 1024 - (n_frames + 4)
    if room < 8:
        raise SystemExit(f"--steps + --warmup = {args.steps + args.warmup} frames do not fit 1024 positions per stream; use at most {1000 - ROOFLINE_FRAMES}")
    steady_lens = [min(L, room) for L in lens]
    # as a server does at start-up (Qwen3TTSInterface.start_zmq_tasks): every frame graph this run can meet is captured
    # before the first request, so first-chunk latencies below are replays, not captures
    top_ctx = max(max(steady_lens) + n_frames, max(lens) + ARRIVAL_FRAMES) + 3
    fe.capture_all(max_context=min(1024, ((top_ctx + 127) // 128) * 128))
    greedy_t = SamplingParams(temperature=1.0, max_tokens=1, do_sample=False)
    greedy_p = SamplingParams(temperature=0.9, max_tokens=17, do_sample=False)
    g = torch.Generator().manual_seed(1000 + rank)
    # one warm-up request per prompt-kernel family (the reference's server fires warm-up requests too, E/server.py:276)
    for L in (16, 200):
        sid = fe.add_stream((torch.randn(L, H, generator=g) * 0.5).to(device), (torch.randn(4, H, generator=g) * 0.1).to(device),
                            max_frames=4, talker_sampling=greedy_t, predictor_sampling=greedy_p)
        fe.run_frames(1)
        fe.remove_stream(sid)

    # ---- the serving workload of configs[2]: staggered arrivals, prefill inside the timed region (rank-local, reported
    # by rank 0 at N = 1 beside the contract number)
    arrivals = None
    if not args.no_arrivals and world == 1:
        admission = os.environ.get("QTTS_ADMISSION", "join")
        assert admission in ADMISSION_MODES, admission
        run_arrivals(fe, device, H, 2, 3, 0.0, lens, (greedy_t, greedy_p), 1999, admission)      # (first use of the second stream)
        arrivals = run_arrivals(fe, device, H, B, ARRIVAL_FRAMES, ARRIVAL_GAP_S, lens, (greedy_t, greedy_p), 2000 + rank, admission)
        assert not fe.streams
        # Eight arrivals are few: where an arrival falls inside the frame in flight decides its first chunk, and with a
        # fixed gap the eight phases are one draw.  Two more runs with gaps that shift the phase (50.9 and 51.7 ms
        # against a 2.5 ms frame) are pooled with the first for a steadier percentile.
        pooled = list(arrivals["first_chunk_ms"])
        for k, gap in enumerate((0.0509, 0.0517)):
            more = run_arrivals(fe, device, H, B, ARRIVAL_FRAMES, gap, lens, (greedy_t, greedy_p), 2100 + 10 * k + rank, admission)
            pooled += more["first_chunk_ms"]
            assert not fe.streams
        arrivals["first_chunk_ms_pooled"] = {"p50": percentile(pooled, 0.5), "p99": percentile(pooled, 0.99), "arrivals": len(pooled),
                                             "gaps_ms": [ARRIVAL_GAP_S * 1e3, 50.9, 51.7]}
        # the same requests all at once (the reference's batch_async_example.py fires its streams together): admission packs
        # the prompts into one prefill
        together = run_arrivals(fe, device, H, B, ARRIVAL_FRAMES, 0.0, lens, (greedy_t, greedy_p), 3000 + rank, admission)
        arrivals["all_at_once"] = {k: together[k] for k in ("first_chunk_ms_p50", "first_chunk_ms_p99", "inter_chunk_ms_p50",
                                                            "inter_chunk_ms_p99", "codec_tokens_per_s", "elapsed_s")}
        assert not fe.streams

    # ---- the contract number: B streams admitted, steady-state frames
    first_frame_ms = []
    prompt_lens = []
    for i in range(B):
        L = steady_lens[i % len(steady_lens)]
        prompt_lens.append(L)
        prompt = (torch.randn(L, H, generator=g) * 0.5).to(device)
        text = (torch.randn(n_frames + 1, H, generator=g) * 0.1).to(device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fe.add_stream(prompt, text, max_frames=n_frames + 2, talker_sampling=greedy_t, predictor_sampling=greedy_p)
        torch.cuda.synchronize()
        first_frame_ms.append((time.perf_counter() - t0) * 1e3)      # prefill -> first talker id + hidden
    # one frame with everything admitted, through the serving call (FrameEngine.step returns when the frame's 16 codes
    # are on the host; the talker step that starts the next frame is still running then): the rest of a stream's
    # first 16-code frame
    t0 = time.perf_counter()
    first = fe.step()
    frame0_ms = (time.perf_counter() - t0) * 1e3
    assert len(first) == B and all(len(c) == 16 for c in first.values())
    torch.cuda.synchronize()

    fe.run_frames(args.warmup)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    fe.run_frames(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, world)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        tokens = whole_job_tokens(B, args.steps, world)
        value = tokens / elapsed
        mean_ctx = sum(prompt_lens) / B + 1 + args.warmup + 1 + args.steps / 2
        folded = bool(getattr(fe.pr, "fold", False))
        bytes_frame = algorithmic_bytes_per_frame(cfg, B, mean_ctx, args.weights, folded)
        timed, _ = time_launches_in_frames(fe, 6)
        per = launch_bytes_and_counts(fe, B, mean_ctx, folded)
        by_launch = []
        for name, _model, _which in LAUNCH_FAMILIES:
            mean_s, median_s, n = timed[name]
            nbytes, count = per[name]
            if mean_s < 2e-7:              # (steps of more than 16 rows run the prompt kernels, which carry no events)
                continue
            by_launch.append({"launch": name, "per_frame": count, "us": mean_s * 1e6, "us_median": median_s * 1e6,
                              "us_per_frame": mean_s * 1e6 * count, "bytes_per_launch": nbytes,
                              "achieved_GBps": nbytes / mean_s / 1e9, "frac": nbytes / mean_s / HBM_PEAK, "launches_timed": n})
        by_launch.sort(key=lambda r: -r["us_per_frame"])
        if not by_launch:                                    # nothing could be timed in place: the frame as a whole
            by_launch = [{"launch": "whole frame", "per_frame": 1, "us": ms_per_step * 1e3, "us_median": ms_per_step * 1e3,
                          "us_per_frame": ms_per_step * 1e3, "bytes_per_launch": bytes_frame,
                          "achieved_GBps": bytes_frame / (ms_per_step * 1e-3) / 1e9, "frac": bytes_frame / (ms_per_step * 1e-3) / HBM_PEAK,
                          "launches_timed": args.steps}]
        # An event pair around one kernel also times what the command processor does around it, which consecutive nodes
        # of the frame graphs do not pay: the families' event times add up to more than the frame takes.  The mean
        # excess per launch (the frame's few untimed launches - picks, glue - counted as free, so this is a lower
        # bound) is reported, and every family's time with it taken off as an estimate of its share of the real frame.
        ev_sum = sum(r["us_per_frame"] for r in by_launch)
        ev_n = sum(r["per_frame"] for r in by_launch)
        ev_over = max(0.0, (ev_sum - ms_per_step * 1e3) / ev_n) if len(by_launch) > 1 else 0.0
        for r in by_launch:
            r["us_in_frame_estimate"] = r["us"] - ev_over
        top = by_launch[0]                                   # the family with the largest summed time per frame
        talker_gu = next((r for r in by_launch if r["launch"] == "talker norm + gate/up + SwiGLU"), top)
        host = host_cost_per_frame(fe, 40)
        stored = args.weights == "bf16" and args.size == "1.7B" and B == 8       # what the stored figures were taken on
        traffic, traffic_src = pmc_traffic() if stored else (None, None)
        prof_us, prof_src = profiler_duration() if stored else (None, None)
        # the same two stored figures for the top family when it is the predictor's gate/up launch
        top_model = "predictor" if top["launch"].startswith("predictor norm + gate/up") else "talker" if top is talker_gu else None
        top_traffic, top_traffic_src = pmc_traffic(top_model) if stored and top_model else (None, None)
        top_prof_us, top_prof_src = profiler_duration(top_model) if stored and top_model else (None, None)
        out = {
            "metric": f"codec tokens/sec at {B} concurrent streams per GPU (RTF alongside), Qwen3-TTS-{args.size}",
            "value": value, "unit": "codec tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.weights == "bf16" else "bf16 (weights stored as fp8 e4m3, converted in registers)",
            "data": "synthetic",
            "config": {"workload": f"`value`: Qwen3-TTS-12Hz-{args.size} shapes, {B} concurrent streams per GPU all admitted, greedy, "
                                   f"steady-state frames (one codes graph + one talker graph per frame), prompts {prompt_lens} "
                                   f"tokens prefilled before the timed region, random-init weights"
                                   + (", fp8 weights" if args.weights == "fp8" else "")
                                   + (" [REHEARSAL: all ranks on one GPU]" if rehearsal else "")
                                   + "; `arrivals`: the same engine under staggered arrivals with prefill inside the timed region",
                       "streams_per_gpu": B, "replicas": world},
            "rtf": ms_per_step * 1e-3 / FRAME_SECONDS,
            "host_enqueue_ms_per_frame": host["step"]["host_enqueue_ms_per_frame"],
            "first_chunk_ms_p50": statistics.median(first_frame_ms) + frame0_ms,
            "prefill_ms": first_frame_ms,
            "frame_roofline": {"bound": "hbm", "bytes_per_frame": bytes_frame,
                               "achieved": bytes_frame / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                               "unit": "GB/s", "frac": bytes_frame / (ms_per_step * 1e-3) / HBM_PEAK},
            "roofline": {"bound": "hbm", "kernel": top["launch"] + f" ({top['per_frame']} launches per frame: the family with the largest summed time)",
                         "achieved": top["achieved_GBps"], "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": top["frac"], "traffic": top_traffic, "traffic_source": top_traffic_src,
                         "us_per_launch_profiler": top_prof_us, "profiler_source": top_prof_src,
                         "bytes_per_launch": top["bytes_per_launch"], "us_per_launch": top["us"], "us_per_launch_median": top["us_median"],
                         "launches_timed": top["launches_timed"],
                         "how": "HIP start/stop events bound to that launch of every layer (hipExtLaunchKernelGGL on the engine's "
                                "stream) while frames are generated eagerly, right after the timed region (an event pair around "
                                "one kernel includes the command processor's work around it, about 2 us that consecutive nodes "
                                "of the frame graphs do not pay: us_per_launch_profiler is the kernel's own begin->end time, "
                                "event_overhead_us_per_launch the mean excess over the graph-replayed frame, us_in_frame_estimate a "
                                "family's time with it taken off); "
                                "`by_launch`: every launch family of the frame the same way, sorted by time per frame",
                         "largest_launch": {"kernel": "linear2_kernel<8,1,true,1> (talker RMSNorm + gate/up GEMM + SwiGLU launch: the most bytes per launch)",
                                            "achieved": talker_gu["achieved_GBps"], "frac": talker_gu["frac"], "us_per_launch": talker_gu["us"],
                                            "bytes_per_launch": talker_gu["bytes_per_launch"], "traffic": traffic, "traffic_source": traffic_src,
                                            "us_per_launch_profiler": prof_us, "profiler_source": prof_src},
                         "events_sum_us_per_frame": ev_sum, "frame_us": ms_per_step * 1e3, "event_overhead_us_per_launch": ev_over,
                         "by_launch": by_launch},
            "host": {"host_enqueue_ms_per_frame": host["step"]["host_enqueue_ms_per_frame"],
                     "serving_ms_per_frame": host["step"]["ms_per_frame"],
                     "polling": host["step_polling"],
                     "launch_margin_ms": fe.launch_margin_s() * 1e3,
                     "note": "FrameEngine.step(): host time per frame not spent waiting (two graph launches, codes read-back, "
                             "bookkeeping); polling: the replica processes' form, the admission hook polled while the frame runs "
                             "(host_cpu_ms_per_frame = process CPU time, sleeps between polls excluded); launch_margin_ms: how long before "
                             "the expected end of the talker step the polling form commits the next frame (follows the measured "
                             "enqueue time)"},
        }
        if arrivals is not None:
            out["arrivals"] = arrivals
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed beside the 1-GPU line only
            out["cpu_baseline"] = cpu_baseline(cfg, tw, pw, B, args.cpu_frames, 16)
        print(json.dumps(out), flush=True)
    fe.exit()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
