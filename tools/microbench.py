"""GPU microbenchmarks used while tuning (not part of the product path).
  python tools/microbench.py launch     # per-launch cost of trivial kernels, eager vs graph
  python tools/microbench.py linear     # skinny-GEMM sweep at the model's shapes
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: F401  (the -DQTTS_TUNING build of the library)
from nano_qwen3tts_vllm import hip

dev = torch.device("cuda", 0)


def timed(fn, n_inner, reps=5):
    s = torch.cuda.Stream()
    best = 1e9
    with torch.cuda.stream(s):
        fn()
        s.synchronize()
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n_inner)
    return best


def launch():
    st = {k: torch.zeros(64, dtype=torch.int32, device=dev) for k in "abc"}
    pos = torch.zeros(64, dtype=torch.int64, device=dev)
    tab = torch.zeros(64 * 4, dtype=torch.int32, device=dev)
    x = torch.randn(8, 2048, device=dev).bfloat16()
    w = torch.ones(2048, device=dev).bfloat16()
    y = torch.empty_like(x)

    def adv():
        hip.check(hip.lib.qtts_advance_decode(hip.ptr(pos), hip.ptr(st["a"]), hip.ptr(st["b"]), hip.ptr(tab), 4, None, None,
                                              1024, 8, hip.stream()))

    def rms():
        hip.check(hip.lib.qtts_rmsnorm(hip.ptr(x), 2048, hip.ptr(w), 1e-6, hip.ptr(y), 2048, 8, 2048, hip.stream()))

    for name, k in (("advance(1 WG)", adv), ("rmsnorm(8 WG)", rms)):
        n = 500
        eager = timed(lambda: [k() for _ in range(n)], n)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            k()
            g = hip.Graph.capture(lambda: [k() for _ in range(n)])
        graph = timed(lambda: g.launch(), n)
        print(f"{name:16s} eager {eager:6.2f} us/launch   graph {graph:6.2f} us/launch", flush=True)


def linear():
    B = 8
    shapes = [("qkv 1.7B", 4096, 2048, True, False), ("o 1.7B", 2048, 2048, False, False),
              ("gate_up 1.7B", 12288, 2048, True, True), ("down 1.7B", 2048, 6144, False, False),
              ("qkv pred", 4096, 1024, True, False), ("o pred", 1024, 2048, False, False),
              ("gate_up pred", 6144, 1024, True, True), ("down pred", 1024, 3072, False, False),
              ("head talker", 3072, 2048, False, False), ("head pred", 2048, 1024, False, False)]
    for name, N, K, norm, silu in shapes:
        nbuf = max(2, int(600e6 // (N * K * 2)))
        ws = [torch.randn(N * K, device=dev).bfloat16() * 0.02 for _ in range(nbuf)]
        x = torch.randn(B, K, device=dev).bfloat16()
        nw = torch.ones(K, device=dev).bfloat16()
        out = torch.empty(B, N // 2 if silu else N, dtype=torch.bfloat16, device=dev)
        res = torch.randn(B, N, device=dev).bfloat16()
        line = f"{name:14s} N={N:5d} K={K:5d} {N*K*2/1e6:6.1f} MB:"
        for ksw, t in ((0, 0), (0, 2), (0, 3), (0, 4), (4, 0)):
            hip.lib.qtts_tune_linear(ksw, t)

            def run():
                for w in ws:
                    if silu:
                        hip.linear_swiglu(x, w, N // 2, norm_w=nw, out=out)
                    else:
                        hip.linear(x, w, N, norm_w=nw if norm else None, residual=None if norm else res, out=out)
            try:
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    run()
                    gr = hip.Graph.capture(run)
                us = timed(lambda: gr.launch(), nbuf)
                line += f"  ksw{ksw}/t{t}: {us:5.1f}us {N*K*2/us/1e6:5.2f}TB/s"
            except Exception as e:
                line += f"  ksw{ksw}/t{t}: n/a"
        hip.lib.qtts_tune_linear(0, 0)
        print(line, flush=True)


def linear2():
    """v1 (one tile set per workgroup) against v2 (tile loop, double buffer) per decoder shape, graph-timed."""
    B = 8
    shapes = [("qkv 1.7B", 4096, 2048, "norm"), ("o 1.7B", 2048, 2048, "res"), ("gate|up 1.7B", 12288, 2048, "silu"),
              ("down 1.7B", 2048, 6144, "res"), ("qkv pred", 4096, 1024, "norm"), ("o pred", 1024, 2048, "res"),
              ("gate|up pred", 6144, 1024, "silu"), ("down pred", 1024, 3072, "res"), ("head talker", 3072, 2048, "norm"),
              ("head pred", 2048, 1024, "norm")]
    for name, N, K, kind in shapes:
        nbuf = max(2, int(600e6 // (N * K * 2)))
        ws = [torch.randn(N * K, device=dev).bfloat16() * 0.02 for _ in range(nbuf)]
        x = torch.randn(B, 2 * K if kind == "swiglu" else K, device=dev).bfloat16()
        nw = torch.ones(K, device=dev).bfloat16()
        out = torch.empty(B, N, dtype=torch.bfloat16, device=dev)
        res = torch.randn(B, N, device=dev).bfloat16()
        line = f"{name:14s} N={N:5d} K={K:5d} {N*K*2/1e6:6.1f} MB:"

        def v1():
            for w in ws:
                if kind == "silu":
                    hip.linear_swiglu(x, w, N // 2, norm_w=nw, out=out[:, :N // 2])
                    continue
                hip.linear(x[:, :K], w, N, norm_w=nw if kind == "norm" else None, residual=res if kind != "norm" else None, out=out)

        def v2():
            for w in ws:
                if kind == "silu":
                    hip.linear_swiglu(x, w, N // 2, norm_w=nw, out=out[:, :N // 2])
                else:
                    hip.linear_fused(x, w, N, K=K, norm_w=nw if kind == "norm" else None, residual=res if kind != "norm" else None,
                                     swiglu_input=(kind == "swiglu"), out=out)

        for label, fn, tunes in (("v1", v1, [(0, 0)]), ("v2", v2, [(0, 0), (0, 2), (0, 3)])):
            for ksw, t in tunes:
                hip.lib.qtts_tune_linear(ksw, t)
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    fn()
                    gr = hip.Graph.capture(fn)
                us = timed(lambda: gr.launch(), nbuf)
                line += f"  {label}/t{t}: {us:5.1f}us {N*K*2/us/1e6:5.2f}TB/s"
        hip.lib.qtts_tune_linear(0, 0)
        print(line, flush=True)


def lat():
    """Small decode GEMM launches, back to back in one graph, with the activations read row-major (dbg0) and as if they
    lay in fragment order (dbg128: one contiguous KiB per request; values are wrong for row-major inputs, timing only;
    plain prologues of 9-16 rows only - the other forms ignore the flag).  QTTS_LAT_HOT=1: one weight buffer replayed.
    QTTS_LAT_ROWS: rows of the launch (default 8)."""
    B = int(os.environ.get("QTTS_LAT_ROWS", "8"))
    shapes = [("o pred", 1024, 2048, "res"), ("down pred", 1024, 3072, "res"), ("qkv pred", 4096, 1024, "norm"),
              ("head pred", 2048, 1024, "norm"), ("o 1.7B", 2048, 2048, "res"), ("qkv 1.7B", 4096, 2048, "norm")]
    hot = os.environ.get("QTTS_LAT_HOT") == "1"           # one weight buffer replayed: it stays in L2 / Infinity Cache
    for name, N, K, kind in shapes:
        nbuf = max(2, int(600e6 // (N * K * 2)))
        ws = [torch.randn(N * K, device=dev).bfloat16() * 0.02 for _ in range(1 if hot else nbuf)]
        if hot:
            ws = ws * nbuf
        pad = int(os.environ.get("QTTS_LAT_PAD", "0"))      # extra elements per activation / residual / output row
        x = torch.randn(max(B, 16), K + pad, device=dev).bfloat16()[:B, :K]
        nw = torch.ones(K, device=dev).bfloat16()
        out = torch.empty(B, N + pad, dtype=torch.bfloat16, device=dev)[:, :N]
        res = torch.randn(B, N + pad, device=dev).bfloat16()[:, :N]
        line = f"{name:10s} N={N:5d} K={K:5d} {N*K*2/1e6:5.1f} MB:"

        def run():
            for w in ws:
                hip.linear_fused(x, w, N, K=K, norm_w=nw if kind == "norm" else None, residual=res if kind != "norm" else None, out=out)
        for flags in (0, 128):                              # 128: activations read as one KiB per request (timing only)
            hip.lib.qtts_debug_linear(flags)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                run()
                gr = hip.Graph.capture(run)
            us = timed(lambda: gr.launch(), nbuf)
            line += f"  dbg{flags}: {us:5.2f}us"
        hip.lib.qtts_debug_linear(0)
        print(line, flush=True)


def trace():
    """Timestamps inside back-to-back GEMM launches of one graph (100 MHz device clock): when the first workgroup
    starts, when its operands have arrived and been multiplied, when the reduction barrier opens, when the last
    workgroup ends, and the gap to the next launch."""
    import numpy as np
    B, NL = 8, 40
    for name, N, K, kind in (("o pred", 1024, 2048, "res"), ("qkv pred", 4096, 1024, "norm"), ("down pred", 1024, 3072, "res"),
                             ("o 1.7B", 2048, 2048, "res"), ("gate/up 1.7B", 12288, 2048, "swiglu"), ("down 1.7B", 2048, 6144, "res")):
        ws = [torch.randn(N * K, device=dev).bfloat16() * 0.02 for _ in range(NL)]
        x = torch.randn(B, K, device=dev).bfloat16()
        nw = torch.ones(K, device=dev).bfloat16()
        out = torch.empty(B, N, dtype=torch.bfloat16, device=dev)
        res = torch.randn(B, N, device=dev).bfloat16()
        buf = torch.zeros(NL * 1024 * 4, dtype=torch.int64, device=dev)

        def run():
            for w in ws:
                if kind == "swiglu":
                    hip.linear_swiglu(x, w, N // 2, norm_w=nw, out=out)
                else:
                    hip.linear_fused(x, w, N, K=K, norm_w=nw if kind == "norm" else None, residual=res if kind != "norm" else None, out=out)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            run()
            hip.lib.qtts_debug_linear_trace(hip.ptr(buf))
            gr = hip.Graph.capture(run)
            hip.lib.qtts_debug_linear_trace(None)
            for _ in range(3):
                gr.launch()
            st.synchronize()
        t = buf.cpu().numpy().reshape(NL, 1024, 4).astype(np.float64) * 10.0      # ns
        nwg = int((t[0, :, 0] > 0).sum())
        t = t[:, :nwg]
        first_start, last_start = t[:, :, 0].min(1), t[:, :, 0].max(1)
        mfma_done, bar_open, end = t[:, :, 1].max(1), t[:, :, 2].max(1), t[:, :, 3].max(1)
        gap = first_start[1:] - end[:-1]
        med = lambda a: float(np.median(a))
        print(f"{name:10s} {nwg:4d} WGs: launch-to-launch {med(np.diff(first_start)):7.0f} ns | gap end->next start {med(gap):6.0f} | "
              f"start spread {med(last_start - first_start):5.0f} | first start -> last MFMA done {med(mfma_done - first_start):6.0f} | "
              f"-> barrier open {med(bar_open - first_start):6.0f} | -> last end {med(end - first_start):6.0f}", flush=True)


def gridbar():
    """Cost of one device-wide barrier inside a resident kernel, for a few barrier shapes."""
    mem = torch.zeros((2 + 64 + 64) * 64, dtype=torch.int32, device=dev)
    for mode, gsize, sleep in ((0, 32, 1), (0, 32, 0), (1, 8, 1), (1, 8, 0), (1, 16, 0), (1, 32, 0), (2, 8, 0), (2, 16, 0), (2, 16, 1)):
        line = f"mode {mode} group {gsize:2d} sleep {sleep}:"
        for nwg in (256, 128, 64):
            res = []
            for iters in (100, 1100):
                def run():
                    hip.check(hip.lib.qtts_debug_grid_barrier(hip.ptr(mem), iters, nwg, 512, mode, gsize, sleep, hip.stream()))
                res.append(timed(run, 1))
            torch.cuda.synchronize()
            line += f"  nwg {nwg:3d}: {(res[1] - res[0]) / 1000:5.2f} us (err {int(mem[0])})"
        print(line, flush=True)


def dominant():
    """A gate/up launch of the path alone (RMSNorm + gate/up GEMM + SwiGLU, 8 rows): the talker's at the 1.7B shape
    cycling through 28 weight buffers, or with QTTS_DOMINANT=predictor the predictor's (hidden 1024, intermediate 3072)
    cycling through 5 - the workloads the PMC traffic passes are collected on."""
    pred = os.environ.get("QTTS_DOMINANT") == "predictor"
    B, H, I, n = (8, 1024, 3072, 5) if pred else (8, 2048, 6144, 28)
    ws = [torch.randn(2 * I * H, device=dev).bfloat16() * 0.02 for _ in range(n)]
    x = torch.randn(B, H, device=dev).bfloat16()
    nw = torch.ones(H, device=dev).bfloat16()
    out = torch.empty(B, I, dtype=torch.bfloat16, device=dev)
    st = torch.cuda.Stream()
    reps = 280 // n
    with torch.cuda.stream(st):
        for _ in range(reps):
            for w in ws:
                hip.linear_swiglu(x, w, I, norm_w=nw, out=out)
        st.synchronize()
    print("launches", reps * n, "algorithmic bytes per launch", 2 * I * H * 2 + B * H * 2 + B * I * 2, flush=True)
    if os.environ.get("QTTS_DOMINANT_TIMED"):
        def run():
            for w in ws:
                hip.linear_swiglu(x, w, I, norm_w=nw, out=out)
        eager = timed(run, len(ws))
        with torch.cuda.stream(st):
            g = hip.Graph.capture(run)
        graph = timed(lambda: g.launch(), len(ws))
        print(f"per launch: eager {eager:.2f} us, inside one graph {graph:.2f} us", flush=True)


def mall():
    """Decode GEMMs at the talker's shapes (8 rows) cycling through 1, 2, 4 or 28 weight matrices inside one graph:
    how much faster a launch is when its weights are still in the memory-side cache (256 MB) or in L2 (8 x 4 MB)."""
    B, H, I = 8, 2048, 6144
    x = torch.randn(B, H, device=dev).bfloat16()
    xi = torch.randn(B, I, device=dev).bfloat16()
    nw = torch.ones(H, device=dev).bfloat16()
    res = torch.randn(B, H, device=dev).bfloat16()
    for name, N, K in (("gate/up", 2 * I, H), ("down", H, I), ("qkv", 4096, H), ("o", H, H)):
        line = f"{name:8s} {N * K * 2 / 1e6:5.1f} MB:"
        for nbuf in (1, 2, 4, 8, 28):
            ws = [torch.randn(N * K, device=dev).bfloat16() * 0.02 for _ in range(nbuf)]
            out = torch.empty(B, N, dtype=torch.bfloat16, device=dev)

            def run():
                for i in range(28):
                    w = ws[i % nbuf]
                    if name == "gate/up":
                        hip.linear_swiglu(x, w, I, norm_w=nw, out=out)
                    elif name == "down":
                        hip.linear_fused(xi, w, N, K=K, residual=res, out=out)
                    elif name == "qkv":
                        hip.linear_fused(x, w, N, K=K, norm_w=nw, out=out)
                    else:
                        hip.linear_fused(x, w, N, K=K, residual=res, out=out)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                run()
                g = hip.Graph.capture(run)
            us = timed(lambda: g.launch(), 28)
            line += f"  {nbuf:2d} bufs {us:6.2f} us ({N * K * 2 / us / 1e6:5.2f} TB/s)"
            del ws
        print(line, flush=True)


def frame():
    import json, tempfile
    sys.path.insert(0, ROOT)
    import bench
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    from nano_qwen3tts_vllm.sampling_params import SamplingParams
    from nano_qwen3tts_vllm.utils import synthetic
    cfg = synthetic.make_config("1.7B")
    tw, pw = bench.make_weights(cfg, dev)
    d = tempfile.mkdtemp()
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    B = 8
    fe = FrameEngine(d, max_streams=B, max_text_frames=64, talker_weights=tw, predictor_weights=pw,
                     num_kvcache_blocks=4 * B, max_num_batched_tokens=4096, codec_eos=None)
    sp = SamplingParams(temperature=1.0, max_tokens=1, do_sample=False)
    plen = int(os.environ.get("QTTS_FRAME_PROMPT", "100"))
    for i in range(B):
        fe.add_stream(torch.randn(plen, 2048, device=dev) * 0.5, torch.randn(8, 2048, device=dev) * 0.1, 60, sp,
                      SamplingParams(temperature=0.9, max_tokens=17, do_sample=False))
    fe.run_frames(3)
    tr, pr = fe.tr, fe.pr
    mctx = (plen + 4 + 127) // 128 * 128
    st, mb = tr.dstage, tr.max_blocks_per_seq

    def talker_body():
        tr.decoder.forward(fe.next_x[:B], st.d("positions")[:B], st.d("slots")[:B], st.d("ctx")[:B],
                           st.d("tables")[:B * mb].view(B, mb), mctx, out=fe.cur_hidden[:B])

    def burst_body():
        pr._burst_body(B)

    hip.lib.qtts_debug_linear(int(os.environ.get("QTTS_LINEAR_DBG", "0")))
    hip.lib.qtts_tune_subtiles(int(os.environ.get("QTTS_SUBTILE_MIN", str(32 << 10))))
    only = os.environ.get("QTTS_FRAME_ONLY")
    for name, body, nk in (("talker 28 layers", talker_body, 1), ("predictor burst", burst_body, 1),
                           ("whole frame", lambda: fe._frame_body(B, mctx), 1)):
        if only and only not in name:
            continue
        with torch.cuda.stream(tr.stream):
            body()
            g = hip.Graph.capture(body)
            us = timed(lambda: g.launch(), 1)
        print(f"{name:18s} {us:8.1f} us per graph launch", flush=True)


def decode_eager():
    """The talker's decode step (28 layers) and the predictor's burst at QTTS_STREAMS (8) streams, launched eagerly a few times (no
    graph: counter collection under the profiler is per launch), contexts of QTTS_FRAME_PROMPT + a few tokens."""
    import json, tempfile
    sys.path.insert(0, ROOT)
    import bench
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    from nano_qwen3tts_vllm.sampling_params import SamplingParams
    from nano_qwen3tts_vllm.utils import synthetic
    cfg = synthetic.make_config("1.7B")
    tw, pw = bench.make_weights(cfg, dev)
    d = tempfile.mkdtemp()
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    B = int(os.environ.get("QTTS_STREAMS", "8"))
    fe = FrameEngine(d, max_streams=B, max_text_frames=64, talker_weights=tw, predictor_weights=pw,
                     num_kvcache_blocks=4 * B, max_num_batched_tokens=4096, codec_eos=None, enforce_eager=True)
    sp = SamplingParams(temperature=1.0, max_tokens=1, do_sample=False)
    plen = int(os.environ.get("QTTS_FRAME_PROMPT", "300"))
    for i in range(B):
        fe.add_stream(torch.randn(plen, 2048, device=dev) * 0.5, torch.randn(8, 2048, device=dev) * 0.1, 60, sp,
                      SamplingParams(temperature=0.9, max_tokens=17, do_sample=False))
    fe.run_frames(int(os.environ.get("QTTS_FRAMES", "6")))
    torch.cuda.synchronize()
    print("done", flush=True)


def rows():
    """Prompt GEMMs at the talker's shapes: rows x (cold weights: 28 different matrices in turn | one hot matrix)."""
    H, I = 2048, 6144
    nl = 28
    hip.lib.qtts_debug_linear(int(os.environ.get("QTTS_LINEAR_DBG", "0")))
    hip.lib.qtts_tune_rows_tm(int(os.environ.get("QTTS_ROWS_TM", "0")))
    shapes = [("qkv", 4096, H, "plain"), ("o_proj", H, H, "res"), ("gate_up", 2 * I, H, "swiglu"), ("down", H, I, "ws")]
    only = os.environ.get("QTTS_ROWS_ONLY")
    for name, N, K, kind in shapes:
        if only and name != only:
            continue
        ws_w = [(torch.randn(N * K // 4, device=dev) * 0.02).bfloat16().repeat(4) for _ in range(nl)]
        for M in [int(v) for v in os.environ.get("QTTS_ROWS", "32,64,128,200,400").split(",")]:
            x = (torch.randn(M + 16, K, device=dev) * 0.5).bfloat16()[:M]     # (room for a padded last row tile)
            res = torch.randn(M, N, device=dev).bfloat16()
            out = torch.empty(M, N // 2 if kind == "swiglu" else N, dtype=torch.bfloat16, device=dev)

            def call(w):
                if kind == "swiglu":
                    hip.linear_swiglu(x, w, N // 2, out=out)
                elif kind == "ws":
                    hip.linear_ws(x, w, N, residual=res, out=out)
                elif kind == "res":
                    hip.linear(x, w, N, residual=res, out=out)
                else:
                    hip.linear(x, w, N, out=out)
            cold = timed(lambda: [call(w) for w in ws_w], nl)
            hot = timed(lambda: [call(ws_w[0]) for _ in range(nl)], nl)
            fl = 2.0 * M * N * K
            print(f"{name:8s} M={M:4d}: cold {cold:7.2f} us ({fl / cold / 1e6:6.1f} TFLOP/s, weights at {N * K * 2 / cold / 1e6:5.2f} TB/s)"
                  f"   hot {hot:7.2f} us", flush=True)


def rowsn():
    """Prompt GEMMs that consume normalised rows, at the predictor's and the talker's shapes, rows in fragment order:
    the rows normalised inside the GEMM from sum-of-squares partials (norm) against rows that arrive normalised
    (plain) and the stand-alone RMSNorm launch that makes them; then the GEMMs that finish residual rows, with and
    without the fragment-order epilogue + partial sums.  Cold weights (one matrix per launch, 28 in turn), in a graph."""
    shapes = [("qkv pred", 4096, 1024, 0), ("gate_up pred", 6144, 1024, 1), ("head pred", 2048, 1024, 0),
              ("qkv 1.7B", 4096, 2048, 0), ("gate_up 1.7B", 12288, 2048, 1)]
    nl = 28
    only = os.environ.get("QTTS_ROWS_ONLY")
    for name, N, K, sw in shapes:
        if only and only not in name:
            continue
        ws_w = [(torch.randn(N * K // 4, device=dev) * 0.02).bfloat16().repeat(4) for _ in range(nl)]
        nw = torch.ones(K, device=dev).bfloat16()
        for M in [int(v) for v in os.environ.get("QTTS_ROWS", "32,64").split(",")]:
            M16 = (M + 15) // 16 * 16
            x = (torch.randn(M16, K, device=dev) * 0.5).bfloat16()
            ssq = torch.rand(M16, K // 16, device=dev) + 0.5
            out = torch.empty(M16, N // 2 if sw else N, dtype=torch.bfloat16, device=dev)
            xn = torch.empty_like(x)

            def call(w, norm):
                hip.check(hip.lib.qtts_debug_linear_rows(hip.ptr(x), K, 1, hip.ptr(ssq) if norm else None, hip.ptr(nw) if norm else None, 1e-6,
                                                         hip.ptr(w), None, 0, 0, hip.ptr(out), N // 2 if sw else N, 0, None, M, N, K, sw, hip.stream()))

            def normk():
                hip.check(hip.lib.qtts_rmsnorm(hip.ptr(x), K, hip.ptr(nw), 1e-6, hip.ptr(xn), K, M, K, hip.stream()))
            res = {}
            for label, fn in (("plain", lambda: [call(w, False) for w in ws_w]), ("norm", lambda: [call(w, True) for w in ws_w]),
                              ("rmsnorm+plain", lambda: [(normk(), call(w, False)) for w in ws_w])):
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    fn()
                    gr = hip.Graph.capture(fn)
                res[label] = timed(lambda: gr.launch(), nl)
            print(f"{name:13s} M={M:4d}: " + "  ".join(f"{k} {v:6.2f} us" for k, v in res.items()), flush=True)


def attn():
    """The talker's fused decode attention launch (q/k norm + RoPE + store + paged attention) at QTTS_ATTN_ROWS rows and
    QTTS_ATTN_CTX tokens of context, cycling through 28 layers' caches inside one graph (cold K/V)."""
    from nano_qwen3tts_vllm.models.hip_decoder import rope_tables
    # QTTS_ATTN_HEADS="2,1" with eight times the rows: the same bytes and workgroups with every (row, KV head)'s cached
    # rows contiguous (what a head-major KV block would look like to the kernel)
    Hq, Hkv = [int(v) for v in os.environ.get("QTTS_ATTN_HEADS", "16,8").split(",")]
    nl = 28
    for B in [int(v) for v in os.environ.get("QTTS_ATTN_ROWS", "8,32,64").split(",")]:
        for ctx in [int(v) for v in os.environ.get("QTTS_ATTN_CTX", "150,290,400,500").split(",")]:
            nb = (ctx + 255) // 256
            nblk = B * nb
            kc = torch.randn(nl, nblk, 256, Hkv, 128, device=dev, dtype=torch.bfloat16)
            vc = torch.randn(nl, nblk, 256, Hkv, 128, device=dev, dtype=torch.bfloat16)
            qkv = torch.randn(B, (Hq + 2 * Hkv) * 128, device=dev).bfloat16()
            qw = torch.ones(128, device=dev).bfloat16()
            ct, st = rope_tables(128, 1000000.0, 2048)
            ct, st = ct.to(dev), st.to(dev)
            bt = torch.zeros(B, 4, dtype=torch.int32)
            for i in range(B):
                bt[i, :nb] = torch.arange(i * nb, (i + 1) * nb)
            bt = bt.to(dev)
            lens = torch.tensor([max(1, ctx - (7 * i) % 40) for i in range(B)], dtype=torch.int32)
            cl = lens.to(dev)
            pos = lens.to(torch.int64).to(dev)
            slots = torch.tensor([int(bt[i, (int(c) - 1) // 256]) * 256 + (int(c) - 1) % 256 for i, c in enumerate(lens)], dtype=torch.int32).to(dev)
            out = torch.empty(B, Hq * 128, dtype=torch.bfloat16, device=dev)
            ws = torch.zeros(hip.lib.qtts_attn_decode_workspace_bytes(B, Hq, Hkv, ctx), dtype=torch.uint8, device=dev)
            line = f"attention B={B:3d} ctx={ctx:4d} ({B * ctx * Hkv * 512 / 1e6:6.1f} MB of K/V):"
            for wide in (0, 1, 2):
                hip.lib.qtts_tune_attn_wide(wide)

                def run():
                    for l in range(nl):
                        hip.check(hip.lib.qtts_attention_decode_fused(hip.ptr(qkv), qkv.stride(0), hip.ptr(qw), hip.ptr(qw), 1e-6, hip.ptr(ct),
                                                                      hip.ptr(st), 2048, hip.ptr(pos), hip.ptr(slots), hip.ptr(kc[l]), hip.ptr(vc[l]),
                                                                      hip.ptr(bt), 4, hip.ptr(cl), ctx, 128 ** -0.5, hip.ptr(out), Hq * 128, B, Hq,
                                                                      Hkv, hip.ptr(ws), hip.stream()))
                sst = torch.cuda.Stream()
                with torch.cuda.stream(sst):
                    run()
                    gr = hip.Graph.capture(run)
                us = timed(lambda: gr.launch(), nl)
                line += f"  {('per shape', '128-token workgroups', '256-token workgroups')[wide]} {us:6.2f} us"
            hip.lib.qtts_tune_attn_wide(0)
            print(line, flush=True)
            del kc, vc


def attntrace():
    """Timestamps inside one fused decode-attention launch at the talker's shape (QTTS_ATTN_ROWS rows, QTTS_ATTN_CTX tokens):
    per workgroup, when it started, had its metadata, had q / k normalised, had its scores (K rows in), had left its
    partial sums in LDS (V rows in), had stored its partials, knew whether it was the row's last, and ended."""
    import numpy as np
    from nano_qwen3tts_vllm.models.hip_decoder import rope_tables
    Hq, Hkv, nl = 16, 8, 4
    names = ["start", "metadata in", "q/k normalised", "scores (K in)", "LDS partials (V in)", "partials stored", "arrival known", "end (merging workgroups)"]
    for B in [int(v) for v in os.environ.get("QTTS_ATTN_ROWS", "8,32,64").split(",")]:
        for ctx in [int(v) for v in os.environ.get("QTTS_ATTN_CTX", "200,300").split(",")]:
            nb = (ctx + 255) // 256
            kc = torch.randn(nl, B * nb, 256, Hkv, 128, device=dev, dtype=torch.bfloat16)
            vc = torch.randn(nl, B * nb, 256, Hkv, 128, device=dev, dtype=torch.bfloat16)
            qkv = torch.randn(B, (Hq + 2 * Hkv) * 128, device=dev).bfloat16()
            qw = torch.ones(128, device=dev).bfloat16()
            ct, st = rope_tables(128, 1000000.0, 2048)
            ct, st = ct.to(dev), st.to(dev)
            bt = torch.zeros(B, 4, dtype=torch.int32)
            for i in range(B):
                bt[i, :nb] = torch.arange(i * nb, (i + 1) * nb)
            bt = bt.to(dev)
            lens = torch.tensor([max(1, ctx - (7 * i) % 40) for i in range(B)], dtype=torch.int32)
            cl, pos = lens.to(dev), lens.to(torch.int64).to(dev)
            slots = torch.tensor([int(bt[i, (int(c) - 1) // 256]) * 256 + (int(c) - 1) % 256 for i, c in enumerate(lens)], dtype=torch.int32).to(dev)
            out = torch.empty(B, Hq * 128, dtype=torch.bfloat16, device=dev)
            ws = torch.zeros(hip.lib.qtts_attn_decode_workspace_bytes(B, Hq, Hkv, ctx), dtype=torch.uint8, device=dev)
            buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)

            uniform = os.environ.get("QTTS_ATTN_UNIFORM") == "1" and ctx <= 32      # rows without metadata (a burst step)

            def call(l):
                if uniform:
                    hip.check(hip.lib.qtts_attention_decode_fused(hip.ptr(qkv), qkv.stride(0), hip.ptr(qw), hip.ptr(qw), 1e-6, hip.ptr(ct), hip.ptr(st),
                                                                  2048, None, None, hip.ptr(kc[l]), hip.ptr(vc[l]), None, 1,
                                                                  None, ctx, 128 ** -0.5, hip.ptr(out), Hq * 128, B, Hq, Hkv, hip.ptr(ws), hip.stream()))
                    return
                hip.check(hip.lib.qtts_attention_decode_fused(hip.ptr(qkv), qkv.stride(0), hip.ptr(qw), hip.ptr(qw), 1e-6, hip.ptr(ct), hip.ptr(st),
                                                              2048, hip.ptr(pos), hip.ptr(slots), hip.ptr(kc[l]), hip.ptr(vc[l]), hip.ptr(bt), 4,
                                                              hip.ptr(cl), ctx, 128 ** -0.5, hip.ptr(out), Hq * 128, B, Hq, Hkv, hip.ptr(ws), hip.stream()))
            for l in range(nl - 1):
                call(l)
            torch.cuda.synchronize()
            hip.lib.qtts_debug_linear_trace(hip.ptr(buf))
            call(nl - 1)
            hip.lib.qtts_debug_linear_trace(None)
            torch.cuda.synchronize()
            t = buf.cpu().numpy().reshape(4096, 8).astype(np.float64) * 10.0
            t = t[t[:, 0] > 0]
            t0 = t[:, 0].min()
            print(f"attention B={B} ctx={ctx}: {len(t)} workgroups, ns after the first workgroup's start (median / max over the workgroups that got there)")
            for k, name in enumerate(names):
                v = t[:, k][t[:, k] > 0] - t0
                if len(v):
                    print(f"  {name:26s} median {np.median(v):7.0f}  max {v.max():7.0f}  (n={len(v)})")
            del kc, vc


def rowtrace():
    """Timestamps inside one prompt GEMM launch (gate/up shape): per workgroup, when the multiplications of tile i
    were issued and when its reduction barrier opened (100 MHz device clock)."""
    import numpy as np
    H, I = 2048, 6144
    M = int(os.environ.get("QTTS_ROWS", "200"))
    which = os.environ.get("QTTS_ROWS_ONLY", "gate_up")
    N, K = {"gate_up": (2 * I, H), "down": (H, I), "qkv": (4096, H), "o_proj": (H, H)}[which]
    ws = [(torch.randn(N * K // 4, device=dev) * 0.02).bfloat16().repeat(4) for _ in range(4)]
    x = (torch.randn(M + 16, K, device=dev) * 0.5).bfloat16()[:M]       # (room for a padded last row tile)
    out = torch.empty(M, I if which == "gate_up" else N, dtype=torch.bfloat16, device=dev)
    buf = torch.zeros(1024 * 32, dtype=torch.int64, device=dev)
    hip.lib.qtts_debug_linear(int(os.environ.get("QTTS_LINEAR_DBG", "0")))

    normed = os.environ.get("QTTS_ROWS_NORM") == "1"        # rows normalised inside the GEMM (fragment-order rows)
    if os.environ.get("QTTS_ROWS_K"):                          # the predictor's shapes: K = 1024
        K = int(os.environ["QTTS_ROWS_K"])
        N = {"gate_up": 6144, "qkv": 4096}[which]
        ws = [(torch.randn(N * K // 4, device=dev) * 0.02).bfloat16().repeat(4) for _ in range(4)]
        x = (torch.randn(M + 16, K, device=dev) * 0.5).bfloat16()[:M]
        out = torch.empty(M + 16, N, dtype=torch.bfloat16, device=dev)
    ssq = torch.rand(M + 16, K // 16, device=dev) + 0.5
    nw = torch.ones(K, device=dev).bfloat16()

    def call(w):
        if normed or os.environ.get("QTTS_ROWS_K"):
            sw = int(which == "gate_up")
            hip.check(hip.lib.qtts_debug_linear_rows(hip.ptr(x), K, 1, hip.ptr(ssq) if normed else None, hip.ptr(nw) if normed else None,
                                                     1e-6, hip.ptr(w), None, 0, 0, hip.ptr(out), N // 2 if sw else N, 0, None, M, N, K, sw,
                                                     hip.stream()))
        elif which == "gate_up":
            hip.linear_swiglu(x, w, I, out=out)
        elif which == "down":
            hip.linear_ws(x, w, N, out=out)
        else:
            hip.linear(x, w, N, out=out)
    for w in ws:
        call(w)
    torch.cuda.synchronize()
    hip.lib.qtts_debug_linear_trace(hip.ptr(buf))
    call(ws[0])
    hip.lib.qtts_debug_linear_trace(None)
    torch.cuda.synchronize()
    if M <= 16:
        t = buf.cpu().numpy()[:4096].reshape(1024, 4).astype(np.float64) * 10.0
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        print(f"{which} M={M}: {len(t)} workgroups; start spread {t[:, 0].max() - t0:.0f} ns; last tile multiplied median "
              f"{np.median(t[:, 1]) - t0:.0f} max {t[:, 1].max() - t0:.0f}; its barrier open max {t[:, 2].max() - t0:.0f}; end max {t[:, 3].max() - t0:.0f} "
              f"median {np.median(t[:, 3]) - t0:.0f} min {t[:, 3].min() - t0:.0f}")
        return
    t = buf.cpu().numpy().reshape(1024, 32).astype(np.float64) * 10.0
    act = t[:, 1] > 0
    t = t[act]
    t0 = t[:, 0].min()
    print(f"{which} M={M} K={K} norm={int(normed)}: {int(act.sum())} active workgroups; start spread {t[:, 0].max() - t0:.0f} ns; last end {t[:, 31].max() - t0:.0f} ns")
    for it in range(15):
        a, b = t[:, 1 + 2 * it], t[:, 2 + 2 * it]
        ok = a > 0
        if not ok.any():
            break
        print(f"  tile {it:2d}: mfma issued median {np.median(a[ok]) - t0:7.0f} max {a[ok].max() - t0:7.0f} | barrier open median "
              f"{np.median(b[ok]) - t0:7.0f} max {b[ok].max() - t0:7.0f}  (n={int(ok.sum())})")


def prefill_batch():
    """One engine step that prefills QTTS_PREFILL_SEQS prompts of QTTS_FRAME_PROMPT tokens together (packed rows)."""
    import json, tempfile
    sys.path.insert(0, ROOT)
    import bench
    from nano_qwen3tts_vllm.llm import TalkerLLM
    from nano_qwen3tts_vllm.sampling_params import SamplingParams
    from nano_qwen3tts_vllm.utils import synthetic
    cfg = synthetic.make_config("1.7B")
    tw, pw = bench.make_weights(cfg, dev)
    d = tempfile.mkdtemp()
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    S = int(os.environ.get("QTTS_PREFILL_SEQS", "8"))
    talker = TalkerLLM(d, num_kvcache_blocks=4 * S + 4, max_num_seqs=64, max_num_batched_tokens=8192, weights=tw)
    sp = SamplingParams(temperature=1.0, max_tokens=1, do_sample=False)
    for plen in [int(v) for v in os.environ.get("QTTS_FRAME_PROMPT", "200").split(",")]:
        ts = []
        for rep in range(5):
            for i in range(S):
                talker.add_request([torch.randn(1, plen, 2048) * 0.5], sp, request_id=f"r{rep}-{i}")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, outs = talker.step_with_outputs()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            assert len(outs) == S, len(outs)
            for i in range(S):
                talker.clear_request(f"r{rep}-{i}")
        print(f"prefill of {S} x {plen} tokens in one step: " + " ".join(f"{v:.2f}" for v in ts) + " ms", flush=True)
    talker.exit()


def prefill():
    """Wall time of a talker prefill (prompt of QTTS_FRAME_PROMPT tokens) through FrameEngine.add_stream."""
    import json, tempfile
    sys.path.insert(0, ROOT)
    import bench
    from nano_qwen3tts_vllm.engine.frame_engine import FrameEngine
    from nano_qwen3tts_vllm.sampling_params import SamplingParams
    from nano_qwen3tts_vllm.utils import synthetic
    cfg = synthetic.make_config("1.7B")
    tw, pw = bench.make_weights(cfg, dev)
    d = tempfile.mkdtemp()
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    fe = FrameEngine(d, max_streams=2, max_text_frames=16, talker_weights=tw, predictor_weights=pw,
                     num_kvcache_blocks=8, max_num_batched_tokens=4096, codec_eos=None)
    sp = SamplingParams(temperature=1.0, max_tokens=1, do_sample=False)
    hip.lib.qtts_debug_linear(int(os.environ.get("QTTS_LINEAR_DBG", "0")))
    hip.lib.qtts_tune_rows_tm(int(os.environ.get("QTTS_ROWS_TM", "0")))
    if os.environ.get("QTTS_ROWS_MIN"):
        hip.lib.qtts_tune_rows(int(os.environ["QTTS_ROWS_MIN"]))
    if os.environ.get("QTTS_MFMA_ATTN_ROWS"):
        hip.lib.qtts_tune_prompt_attention(int(os.environ["QTTS_MFMA_ATTN_ROWS"]))
    for plen in [int(v) for v in os.environ.get("QTTS_FRAME_PROMPT", "200").split(",")]:
        ts = []
        for i in range(6):
            x = torch.randn(plen, 2048, device=dev) * 0.5
            t = torch.randn(8, 2048, device=dev) * 0.1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sid = fe.add_stream(x, t, 8, sp, SamplingParams(temperature=0.9, max_tokens=17, do_sample=False))
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            fe.remove_stream(sid)
        print(f"prefill {plen:4d} tokens: " + " ".join(f"{v:.2f}" for v in ts) + " ms", flush=True)


if __name__ == "__main__":
    _tuning.apply_env_flags()
    {"prefill": prefill, "prefill_batch": prefill_batch, "decode_eager": decode_eager, "rows": rows, "rowsn": rowsn, "attn": attn, "attntrace": attntrace, "rowtrace": rowtrace, "launch": launch, "linear": linear, "linear2": linear2, "frame": frame, "dominant": dominant, "lat": lat, "mall": mall, "gridbar": gridbar, "trace": trace}[sys.argv[1]]()
