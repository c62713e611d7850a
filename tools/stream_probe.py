"""Per-launch time of the 8-row, K=2048, N=12288 decode GEMM (50 MB of weights, inside one graph, cold weights)
with and without the RMSNorm prologue and the SwiGLU epilogue; QTTS_PROBE_SIZES=1 sweeps the weight bytes instead."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: F401  (the -DQTTS_TUNING build of the library)
from nano_qwen3tts_vllm import hip
from tools.microbench import timed

dev = torch.device("cuda", 0)
x = torch.randn(8, 2048, device=dev).bfloat16()
nw = torch.ones(2048, device=dev).bfloat16()


def per_launch(fn, ws):
    def run():
        for w in ws:
            fn(w)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        run()
        g = hip.Graph.capture(run)
    return timed(lambda: g.launch(), len(ws))


if os.environ.get("QTTS_PROBE_SIZES"):
    for N in (6144, 12288, 24576, 49152):
        n = max(4, int(2.2e9 / (N * 2048 * 2)))
        ws = [(torch.randn(N * 2048 // 8, device=dev) * 0.02).bfloat16().repeat(8) for _ in range(n)]
        out = torch.empty(8, N, dtype=torch.bfloat16, device=dev)
        t = per_launch(lambda w: hip.linear(x, w, N, out=out), ws)
        print(f"N={N:6d}: {N * 4096 / 1e6:6.1f} MB  {t:7.2f} us per launch  {N * 4096 / t / 1e6:5.2f} TB/s", flush=True)
        del ws
elif os.environ.get("QTTS_PROBE_PRED"):
    # the predictor's gate/up launch (K=1024, N=6144: 384 tiles) with 1, 2 or 3 tiles per workgroup
    N, K = 6144, 1024
    xk = torch.randn(8, K, device=dev).bfloat16()
    nk = torch.ones(K, device=dev).bfloat16()
    ws = [(torch.randn(N * K // 8, device=dev) * 0.02).bfloat16().repeat(8) for _ in range(60)]
    out = torch.empty(8, N // 2, dtype=torch.bfloat16, device=dev)
    for t in (0, 4, 3, 2):
        hip.lib.qtts_tune_linear(0, t)
        tm = per_launch(lambda w: hip.linear_swiglu(xk, w, N // 2, norm_w=nk, out=out), ws)
        print(f"target {t * 64 or 'auto':>4} workgroups: {tm:6.2f} us per launch", flush=True)
    hip.lib.qtts_tune_linear(0, 0)
else:
    N = 12288
    ws = [(torch.randn(N * 2048 // 8, device=dev) * 0.02).bfloat16().repeat(8) for _ in range(40)]
    out = torch.empty(8, N, dtype=torch.bfloat16, device=dev)
    for name, fn in (("plain", lambda w: hip.linear(x, w, N, out=out)),
                     ("norm", lambda w: hip.linear(x, w, N, norm_w=nw, out=out)),
                     ("swiglu", lambda w: hip.linear_swiglu(x, w, N // 2, out=out)),
                     ("norm+swiglu", lambda w: hip.linear_swiglu(x, w, N // 2, norm_w=nw, out=out))):
        t = per_launch(fn, ws)
        print(f"{name:12s} {t:7.2f} us per launch  {N * 4096 / t / 1e6:5.2f} TB/s", flush=True)
