"""Is a decode GEMM faster when its weights were read recently (Infinity Cache, 256 MB)?  Times the talker's
gate/up and down_proj launches cycling through working sets of different sizes, then the same chain with a side
branch of the graph that reads the next launch's weights while the current one runs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: F401  (the -DQTTS_TUNING build of the library)
from nano_qwen3tts_vllm import hip
from microbench import timed, dev

B = 8


def working_sets():
    for name, N, K, silu in (("gate_up", 12288, 2048, True), ("down", 2048, 6144, False), ("o", 2048, 2048, False)):
        x = torch.randn(B, K, device=dev).bfloat16()
        nw = torch.ones(K, device=dev).bfloat16()
        res = torch.randn(B, N, device=dev).bfloat16()
        out = torch.empty(B, N, dtype=torch.bfloat16, device=dev)
        allw = [torch.randn(N * K, device=dev).bfloat16() * 0.02 for _ in range(max(2, int(1500e6 // (N * K * 2))))]
        line = f"{name:8s} {N*K*2/1e6:5.1f} MB per launch:"
        for total_mb in (50, 100, 200, 400, 1500):
            n = max(1, min(len(allw), int(total_mb * 1e6 // (N * K * 2))))
            ws = allw[:n]
            reps = max(1, 24 // n)

            def run():
                for _ in range(reps):
                    for w in ws:
                        if silu:
                            hip.linear_swiglu(x, w, N // 2, norm_w=nw, out=out[:, :N // 2])
                        else:
                            hip.linear(x, w, N, residual=res, out=out)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                run()
                g = hip.Graph.capture(run)
            us = timed(lambda: g.launch(), n * reps, reps=8)
            line += f"  set {n*N*K*2/1e6:6.0f} MB: {us:5.2f} us"
        print(line, flush=True)


def side_branch():
    """Chain of layers (o, gate_up, down + a small kernel standing for attention); variant with a second branch
    that reads layer l+1's weights (torch reduction) while layer l runs."""
    H, I = 2048, 6144
    L = 14
    x = torch.randn(B, H, device=dev).bfloat16()
    xi = torch.randn(B, I, device=dev).bfloat16()
    nw = torch.ones(H, device=dev).bfloat16()
    res = torch.randn(B, H, device=dev).bfloat16()
    out = torch.empty(B, I, dtype=torch.bfloat16, device=dev)
    outh = torch.empty(B, H, dtype=torch.bfloat16, device=dev)
    layers = []
    for _ in range(L):
        blob = torch.randn((H * H + 2 * I * H + H * I), device=dev).bfloat16() * 0.02     # one allocation per layer
        layers.append((blob, blob[:H * H], blob[H * H:H * H + 2 * I * H], blob[H * H + 2 * I * H:]))
    sink = torch.zeros(L, dtype=torch.float32, device=dev)

    def layer(l):
        _, wo, wgu, wd = layers[l]
        hip.check(hip.lib.qtts_rmsnorm(hip.ptr(x), H, hip.ptr(nw), 1e-6, hip.ptr(outh), H, B, H, hip.stream()), 'rmsnorm')   # stands for the attention launch
        hip.linear(x, wo, H, residual=res, out=outh)
        hip.linear_swiglu(x, wgu, I, norm_w=nw, out=out)
        hip.linear(xi, wd, H, residual=res, out=outh)

    def plain():
        for l in range(L):
            layer(l)

    side = torch.cuda.Stream()

    def with_prefetch():
        main = torch.cuda.current_stream()
        for l in range(L):
            if l + 1 < L:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    layers[l + 1][0].view(torch.float32).sum()
            layer(l)
            if l + 1 < L:
                main.wait_stream(side)

    for name, fn in (("plain", plain), ("side-branch read of the next layer", with_prefetch)):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            fn()
            torch.cuda.synchronize()
            g = hip.Graph.capture(fn)
        us = timed(lambda: g.launch(), L, reps=8)
        print(f"{name:40s} {us:6.2f} us per layer", flush=True)


if __name__ == "__main__":
    working_sets()
    if len(sys.argv) > 1:
        side_branch()
