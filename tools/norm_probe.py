import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: F401  (the -DQTTS_TUNING build of the library)
from nano_qwen3tts_vllm import hip
sys.path.insert(0, "/root/repo/tools")
dev = torch.device("cuda")
def timed(fn, n, reps=30):
    st = torch.cuda.current_stream()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / n
B = 8
for name, N, K in (("qkv pred", 4096, 1024), ("qkv 1.7B", 4096, 2048), ("head talker", 3072, 2048), ("gate_up pred", 6144, 1024), ("gate_up 1.7B", 12288, 2048)):
    nbuf = max(2, int(600e6 // (N * K * 2)))
    ws = [torch.randn(N * K, device=dev).bfloat16() * 0.02 for _ in range(nbuf)]
    x = torch.randn(B, K, device=dev).bfloat16()
    nw = torch.ones(K, device=dev).bfloat16()
    silu = name.startswith("gate")
    out = torch.empty(B, N // 2 if silu else N, dtype=torch.bfloat16, device=dev)
    line = f"{name:14s}"
    for norm in (False, True):
        def run():
            for w in ws:
                if silu: hip.linear_swiglu(x, w, N // 2, norm_w=nw if norm else None, out=out)
                else: hip.linear(x, w, N, norm_w=nw if norm else None, out=out)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            run(); gr = hip.Graph.capture(run)
            us = timed(lambda: gr.launch(), nbuf)
        line += f"  norm={int(norm)}: {us:5.2f} us"
    print(line, flush=True)
