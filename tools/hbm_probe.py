"""What a plain device copy / reduction of 50 MB buffers (cycling through 2 GB, i.e. from HBM) achieves per launch:
the yardstick for the 50 MB weight stream of the dominant GEMM launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device("cuda", 0)
ws = [torch.randn(12288 * 2048, device=dev).bfloat16() for _ in range(40)]
dst = torch.empty_like(ws[0])
acc = torch.zeros(1, device=dev)


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


t = timed(lambda: [dst.copy_(w) for w in ws], len(ws))
print(f"copy 50 MB -> 50 MB: {t:.2f} us per launch, {2 * ws[0].numel() * 2 / t / 1e6:.2f} TB/s read+write")
t = timed(lambda: [torch.sum(w, dtype=torch.float32) for w in ws], len(ws))
print(f"sum of 50 MB: {t:.2f} us per launch (may be 2 kernels), {ws[0].numel() * 2 / t / 1e6:.2f} TB/s read")
v = [w.view(torch.int32) for w in ws]
t = timed(lambda: [torch.bitwise_xor(a, 1, out=dst.view(torch.int32)) for a in v], len(ws))
print(f"xor 50 MB -> 50 MB: {t:.2f} us per launch, {2 * ws[0].numel() * 2 / t / 1e6:.2f} TB/s read+write")
