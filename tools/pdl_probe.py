"""Probe (not product): a chain of dependent decode-GEMM launches run the plain way (one queue, kernel boundaries)
and with each launch started before its predecessor has ended (two queues carrying alternate launches, a completion
word per launch, agent-scope stores and loads instead of the boundary's cache maintenance) - tools/pdl_probe.hip.

Build in the container:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/libpdl_probe.so tools/pdl_probe.hip
Run on the GPU box:      python3 tools/pdl_probe.py
Prints microseconds per launch for both forms, whether the two chains end with the same bytes, and how many waits
ran out (must be 0)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(ROOT, "libpdl_probe.so"))
lib.pdl_probe_launch.restype = ctypes.c_int
lib.pdl_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                 ctypes.c_void_p]

dev = torch.device("cuda:0")
NL = int(os.environ.get("PDL_LAUNCHES", "40"))
N, K = 4096, 1024
g = torch.Generator(device="cpu").manual_seed(3)
ws = [(torch.randn(N * K, generator=g) * 0.05).bfloat16().to(dev) for _ in range(NL)]
x0 = (torch.randn(8, N, generator=g)).bfloat16().to(dev)
bufs = [torch.zeros(8, N, dtype=torch.bfloat16, device=dev) for _ in range(2)]
flags = torch.zeros((NL + 1) * 256, dtype=torch.int32, device=dev)      # a word per launch, or 256 (one per workgroup)
counters = torch.zeros(NL + 1, dtype=torch.int32, device=dev)
timeouts = torch.zeros(1, dtype=torch.int32, device=dev)


PER_WG = False
ONE_QUEUE = False            # the overlapped form's kernels on one queue: what the protocol alone costs


def launch(i, overlap, stream):
    src = x0 if i == 0 else bufs[(i - 1) % 2]
    dst = bufs[i % 2]
    fin = flags[(i - 1) * 256:].data_ptr() if (overlap and i > 0) else None
    rc = lib.pdl_probe_launch(src.data_ptr(), N, ws[i].data_ptr(), dst.data_ptr(), N, N, 1 if overlap else 0, fin,
                              flags[i * 256:].data_ptr(), None if PER_WG else counters[i:i + 1].data_ptr(), 1,
                              timeouts.data_ptr(), stream.cuda_stream)
    assert rc == 0, rc


def chain_plain(s1, s2):
    for i in range(NL):
        launch(i, False, s1)


def chain_overlap(s1, s2):
    flags.zero_()                                   # on s1 (current), in front of the fork
    fork = torch.cuda.Event()
    fork.record(s1)
    s2.wait_event(fork)
    for i in range(NL):
        launch(i, True, s1 if (i % 2 == 0 or ONE_QUEUE) else s2)
    join = torch.cuda.Event()
    join.record(s2)
    s1.wait_event(join)


def timed(chain, label):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        chain(s1, s2)                               # eager once (also checks the protocol outside a graph)
        s1.synchronize()
        s2.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s1):
            chain(s1, s2)
        for _ in range(3):
            graph.replay()
        s1.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        a.record(s1)
        for _ in range(reps):
            graph.replay()
        b.record(s1)
        b.synchronize()
    us = a.elapsed_time(b) * 1e3 / (reps * NL)
    result = bufs[(NL - 1) % 2].clone()
    print(f"{label:28s} {us:6.2f} us per launch ({NL} launches of {N * K * 2 / 1e6:.1f} MB per replay)", flush=True)
    return us, result


if __name__ == "__main__":
    p_us, p_out = timed(chain_plain, "one queue, kernel boundaries")
    o_us, o_out = timed(chain_overlap, "two queues, word per launch")
    PER_WG = True
    w_us, w_out = timed(chain_overlap, "two queues, word per workgroup")
    ONE_QUEUE = True
    timed(chain_overlap, "one queue, word per workgroup")
    PER_WG = False
    timed(chain_overlap, "one queue, word per launch")
    same = torch.equal(p_out, o_out) and torch.equal(p_out, w_out)
    print(f"same bytes at the end of the chains: {same}; waits that ran out: {int(timeouts.item())}; finite: {bool(torch.isfinite(o_out.float()).all())}")
    sys.exit(0 if same and int(timeouts.item()) == 0 else 1)
