"""CPU: host-side scheduling logic against traces recorded from the reference's own
Scheduler / BlockManager / TalkerScheduler (tests/golden/sched.json, written by
oracle/gen_golden.py), plus the C-ABI export check (no compute without a GPU)."""
import ctypes
import json
import os
import re
from types import SimpleNamespace

import torch

from helpers import GOLDEN
from nano_qwen3tts_vllm.engine.llm_engine.talker_llm_engine import TalkerScheduler
from nano_qwen3tts_vllm.engine.scheduler import Scheduler
from nano_qwen3tts_vllm.engine.sequence import Sequence
from nano_qwen3tts_vllm.sampling_params import SamplingParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    return SimpleNamespace(max_num_seqs=4, max_num_batched_tokens=2048, eos=-1, num_kvcache_blocks=6, kvcache_block_size=256)


def test_talker_scheduler_trace():
    trace = json.load(open(os.path.join(GOLDEN, "sched.json")))["talker"]
    ts = TalkerScheduler(_cfg())
    sp = SamplingParams(temperature=1.0, max_tokens=1)
    seqs = {}
    for rid, L in {"a": 10, "b": 255, "c": 300}.items():
        s = Sequence([], input_embeds=torch.zeros(1, L, 8), sampling_params=sp, request_id=rid)
        ts.request_id_to_seq[rid] = s
        ts.add(s)
        seqs[rid] = s

    def snap(batch, pre):
        return {"prefill": bool(pre),
                "seqs": [{"rid": s.request_id, "len": len(s), "block_table": list(s.block_table),
                          "last_block_num_tokens": s.last_block_num_tokens} for s in batch],
                "free": list(ts.block_manager.free_block_ids)}

    batch, pre = ts.schedule()
    assert snap(batch, pre) == trace[0]
    ts.postprocess(batch, [11] * len(batch), [None] * len(batch))
    for rec in trace[1:-1]:
        for rid in rec["fed"]:
            seqs[rid].decode_input_embeds = torch.zeros(1, 1, 8)
        batch, pre = ts.schedule()
        got = {"fed": rec["fed"], **snap(batch, pre)}
        assert got == rec
        ts.postprocess(batch, [12] * len(batch), [None] * len(batch))
    ts.clear_request("b")
    assert {"cleared": "b", "free": list(ts.block_manager.free_block_ids),
            "running": [s.request_id for s in ts.running]} == trace[-1]


def test_predictor_scheduler_trace():
    trace = json.load(open(os.path.join(GOLDEN, "sched.json")))["predictor"]
    ps = Scheduler(_cfg())
    sp = SamplingParams(temperature=0.9, max_tokens=17)
    for rid in ["x", "y", "z"]:
        ps.add(Sequence([], input_embeds=torch.zeros(1, 2, 8), sampling_params=sp, request_id=rid))
    got = []
    while not ps.is_finished():
        batch, pre = ps.schedule()
        rec = {"prefill": bool(pre), "seqs": [{"rid": s.request_id, "len": len(s), "block_table": list(s.block_table)} for s in batch]}
        ps.postprocess(batch, [5] * len(batch))
        rec["finished"] = [s.request_id for s in batch if s.is_finished]
        rec["free"] = list(ps.block_manager.free_block_ids)
        got.append(rec)
    assert got == trace
    assert len(got) == 15            # 1 prefill + 14 decodes, 15 generated tokens (max_tokens=17 counts the 2 prompt positions)


def test_prompt_of_exactly_one_block_can_decode():
    """The reference asserts on the first decode of a 256-token prompt (block_manager.py:102-103);
    this build's manager appends a fresh block instead."""
    ts = TalkerScheduler(_cfg())
    s = Sequence([], input_embeds=torch.zeros(1, 256, 8), sampling_params=SamplingParams(max_tokens=1), request_id="r")
    ts.request_id_to_seq["r"] = s
    ts.add(s)
    batch, pre = ts.schedule()
    assert pre and len(s.block_table) == 1
    ts.postprocess(batch, [1], [None])
    s.decode_input_embeds = torch.zeros(1, 1, 8)
    batch, pre = ts.schedule()
    assert not pre and batch == [s] and len(s.block_table) == 2


def test_sampling_params_floor():
    import pytest
    with pytest.raises(AssertionError):
        SamplingParams(temperature=0.0)


def test_c_abi_exports_every_declared_symbol():
    """libqtts_hip.so loads without a GPU and exports every function include/qtts_hip.h declares."""
    header = open(os.path.join(ROOT, "include", "qtts_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(qtts_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = ctypes.CDLL(os.path.join(ROOT, "nano-qwen3tts-vllm_amd", "csrc", "libqtts_hip.so"))
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    from nano_qwen3tts_vllm import hip
    assert set(hip.EXPORTED) == declared
    assert hip.lib.qtts_version() >= 100
    # argument validation happens before any launch, so it can be exercised without a device
    assert hip.lib.qtts_linear(None, 0, None, None, None, 0, None, 0.0, None, 0, 1, 16, 32, None) == -1
    assert hip.lib.qtts_last_error().startswith(b"linear:")


def test_single_caller_guard_is_reentrant_for_one_thread_and_refuses_a_second():
    """engine/guards.py: the frame engine's public calls hold this guard - re-entrant for the thread inside (the poll
    hook admits streams from within step), an error for another thread arriving meanwhile (two host threads would
    interleave stream records and waits that are only safe in one host order)."""
    import threading
    from nano_qwen3tts_vllm.engine.guards import SingleCaller
    g = SingleCaller("engine")
    seen = []
    with g:
        with g:                                   # same thread, nested
            seen.append("nested")

        def other():
            try:
                with g:
                    seen.append("entered")
            except RuntimeError as e:
                seen.append(str(e))
        t = threading.Thread(target=other)
        t.start()
        t.join()
    assert seen[0] == "nested" and "another thread" in seen[1]
    t = threading.Thread(target=other)            # free again once the first thread has left
    t.start()
    t.join()
    assert seen[2] == "entered"
