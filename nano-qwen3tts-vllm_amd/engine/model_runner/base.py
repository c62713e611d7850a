"""Shared runner plumbing: checkpoint reading, KV sizing, batch metadata staging.

Reference: P/engine/model_runner/base.py:29-312.  What differs by design:
  * no process group, no tensor-parallel shared-memory RPC (one replica per GPU);
  * batch metadata (positions, slot mapping, context lengths, block tables, row→
    sequence map) is built in ONE pinned host buffer and uploaded with ONE copy
    into a device buffer at fixed offsets, which is also what captured graphs read;
  * prefill attention reads the just-written paged cache with per-token context
    lengths (position+1), so prefill and decode share one attention kernel and
    the `cu_seqlens` form of the reference is only used on the host.
"""
from __future__ import annotations

import json
import os

import torch

from ... import hip
from ...config import Config, Qwen3TTSConfig
from ...models.hip_decoder import BLOCK
from ..sequence import Sequence


def read_checkpoint(model_dir: str, want_prefix: str, strip: str, skip_prefixes=()):
    """Tensors of `model.safetensors` whose key starts with `want_prefix`, with `strip`
    removed (the prefix handling of P/models/qwen3_tts_talker.py:81-109 and
    qwen3_tts_predictor.py:63-80), read lazily so unrelated tables stay on disk."""
    from safetensors import safe_open

    out = {}
    with safe_open(os.path.join(model_dir, "model.safetensors"), framework="pt", device="cpu") as f:
        for k in f.keys():
            if not k.startswith(want_prefix) or any(k.startswith(s) for s in skip_prefixes):
                continue
            out[k[len(strip):]] = f.get_tensor(k)
    return out


class Staging:
    """int32 host(pinned)+device staging area with named, fixed-offset fields."""

    def __init__(self, fields: list[tuple[str, int, torch.dtype]], device):
        self.off = {}
        n = 0
        for name, count, dt in fields:
            if dt == torch.int64:
                n = (n + 1) & ~1
                self.off[name] = (n, count * 2, dt)
                n += count * 2
            else:
                self.off[name] = (n, count, dt)
                n += count
        n = (n + 3) & ~3
        self.host = torch.zeros(n, dtype=torch.int32)
        if torch.cuda.is_available():
            self.host = self.host.pin_memory()
        self.dev = torch.zeros(n, dtype=torch.int32, device=device)

    def h(self, name):
        o, c, dt = self.off[name]
        v = self.host[o:o + c]
        return v.view(dt) if dt != torch.int32 else v

    def d(self, name):
        o, c, dt = self.off[name]
        v = self.dev[o:o + c]
        return v.view(dt) if dt != torch.int32 else v

    def upload(self):
        self.dev.copy_(self.host, non_blocking=True)


class ModelRunner:
    def __init__(self, config: Config, rank: int = 0, event=None):
        assert rank == 0 and config.tensor_parallel_size == 1
        if not torch.cuda.is_available():
            raise RuntimeError("the HIP model runner needs a GPU (no CPU fallback exists)")
        self.config = config
        self.block_size = config.kvcache_block_size
        assert self.block_size == BLOCK
        self.enforce_eager = config.enforce_eager
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.full_config = Qwen3TTSConfig.from_model_dir(config.model)
        self.max_blocks_per_seq = (config.max_model_len + self.block_size - 1) // self.block_size
        # QTTS_SEED pins the sampler's counter RNG (reproducible runs); unset, every process draws its own stream, as
        # the reference's unseeded torch generator does
        seed = os.environ.get("QTTS_SEED")
        self.rng_seed = int(seed) if seed is not None else (int.from_bytes(os.urandom(8), "little") >> 1)
        self.rng_offset = 0
        # All launches go to one side stream: hipGraph capture is not allowed on the legacy
        # default stream, and a private stream keeps replays ordered with the staging copies.
        self.stream = torch.cuda.Stream(device=self.device)

    def on_stream(self):
        """Context manager: make `self.stream` current, ordered after the caller's stream on entry
        and with the caller's stream ordered after it on exit."""
        runner = self

        class _Ctx:
            def __enter__(self_inner):
                self_inner.outer = torch.cuda.current_stream(runner.device)
                runner.stream.wait_stream(self_inner.outer)
                self_inner.cm = torch.cuda.stream(runner.stream)
                self_inner.cm.__enter__()

            def __exit__(self_inner, *exc):
                self_inner.cm.__exit__(*exc)
                self_inner.outer.wait_stream(runner.stream)
                return False

        return _Ctx()

    # ---- sizing -----------------------------------------------------------------
    def kv_blocks_for(self, block_bytes: int, reserve_bytes: int) -> int:
        """As P/engine/model_runner/base.py:131-151, capped at what max_num_seqs can use."""
        cfg = self.config
        if cfg.num_kvcache_blocks and cfg.num_kvcache_blocks > 0:
            return cfg.num_kvcache_blocks
        free, total = torch.cuda.mem_get_info()
        budget = total * cfg.gpu_memory_utilization
        if cfg.process_gpu_memory_fraction is not None:
            budget *= cfg.process_gpu_memory_fraction
        used = total - free
        n = int(budget - used - reserve_bytes) // block_bytes
        cap = cfg.max_num_seqs * self.max_blocks_per_seq
        n = max(1, min(n, cap))
        cfg.num_kvcache_blocks = n
        return n

    # ---- batch metadata (host lists, as the reference's prepare_* build them) -----
    @staticmethod
    def prefill_metadata(seqs: list[Sequence], block_size: int):
        positions, slots, row_seq, ctx, cu = [], [], [], [], [0]
        for si, seq in enumerate(seqs):
            L = len(seq)
            positions.extend(range(L))
            row_seq.extend([si] * L)
            ctx.extend(range(1, L + 1))
            cu.append(cu[-1] + L)
            bt = seq.block_table
            assert len(bt) * block_size >= L, "sequence has no KV blocks allocated"
            slots.extend(bt[t // block_size] * block_size + t % block_size for t in range(L))
        return positions, slots, row_seq, ctx, cu

    def call(self, method_name, *args):
        return getattr(self, method_name)(*args)

    def exit(self):
        torch.cuda.synchronize()

    def next_rng_offset(self) -> int:
        self.rng_offset += 1
        return self.rng_offset
