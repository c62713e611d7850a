"""Asyncio orchestrator loops in the API process (reference: P/zmq/engine_loop_mp.py:27-158).

Each loop waits until some request has fed its engine, optionally lingers a few
milliseconds so concurrent requests land in the same batch, asks the worker for one
step (talker) or one burst (predictor), and fans the results out to the per-request
queues as `(engine_type, msg_type, payload)` tuples."""
from __future__ import annotations

import asyncio
import logging
import os
import time
from typing import Any

logger = logging.getLogger(__name__)


def _float_env(name: str, default: float) -> float:
    try:
        return float(os.environ.get(name, default))
    except (TypeError, ValueError):
        return default


PREDICTOR_COLLECT_MS = _float_env("PREDICTOR_COLLECT_MS", 3.0)
PREFILL_COLLECT_MS = _float_env("PREFILL_COLLECT_MS", 5.0)
POLL_S = 0.0005


async def _active(request_queues: dict, lock: asyncio.Lock) -> set:
    async with lock:
        return set(request_queues.keys())


async def _put(request_queues: dict, lock: asyncio.Lock, request_id, item) -> None:
    async with lock:
        q = request_queues.get(request_id)
    if q is not None:
        try:
            q.put_nowait(item)
        except asyncio.QueueFull:
            pass


async def run_talker_loop_mp(talker_client: Any, request_queues: dict, queues_lock: asyncio.Lock,
                             talker_ready: set) -> None:
    while True:
        await asyncio.sleep(POLL_S)
        if not talker_ready:
            continue
        active = await _active(request_queues, queues_lock)
        if len(talker_ready) < len(active):          # give late prefills a moment to join the batch
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1000 < PREFILL_COLLECT_MS:
                await asyncio.sleep(0.001)
                if talker_ready >= await _active(request_queues, queues_lock):
                    break
        if not talker_ready:
            continue
        try:
            _, outputs_all = await talker_client.run_step_async()
        except asyncio.CancelledError:
            break
        except Exception:
            logger.exception("[talker_loop_mp] step failed")
            continue
        served = set()
        for request_id, _seq_id, token_ids, hidden_states, is_finished in outputs_all or []:
            served.add(request_id)
            await _put(request_queues, queues_lock, request_id,
                       ("talker", "token", {"token_ids": token_ids, "hidden_states": hidden_states}))
            if is_finished:
                await _put(request_queues, queues_lock, request_id, ("talker", "done", {}))
        talker_ready -= served


async def run_predictor_loop_mp(predictor_client: Any, request_queues: dict, queues_lock: asyncio.Lock,
                                predictor_ready: set) -> None:
    while True:
        await asyncio.sleep(POLL_S)
        if not predictor_ready:
            continue
        active = await _active(request_queues, queues_lock)
        if len(predictor_ready) < len(active) and len(active) > 1:
            await asyncio.sleep(PREDICTOR_COLLECT_MS / 1000.0)
        if not predictor_ready:
            continue
        try:
            _, outputs_all = await predictor_client.run_step_async()
        except asyncio.CancelledError:
            break
        except Exception:
            logger.exception("[predictor_loop_mp] burst failed")
            continue
        for request_id, _seq_id, token_ids in outputs_all or []:
            await _put(request_queues, queues_lock, request_id, ("predictor", "token", {"token_ids": token_ids}))
            predictor_ready.discard(request_id)
