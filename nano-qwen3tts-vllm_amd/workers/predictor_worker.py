"""Predictor worker (reference: P/workers/predictor_worker.py).  A run_step command
runs the engine until nothing is waiting or running — a burst — and answers with every
sequence that finished during it (predictor_worker.py:125-134)."""
from __future__ import annotations

import logging
import sys

import torch

from .protocol import (CMD_ADD_REQUEST, CMD_CLEAR_REQUEST, CMD_RUN_STEP, CMD_SHUTDOWN, deserialize_command,
                       sampling_params_from_dict, serialize_predictor_result)

logger = logging.getLogger(__name__)
_sampling_params_from_dict = sampling_params_from_dict


def serve(predictor_llm, recv, send, device=None) -> int:
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    bursts = 0
    while True:
        cmd = deserialize_command(recv())
        kind = cmd.get("cmd")
        if kind == CMD_SHUTDOWN:
            return bursts
        if kind == CMD_ADD_REQUEST:
            embeds = [torch.from_numpy(a).to(device) for a in cmd["inputs_embeds"]]
            predictor_llm.add_request(embeds, sampling_params_from_dict(cmd.get("sampling_params", {})),
                                      request_id=cmd["request_id"])
        elif kind == CMD_CLEAR_REQUEST:
            predictor_llm.clear_request(cmd["request_id"])
        elif kind == CMD_RUN_STEP:
            finished = []
            try:
                while predictor_llm.scheduler.waiting or predictor_llm.scheduler.running:
                    outputs, _ = predictor_llm.step()
                    finished.extend(outputs)
            except Exception:
                logger.exception("[predictor_worker] burst failed")
                send(serialize_predictor_result(cmd["step_id"], []))
                raise
            send(serialize_predictor_result(cmd["step_id"], finished))
            bursts += 1
        else:
            logger.warning("[predictor_worker] unknown command: %r", kind)


def run_predictor_worker(command_connect_addr: str, result_connect_addr: str, model_path: str, *,
                         gpu_memory_utilization: float = 0.9, enforce_eager: bool = False,
                         tensor_parallel_size: int = 1, engine_factory: str | None = None) -> None:
    """The worker process's body (reference: P/workers/predictor_worker.py:42-145): build the engine, connect to the API
    process's command and result addresses (`tcp://` -> ZeroMQ as the reference, `pipe://` -> workers/transport.py),
    serve until shutdown.  engine_factory ("module:callable", called with the model path and the engine keywords)
    replaces PredictorLLM - stand-in engines for tests of the process and transport plumbing."""
    from .transport import open_worker_channels
    if engine_factory:
        import importlib
        mod, _, attr = engine_factory.partition(":")
        predictor_llm = getattr(importlib.import_module(mod), attr)(model_path, enforce_eager=enforce_eager,
                                                                    tensor_parallel_size=tensor_parallel_size)
    else:
        from ..interface import _compute_memory_split
        from ..llm import PredictorLLM
        mem = _compute_memory_split(model_path, gpu_memory_utilization)
        predictor_llm = PredictorLLM(model_path, enforce_eager=enforce_eager, tensor_parallel_size=tensor_parallel_size,
                                     gpu_memory_utilization=min(mem["predictor_util"], 0.85),
                                     process_gpu_memory_fraction=mem["process_gpu_memory_fraction"])
    recv, send, close = open_worker_channels(command_connect_addr, result_connect_addr)
    try:
        serve(predictor_llm, recv, send)
    finally:
        close()
        try:
            predictor_llm.exit()
        except Exception:
            pass


def main():
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("--command_connect", required=True)
    p.add_argument("--result_connect", required=True)
    p.add_argument("--model_path", required=True)
    p.add_argument("--gpu_memory_utilization", type=float, default=0.9)
    p.add_argument("--enforce_eager", action="store_true")
    p.add_argument("--tensor_parallel_size", type=int, default=1)
    p.add_argument("--engine_factory", default=None)
    a = p.parse_args()
    logging.basicConfig(level=logging.INFO)
    run_predictor_worker(a.command_connect, a.result_connect, a.model_path,
                         gpu_memory_utilization=a.gpu_memory_utilization, enforce_eager=a.enforce_eager,
                         tensor_parallel_size=a.tensor_parallel_size, engine_factory=a.engine_factory)


if __name__ == "__main__":
    main()
    sys.exit(0)
