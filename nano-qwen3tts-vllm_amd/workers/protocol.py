"""Command / result wire format between the API process and the engine workers
(reference: P/workers/protocol.py:14-136).  Pickled dicts; tensors travel as numpy
arrays, bfloat16 widened to float32 because numpy has no bf16 — values are therefore
exact, and the receiving engine rounds to bf16 once when it packs its batch."""
from __future__ import annotations

import pickle
from typing import Any

import numpy as np
import torch

CMD_ADD_REQUEST = "add_request"
CMD_RUN_STEP = "run_step"
CMD_CLEAR_REQUEST = "clear_request"
CMD_SHUTDOWN = "shutdown"

_PROTO = pickle.HIGHEST_PROTOCOL


def _tensor_to_numpy(x: Any) -> Any:
    if isinstance(x, torch.Tensor):
        t = x.detach().cpu()
        return (t.float() if t.dtype == torch.bfloat16 else t).numpy()
    return x


def _numpy_to_tensor(x: Any, device: str = "cuda") -> Any:
    return torch.from_numpy(x).to(device) if isinstance(x, np.ndarray) else x


def _add_request(request_id, inputs_embeds, sampling_params: dict) -> bytes:
    return pickle.dumps({"cmd": CMD_ADD_REQUEST, "request_id": request_id,
                         "inputs_embeds": [_tensor_to_numpy(t) for t in inputs_embeds],
                         "sampling_params": sampling_params}, protocol=_PROTO)


def serialize_talker_add_request(request_id: str, inputs_embeds, sampling_params: dict) -> bytes:
    return _add_request(request_id, inputs_embeds, sampling_params)


def serialize_predictor_add_request(request_id: str, inputs_embeds, sampling_params: dict) -> bytes:
    return _add_request(request_id, inputs_embeds, sampling_params)


def serialize_talker_run_step(step_id: str) -> bytes:
    return pickle.dumps({"cmd": CMD_RUN_STEP, "step_id": step_id}, protocol=_PROTO)


def serialize_predictor_run_step(step_id: str) -> bytes:
    return pickle.dumps({"cmd": CMD_RUN_STEP, "step_id": step_id}, protocol=_PROTO)


def serialize_clear_request(request_id: str) -> bytes:
    return pickle.dumps({"cmd": CMD_CLEAR_REQUEST, "request_id": request_id}, protocol=_PROTO)


def serialize_shutdown() -> bytes:
    return pickle.dumps({"cmd": CMD_SHUTDOWN}, protocol=_PROTO)


def deserialize_command(payload: bytes) -> dict:
    return pickle.loads(payload)


def serialize_talker_result(step_id: str, outputs_all: list) -> bytes:
    """outputs_all: (request_id, seq_id, token_ids, last_hidden_state, is_finished) per sequence."""
    out = [(rid, sid, toks, None if h is None else _tensor_to_numpy(h), fin) for rid, sid, toks, h, fin in outputs_all]
    return pickle.dumps({"step_id": step_id, "outputs_all": out}, protocol=_PROTO)


def deserialize_talker_result(payload: bytes):
    obj = pickle.loads(payload)
    return obj["step_id"], obj["outputs_all"]


def serialize_predictor_result(step_id: str, outputs_all: list) -> bytes:
    """outputs_all: (request_id, seq_id, token_ids) per finished sequence."""
    return pickle.dumps({"step_id": step_id, "outputs_all": outputs_all}, protocol=_PROTO)


def deserialize_predictor_result(payload: bytes):
    obj = pickle.loads(payload)
    return obj["step_id"], obj["outputs_all"]


def sampling_params_to_dict(sp) -> dict:
    return {"temperature": sp.temperature, "max_tokens": sp.max_tokens, "ignore_eos": sp.ignore_eos,
            "do_sample": sp.do_sample, "top_k": sp.top_k, "top_p": sp.top_p}


def sampling_params_from_dict(d: dict):
    from ..sampling_params import SamplingParams
    return SamplingParams(temperature=d.get("temperature", 1.0), max_tokens=d.get("max_tokens", 64),
                          ignore_eos=d.get("ignore_eos", False), do_sample=d.get("do_sample", True),
                          top_k=d.get("top_k", 50), top_p=d.get("top_p", 1.0))
