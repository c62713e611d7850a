"""ctypes binding of libqtts_hip.so (C ABI: include/qtts_hip.h).

torch is used only as the owner of device memory and streams: every call below
passes raw device pointers, sizes and the current HIP stream handle.  There is no
CPU fallback — if the library is missing, import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QTTS_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libqtts_hip.so")   # QTTS_LIB: A/B builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python nano-qwen3tts-vllm_amd/csrc/build.py` "
        "(hipcc --offload-arch=gfx950). The HIP path has no fallback.")

lib = C.CDLL(LIB_PATH)

_p, _i, _f, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64

_SIGS = {
    "qtts_version": (_i, []),
    "qtts_last_error": (C.c_char_p, []),
    "qtts_packed_weight_bytes": (_sz, [_i, _i]),
    "qtts_pack_weight": (_i, [_p, _p, _i, _i, _p]),
    "qtts_pack_gate_up": (_i, [_p, _p, _p, _i, _i, _p]),
    "qtts_linear": (_i, [_p, _i, _p, _p, _p, _i, _p, _f, _p, _i, _i, _i, _i, _p]),
    "qtts_linear_fused": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _p, _f, _p, _i, _i, _p, _i, _i, _i, _i, _p]),
    "qtts_attention_decode_fused": (_i, [_p, _i, _p, _p, _f, _p, _p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _f, _p, _i, _i,
                                         _i, _i, _p, _p]),
    "qtts_attention_decode_fused_static": (_i, [_p, _i, _p, _p, _f, _p, _p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _f, _p, _i, _i,
                                                _i, _i, _p]),
    "qtts_model_forward_logits": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _i, _p, _p]),
    "qtts_model_forward_logits_pairs": (_i, [_p, _p, _p, _p, _i, _p, _p, _i, _p, _p]),
    "qtts_model_forward_logits_gathered": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p, _p, _i, _p, _p]),
    "qtts_linear_workspace_bytes": (_sz, [_i, _i, _i]),
    "qtts_linear_ws": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "qtts_packed_weight_fp8_bytes": (_sz, [_i, _i]),
    "qtts_pack_weight_fp8": (_i, [_p, _p, _p, _i, _i, _p]),
    "qtts_pack_gate_up_fp8": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "qtts_linear_fp8": (_i, [_p, _i, _p, _p, _p, _p, _i, _p, _f, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "qtts_linear_swiglu_fp8": (_i, [_p, _i, _p, _p, _p, _f, _p, _i, _i, _i, _i, _p]),
    "qtts_linear_swiglu": (_i, [_p, _i, _p, _p, _f, _p, _i, _i, _i, _i, _p]),
    "qtts_rmsnorm": (_i, [_p, _i, _p, _f, _p, _i, _i, _i, _p]),
    "qtts_qknorm_rope_store": (_i, [_p, _i, _p, _p, _f, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "qtts_store_kvcache": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _p]),
    "qtts_attn_workspace_bytes": (_sz, [_i, _i, _i]),
    "qtts_attn_decode_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "qtts_paged_attention": (_i, [_p, _i, _p, _p, _p, _i, _p, _p, _i, _f, _p, _i, _i, _i, _i, _p, _p]),
    "qtts_argmax": (_i, [_p, _i, _p, _i, _i, _p]),
    "qtts_sample_topk": (_i, [_p, _i, _p, _i, _u64, _u64, _p, _i, _i, _p]),
    "qtts_pick_token": (_i, [_p, _i, _p, _p, _i, _u64, _u64, _p, _p, _i, _i, _i, _p]),
    "qtts_embedding_gather": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "qtts_predictor_input": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "qtts_frame_glue": (_i, [_p, _p, _p, _p, _i, _p, C.c_long, _p, _p, _p, _i, _i, _i, _i, _p]),
    "qtts_advance_decode": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _p]),
    "qtts_model_create": (_p, [_i, _i, _i, _i, _i, _i, _f]),
    "qtts_model_destroy": (None, [_p]),
    "qtts_model_set_layer_fp8": (_i, [_p, _i] + [_p] * 12),
    "qtts_model_set_layer": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "qtts_model_set_final_norm": (_i, [_p, _p]),
    "qtts_model_set_layer_events": (_i, [_p, _i, _p, _p]),
    "qtts_model_set_rope": (_i, [_p, _p, _p, _i]),
    "qtts_model_set_kv_cache": (_i, [_p, _p, _i]),
    "qtts_model_workspace_bytes": (_sz, [_p, _i, _i]),
    "qtts_model_set_workspace": (_i, [_p, _p, _sz, _i, _i]),
    "qtts_model_forward": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "qtts_graph_begin": (_i, [_p]),
    "qtts_graph_end": (_p, [_p]),
    "qtts_graph_launch": (_i, [_p, _p]),
    "qtts_graph_destroy": (None, [_p]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = list(_SIGS)


class QttsError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        raise QttsError(f"{what} failed (rc={rc}): {lib.qtts_last_error().decode()}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------- tensor-level wrappers
def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """[N,K] bf16 device tensor -> packed MFMA-fragment-order buffer (same byte size)."""
    assert w.is_cuda and w.dtype == torch.bfloat16 and w.dim() == 2 and w.is_contiguous()
    N, K = w.shape
    out = torch.empty(N * K, dtype=torch.bfloat16, device=w.device)
    check(lib.qtts_pack_weight(ptr(w), ptr(out), N, K, stream()), "pack_weight")
    return out


def pack_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    assert gate.shape == up.shape and gate.is_contiguous() and up.is_contiguous()
    I, K = gate.shape
    out = torch.empty(2 * I * K, dtype=torch.bfloat16, device=gate.device)
    check(lib.qtts_pack_gate_up(ptr(gate), ptr(up), ptr(out), I, K, stream()), "pack_gate_up")
    return out


def linear(x, w_packed, N, bias=None, residual=None, norm_w=None, eps=1e-6, out=None):
    M, K = x.shape
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    check(lib.qtts_linear(ptr(x), x.stride(0), ptr(w_packed), ptr(bias), ptr(residual),
                          residual.stride(0) if residual is not None else 0, ptr(norm_w), eps,
                          ptr(out), out.stride(0), M, N, K, stream()), "linear")
    return out


def pack_weight_fp8(w: torch.Tensor):
    """[N, K] bf16 -> (fp8 fragments as uint8 [N*K], fp32 power-of-two scale per row)."""
    assert w.dtype == torch.bfloat16 and w.is_contiguous()
    N, K = w.shape
    packed = torch.empty(N * K, dtype=torch.uint8, device=w.device)
    scales = torch.empty(N, dtype=torch.float32, device=w.device)
    check(lib.qtts_pack_weight_fp8(ptr(w), ptr(packed), ptr(scales), N, K, stream()), "pack_weight_fp8")
    return packed, scales


def pack_gate_up_fp8(gate: torch.Tensor, up: torch.Tensor):
    assert gate.shape == up.shape and gate.is_contiguous() and up.is_contiguous()
    I, K = gate.shape
    packed = torch.empty(2 * I * K, dtype=torch.uint8, device=gate.device)
    scales = torch.empty(2 * I, dtype=torch.float32, device=gate.device)
    check(lib.qtts_pack_gate_up_fp8(ptr(gate), ptr(up), ptr(packed), ptr(scales), I, K, stream()), "pack_gate_up_fp8")
    return packed, scales


def linear_fp8(x, w_packed, w_scales, N, bias=None, residual=None, norm_w=None, eps=1e-6, out=None):
    M, K = x.shape
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    nbytes = int(lib.qtts_linear_workspace_bytes(M, N, K)) if norm_w is None else 0
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    check(lib.qtts_linear_fp8(ptr(x), x.stride(0), ptr(w_packed), ptr(w_scales), ptr(bias), ptr(residual),
                              residual.stride(0) if residual is not None else 0, ptr(norm_w), eps, ptr(out),
                              out.stride(0), M, N, K, ptr(ws), nbytes, stream()), "linear_fp8")
    return out


def linear_swiglu_fp8(x, w_gu_packed, w_scales, I, norm_w=None, eps=1e-6, out=None):
    M, K = x.shape
    if out is None:
        out = torch.empty(M, I, dtype=torch.bfloat16, device=x.device)
    check(lib.qtts_linear_swiglu_fp8(ptr(x), x.stride(0), ptr(w_gu_packed), ptr(w_scales), ptr(norm_w), eps, ptr(out),
                                     out.stride(0), M, I, K, stream()), "linear_swiglu_fp8")
    return out


def linear_ws(x, w_packed, N, bias=None, residual=None, out=None):
    """qtts_linear_ws with a workspace of the size the library asks for."""
    M, K = x.shape
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    nbytes = int(lib.qtts_linear_workspace_bytes(M, N, K))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    check(lib.qtts_linear_ws(ptr(x), x.stride(0), ptr(w_packed), ptr(bias), ptr(residual),
                             residual.stride(0) if residual is not None else 0, ptr(out), out.stride(0), M, N, K,
                             ptr(ws), nbytes, stream()), "linear_ws")
    return out


def linear_fused(x, w_packed, N, K=None, gather_ids=None, bias=None, residual=None, norm_w=None, eps=1e-6,
                 xnorm_out=None, swiglu_input=False, out=None, M=None):
    """qtts_linear_fused; with gather_ids, x is the table and M = len(gather_ids)."""
    if gather_ids is not None:
        M = gather_ids.shape[0] if M is None else M
        rows = x.shape[0]
        K = x.shape[1] if K is None else K
    else:
        M = x.shape[0] if M is None else M
        rows = 0
        K = (x.shape[1] // 2 if swiglu_input else x.shape[1]) if K is None else K
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    check(lib.qtts_linear_fused(ptr(x), x.stride(0), ptr(gather_ids), rows, ptr(w_packed), ptr(bias), ptr(residual),
                                residual.stride(0) if residual is not None else 0, ptr(norm_w), eps, ptr(xnorm_out),
                                xnorm_out.stride(0) if xnorm_out is not None else 0, 1 if swiglu_input else 0,
                                ptr(out), out.stride(0), M, N, K, stream()), "linear_fused")
    return out


def linear_swiglu(x, w_gu_packed, I, norm_w=None, eps=1e-6, out=None):
    M, K = x.shape
    if out is None:
        out = torch.empty(M, I, dtype=torch.bfloat16, device=x.device)
    check(lib.qtts_linear_swiglu(ptr(x), x.stride(0), ptr(w_gu_packed), ptr(norm_w), eps, ptr(out), out.stride(0),
                                 M, I, K, stream()), "linear_swiglu")
    return out


def rmsnorm(x, w, eps=1e-6):
    M, H = x.shape
    y = torch.empty_like(x)
    check(lib.qtts_rmsnorm(ptr(x), x.stride(0), ptr(w), eps, ptr(y), y.stride(0), M, H, stream()), "rmsnorm")
    return y


def qknorm_rope_store(qkv, q_w, k_w, eps, cos, sin, positions, slot_mapping, k_cache, v_cache, Hq, Hkv):
    N = qkv.shape[0]
    check(lib.qtts_qknorm_rope_store(ptr(qkv), qkv.stride(0), ptr(q_w), ptr(k_w), eps, ptr(cos), ptr(sin),
                                     cos.shape[0], ptr(positions), ptr(slot_mapping), ptr(k_cache), ptr(v_cache),
                                     N, Hq, Hkv, stream()), "qknorm_rope_store")


def store_kvcache(k, v, k_cache, v_cache, slot_mapping):
    N = k.shape[0]
    Hkv = k_cache.shape[-2]
    k2, v2 = k.reshape(N, -1), v.reshape(N, -1)
    check(lib.qtts_store_kvcache(ptr(k2), k2.stride(0), ptr(v2), v2.stride(0), ptr(k_cache), ptr(v_cache),
                                 ptr(slot_mapping), N, Hkv, stream()), "store_kvcache")


def paged_attention(q, k_cache, v_cache, block_tables, context_lens, max_context, scale, Hq, Hkv, row_seq=None,
                    out=None):
    """q: [N, >=Hq*128] (row stride may be larger); returns [N, Hq*128]."""
    N = q.shape[0]
    if out is None:
        out = torch.empty(N, Hq * 128, dtype=torch.bfloat16, device=q.device)
    wsb = lib.qtts_attn_workspace_bytes(N, Hq, max_context)
    ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
    check(lib.qtts_paged_attention(ptr(q), q.stride(0), ptr(k_cache), ptr(v_cache), ptr(block_tables),
                                   block_tables.shape[1], ptr(row_seq), ptr(context_lens), max_context, scale,
                                   ptr(out), out.stride(0), N, Hq, Hkv, ptr(ws), stream()), "paged_attention")
    return out


def argmax(logits):
    M, V = logits.shape
    out = torch.empty(M, dtype=torch.int64, device=logits.device)
    check(lib.qtts_argmax(ptr(logits), logits.stride(0), ptr(out), M, V, stream()), "argmax")
    return out


def sample_topk(logits, temperatures, top_k=50, seed=0, offset=0):
    M, V = logits.shape
    out = torch.empty(M, dtype=torch.int64, device=logits.device)
    check(lib.qtts_sample_topk(ptr(logits), logits.stride(0), ptr(temperatures), top_k, seed, offset, ptr(out), M, V,
                               stream()), "sample_topk")
    return out


def embedding_gather(table, ids):
    M = ids.shape[0]
    vocab, H = table.shape
    out = torch.empty(M, H, dtype=torch.bfloat16, device=table.device)
    check(lib.qtts_embedding_gather(ptr(table), ptr(ids), ptr(out), out.stride(0), M, H, vocab, stream()),
          "embedding_gather")
    return out


class Graph:
    """A captured sequence of library calls (hipGraph), replayable on a stream."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def capture(fn):
        s = stream()
        if s == 0:
            raise QttsError("graph capture needs a non-default stream (use torch.cuda.stream(...))")
        check(lib.qtts_graph_begin(s), "graph_begin")
        try:
            fn()
        finally:
            h = lib.qtts_graph_end(s)
        if not h:
            raise QttsError("graph capture failed: " + lib.qtts_last_error().decode())
        return Graph(h)

    def launch(self):
        check(lib.qtts_graph_launch(self._h, stream()), "graph_launch")

    def __del__(self):
        try:
            if self._h:
                lib.qtts_graph_destroy(self._h)
                self._h = None
        except Exception:
            pass
