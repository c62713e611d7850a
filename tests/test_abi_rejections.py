"""What the C ABI refuses (include/qtts_hip.h): every entry point validates its arguments on the host before it
launches anything, returns QTTS_EINVAL (-1) and leaves a message naming the entry point in qtts_last_error().
Validation is host code, so it is exercised here without a device; the pointers below are never dereferenced (they
are made-up, suitably aligned addresses), which is also why this file refuses to run on a box that has a GPU — there a
check that went missing would launch a kernel on them."""
import ctypes

import pytest
import torch

from nano_qwen3tts_vllm import hip

pytestmark = pytest.mark.skipif(torch.cuda.is_available(), reason="made-up device pointers: host-side validation only")

P = 0x7F0000001000          # "a device pointer": non-null, 4 KiB aligned
INTS = (ctypes.c_int, ctypes.c_long, ctypes.c_longlong, ctypes.c_uint64, ctypes.c_int64, ctypes.c_size_t, ctypes.c_uint32)


def err():
    return hip.lib.qtts_last_error().decode()


def test_every_entry_point_refuses_null_and_zero():
    n = 0
    for name in sorted(hip.EXPORTED):
        fn = getattr(hip.lib, name)
        if fn.restype is not ctypes.c_int or name in ("qtts_version", "qtts_graph_begin"):
            continue
        args = [0 if a in INTS else 0.0 if a in (ctypes.c_float, ctypes.c_double) else None for a in fn.argtypes]
        assert fn(*args) == -1, name
        assert ":" in err(), (name, err())
        n += 1
    assert n >= 30


def linear(x=P, ldx=2048, w=P, bias=None, res=None, ldr=0, nw=None, out=P, ldo=2048, M=8, N=2048, K=2048):
    return hip.lib.qtts_linear(x, ldx, w, bias, res, ldr, nw, 1e-6, out, ldo, M, N, K, None)


@pytest.mark.parametrize("kwargs,message", [
    (dict(K=100, ldx=104), "K%32==0"),
    (dict(N=2040, ldo=2040), "N%16==0"),
    (dict(ldx=1024), "x row stride"),
    (dict(ldx=2052), "x row stride"),                    # not a multiple of 8 elements
    (dict(ldo=1024), "out row stride"),
    (dict(res=P, ldr=16), "residual row stride"),
    (dict(x=P + 8), "misaligned"),
    (dict(out=P + 2), "misaligned"),
    (dict(M=0), "empty shape"),
    (dict(K=32 * 11, ldx=352), "not supported"),         # 11 k-steps: no split into at most 8 waves
])
def test_linear_refuses(kwargs, message):
    assert linear(**kwargs) == -1
    assert err().startswith("linear:") and message in err(), err()


def test_other_shape_and_range_checks():
    L = hip.lib
    assert L.qtts_rmsnorm(P, 16, P, 1e-6, P, 16, 4, 12, None) == -1 and err().startswith("rmsnorm:")             # H % 8
    assert L.qtts_rmsnorm(P, 8, P, 1e-6, P, 16, 4, 16, None) == -1                                                # ldx < H
    assert L.qtts_pack_weight(P, P, 24, 64, None) == -1 and "N%16==0" in err()
    assert L.qtts_pack_weight_fp8(P, P, P, 16, 96, None) == -1 and "K%64==0" in err()
    assert L.qtts_pack_gate_up(P, P, P, 12, 64, None) == -1 and "I%8==0" in err()
    assert L.qtts_linear_swiglu(P, 2048, P, None, 1e-6, P, 512, 8, 1024, 2048, None) == -1 and "out row stride" in err()
    assert L.qtts_store_kvcache(P, 512, P, 1024, P, P, P, 4, 8, None) == -1 and err().startswith("store_kvcache:")  # ldk < Hkv*128
    assert L.qtts_qknorm_rope_store(P, 1024, P, P, 1e-6, P, P, 2048, P, P, P, P, 4, 16, 8, None) == -1             # ldqkv too small
    assert L.qtts_qknorm_rope_store(P, 4096, P, P, 1e-6, P, P, 0, P, P, P, P, 4, 16, 8, None) == -1 and "rope table" in err()
    # picks: vocabulary bound of the one-workgroup sampler, top_k range
    assert L.qtts_pick_token(P, 8192, P, None, 50, 1, 0, None, P, 1, 8, 8192, None) == -1 and "V=8192" in err()
    assert L.qtts_pick_token(P, 3072, P, None, 0, 1, 0, None, P, 1, 8, 3072, None) == -1 and "top_k=0" in err()
    assert L.qtts_sample_topk(P, 3072, P, 4000, 1, 0, P, 8, 3072, None) == -1 and "top_k=4000" in err()
    assert L.qtts_argmax(P, 100, P, 4, 3072, None) == -1                                                          # ld < V
    assert L.qtts_embedding_gather(P, P, P, 2048, 4, 2044, 3072, None) == -1                                      # H % 8
    assert L.qtts_predictor_input(P, P, P, P, 8, 2048, 0, None) == -1                                             # empty vocabulary
    assert L.qtts_frame_glue(P, P, P, P, 4, P, 2048, None, P, None, 8, 2048, 3072, 2048, None) == -1             # ldp < B
    assert L.qtts_frame_glue(P, P, P, P, 8, P, 2044, None, P, None, 8, 2044, 3072, 2048, None) == -1 and "multiple of 8" in err()
    assert L.qtts_advance_decode(P, P, P, P, 0, None, None, None, 1024, 8, None) == -1                            # no block table columns


def attention(fn, static, B=8, Hq=16, Hkv=8, ctx=128, max_blocks=4, ldqkv=4096, ldo=2048, ws=None, ctx_ptr=P):
    args = [P, ldqkv, P, P, 1e-6, P, P, 2048, P, P, P, P, P, max_blocks, ctx_ptr, ctx, 0.088, P, ldo, B, Hq, Hkv]
    if not static:
        args.append(ws)
    return fn(*args, None)


def test_attention_refuses():
    L = hip.lib
    f, s = L.qtts_attention_decode_fused, L.qtts_attention_decode_fused_static
    assert attention(f, False, Hq=16, Hkv=5) == -1                                  # heads not a multiple of KV heads
    assert attention(f, False, ldqkv=2048) == -1                                    # qkv rows too short for the heads
    assert attention(f, False, ctx=2048, max_blocks=4) == -1                        # context beyond the block table
    assert attention(f, False, ctx=512, ws=None) == -1 and "workspace required" in err()
    assert attention(f, False, Hq=24, Hkv=8, ldqkv=5120, ldo=3072) == -1 and "head ratio" in err()
    # the entry point with the built-in scratch: its stated limits
    assert attention(s, True, B=32, Hq=16) == -1 and err().startswith("attention_decode_fused_static:")      # B*Hq > 256
    assert attention(s, True, ctx=2048, max_blocks=8) == -1 and err().startswith("attention_decode_fused_static:")
    assert L.qtts_paged_attention(P, 2048, P, P, P, 4, None, P, 2048, 0.088, P, 2048, 8, 16, 8, None, None) == -1   # ctx > blocks*256
    assert L.qtts_paged_attention(P, 2048, P, P, P, 4, None, P, 512, 0.088, P, 2048, 8, 16, 8, None, None) == -1 and "workspace" in err()
    assert L.qtts_paged_attention(P, 1024, P, P, P, 4, None, P, 128, 0.088, P, 2048, 8, 16, 8, None, None) == -1 and "strides" in err()


def test_model_handle_refuses():
    L = hip.lib
    assert L.qtts_model_create(2048, 28, 16, 8, 64, 6144, 1e-6) is None and "head_dim" in err()       # head_dim must be 128
    m = L.qtts_model_create(2048, 2, 16, 8, 128, 6144, 1e-6)
    assert m
    try:
        assert L.qtts_model_set_layer(m, 2, P, P, P, P, P, P, P, P) == -1 and "layer index" in err()
        assert L.qtts_model_set_rope(m, P, P, 0) == -1
        assert L.qtts_model_set_kv_cache(m, P, 0) == -1
        assert L.qtts_model_set_workspace(m, P + 16, 1 << 30, 16, 256) == -1 and "aligned" in err()
        assert L.qtts_model_set_workspace(m, P, 1024, 16, 256) == -1 and "required" in err()
        assert L.qtts_model_workspace_bytes(m, 16, 256) > 1024
        # a model without weights, rope tables, KV cache and workspace must not run
        rc = L.qtts_model_forward(m, P, P, P, None, P, P, 4, 128, 8, P, None)
        assert rc != 0 and "not fully configured" in err(), err()
        assert L.qtts_model_set_layer_events(m, 99, None, None) == -1 and "unknown launch" in err()
    finally:
        L.qtts_model_destroy(m)
