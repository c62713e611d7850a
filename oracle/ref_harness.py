"""TEST INFRASTRUCTURE — not part of the product path.

Loads the upstream reference's own Python modules (read-only, from
/root/reference) inside THIS build container so that golden vectors can be
generated from the reference itself.  Nothing here travels to the GPU box as
a dependency: the GPU box only sees the fixtures under tests/golden/.

What the harness supplies (none of it is reference code):
  * a stand-in `flash_attn` module: the two entry points the reference calls
    (layers/attention.py:88-95) implemented as the textbook fp32
    softmax(QK^T*scale)V with a bottom-right aligned causal mask, GQA by head
    repeat, and block-table gathering for the paged decode call;
  * a CPU replacement for the Triton `store_kvcache` (layers/attention.py:33-54):
    an index copy that skips slot == -1;
  * a 1-rank gloo process group (layer constructors query it,
    layers/linear.py:23-24);
  * TORCHDYNAMO_DISABLE=1 so the reference's @torch.compile decorators run
    eagerly.
"""
import importlib
import importlib.util
import os
import sys
import types

REF_ROOT = "/root/reference/nano-qwen3tts-vllm"
REF_NAME = "nano_qwen3tts_vllm"


def _flash_attn_module():
    import torch

    mod = types.ModuleType("flash_attn")
    # libraries that probe for optional packages with importlib.util.find_spec need a spec on the stand-in
    import importlib.machinery
    mod.__spec__ = importlib.machinery.ModuleSpec("flash_attn", None)

    def _attend(q, k, v, scale, causal_offset):
        # q [Lq,Hq,D], k/v [Lk,Hkv,D]; fp32 textbook attention.
        Lq, Hq, D = q.shape
        Lk, Hkv, _ = k.shape
        g = Hq // Hkv
        qf = q.float().transpose(0, 1)                      # [Hq,Lq,D]
        kf = k.float().repeat_interleave(g, dim=1).transpose(0, 1)
        vf = v.float().repeat_interleave(g, dim=1).transpose(0, 1)
        s = torch.matmul(qf, kf.transpose(1, 2)) * scale     # [Hq,Lq,Lk]
        iq = torch.arange(Lq).unsqueeze(1) + causal_offset   # bottom-right aligned
        ik = torch.arange(Lk).unsqueeze(0)
        s = s.masked_fill(ik > iq, float("-inf"))
        p = torch.softmax(s, dim=-1)
        o = torch.matmul(p, vf)                              # [Hq,Lq,D]
        return o.transpose(0, 1).to(q.dtype)

    def flash_attn_varlen_func(q, k, v, max_seqlen_q=None, cu_seqlens_q=None,
                               max_seqlen_k=None, cu_seqlens_k=None,
                               softmax_scale=None, causal=True, block_table=None):
        assert causal and block_table is None
        outs = []
        S = cu_seqlens_q.numel() - 1
        for i in range(S):
            q0, q1 = int(cu_seqlens_q[i]), int(cu_seqlens_q[i + 1])
            k0, k1 = int(cu_seqlens_k[i]), int(cu_seqlens_k[i + 1])
            outs.append(_attend(q[q0:q1], k[k0:k1], v[k0:k1], softmax_scale,
                                (k1 - k0) - (q1 - q0)))
        return torch.cat(outs, dim=0)

    def flash_attn_with_kvcache(q, k_cache, v_cache, cache_seqlens=None,
                                block_table=None, softmax_scale=None, causal=True):
        # q [B,1,Hq,D]; caches [num_blocks, block, Hkv, D]
        B = q.shape[0]
        bs = k_cache.shape[1]
        outs = []
        for b in range(B):
            n = int(cache_seqlens[b])
            nb = (n + bs - 1) // bs
            blocks = [int(x) for x in block_table[b, :nb]]
            kk = torch.cat([k_cache[i] for i in blocks], dim=0)[:n]
            vv = torch.cat([v_cache[i] for i in blocks], dim=0)[:n]
            outs.append(_attend(q[b], kk, vv, softmax_scale, n - 1))
        return torch.stack(outs, dim=0)

    mod.flash_attn_varlen_func = flash_attn_varlen_func
    mod.flash_attn_with_kvcache = flash_attn_with_kvcache
    return mod


def load_reference():
    """Register the reference package under its own name and return it."""
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    import torch
    import torch.distributed as dist

    if REF_NAME in sys.modules:
        return sys.modules[REF_NAME]
    sys.modules["flash_attn"] = _flash_attn_module()
    spec = importlib.util.spec_from_file_location(
        REF_NAME, os.path.join(REF_ROOT, "__init__.py"),
        submodule_search_locations=[REF_ROOT])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules[REF_NAME] = pkg
    spec.loader.exec_module(pkg)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29613")
        dist.init_process_group("gloo", rank=0, world_size=1)

    attn = importlib.import_module(REF_NAME + ".layers.attention")

    def store_kvcache_cpu(key, value, k_cache, v_cache, slot_mapping):
        N, H, D = key.shape
        kc = k_cache.view(-1, H, D)
        vc = v_cache.view(-1, H, D)
        for i in range(N):
            s = int(slot_mapping[i])
            if s == -1:
                continue
            kc[s] = key[i]
            vc[s] = value[i]

    attn.store_kvcache = store_kvcache_cpu
    return pkg
