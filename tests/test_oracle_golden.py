"""CPU: the oracle (oracle/qtts_oracle.py) reproduces, bit for bit, what the
reference's own modules produced in the build container (tests/golden/*.npz,
written by oracle/gen_golden.py).  Attention arithmetic is the fp32 textbook
definition on both sides (flash_attn is absent: parity unpinned for that op)."""
import numpy as np
import pytest
import torch

import qtts_oracle as O
from helpers import bf, golden_weights, load_npz, oracle_cfgs, weight_checksum


def eq(a, b):
    a = a.float().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.array_equal(a, np.asarray(b))


def test_ops_bit_exact():
    g = load_npz("ops.npz")
    assert eq(O.rms_norm(bf(g["rms_x"]), bf(g["rms_w"]), 1e-6), g["rms_y"])
    assert eq(O.rms_norm(bf(g["hn_x"]), bf(g["hn_w"]), 1e-6), g["hn_y"])
    pos = torch.from_numpy(g["rope_pos"])
    q, k = bf(g["rope_q"]), bf(g["rope_k"])
    cos, sin = O.rope_tables(pos, 128, 1000000.0)
    qo, ko = O.apply_rope(q, k, cos, sin)
    # 3-row interleaved M-RoPE with equal rows == 1-D RoPE (both reference classes)
    assert eq(qo, g["mrope_qo"]) and eq(ko, g["mrope_ko"])
    assert eq(qo, g["rope1d_qo"]) and eq(ko, g["rope1d_ko"])
    cos, sin = O.rope_tables(pos, 128, 10000.0)
    qo, ko = O.apply_rope(q, k, cos, sin)
    assert eq(qo, g["mrope_ni_qo"]) and eq(ko, g["mrope_ni_ko"])
    assert eq(O.silu_mul(bf(g["act_g"]), bf(g["act_u"])), g["act_y"])


@pytest.mark.parametrize("size,seed,tag", [("tiny", 11, "tiny"), ("tiny-id", 12, "tiny_id")])
def test_models_bit_exact(size, seed, tag):
    g = load_npz(f"model_{tag}.npz")
    cfg, sd, tw, pw = golden_weights(size, seed)
    assert abs(weight_checksum(sd) - float(g["weight_checksum"])) < 1e-6 * float(g["weight_checksum"])
    tc, pc = oracle_cfgs(cfg)

    # talker prefill + 3 decodes with the fixture's block tables
    talker = O.OracleTalker(tw, tc, num_blocks=8)
    tables = [[int(b) for b in row if b >= 0] for row in g["t_tables"]]
    order = [b for row in tables for b in row]
    talker.pages.free = order + [b for b in range(8) if b not in order]
    prompts = [torch.from_numpy(g[f"t_prompt{i}"]) for i in range(3)]
    seqs, toks, hl, logits = talker.prefill(prompts)
    assert [s.block_table for s in seqs] == tables
    assert eq(hl, g["t_prefill_hidden"]) and eq(logits, g["t_prefill_logits"])
    for step in range(3):
        toks, h, lg = talker.decode(seqs, torch.from_numpy(g[f"t_dec{step}_in"]))
        assert eq(h, g[f"t_dec{step}_hidden"]), step
        assert eq(lg, g[f"t_dec{step}_logits"]), step

    # predictor burst
    pred = O.OraclePredictor(pw, pc, num_blocks=4)
    pred.pages.free = [2, 0, 3, 1]
    pin = torch.from_numpy(g["p_in"])
    toks, all_logits = pred.burst([pin[i] for i in range(3)], return_logits=True)
    assert np.array_equal(np.array(toks), g["p_tokens"])
    assert eq(torch.stack(all_logits, 1), g["p_logits"])

    # frame loop (talker -> predictor burst -> fp32 glue -> talker)
    prompts = [torch.from_numpy(g[f"f_prompt{i}"]) for i in range(2)]
    frames, margins, _ = O.frame_loop(tw, tc, pw, pc, prompts, torch.from_numpy(g["f_text_adds"]), 6)
    assert np.array_equal(frames.numpy(), g["f_frames"])
    assert np.allclose(margins.numpy(), g["f_margins"])


def test_fp8_row_quantisation_properties():
    """The fp8 extension's quantiser: power-of-two scales, |q| <= 448, q * scale exactly a bf16 number, zero rows
    keep scale 1, and the error against the original stays within half an e4m3 step of the row scale."""
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(48, 256, generator=g) * 0.07).bfloat16()
    w[2] = 0
    w[5, 0] = 448.0 * 2.0 ** -5
    w[5] = w[5].clamp(-14.0, 14.0)
    q, scale, deq = O.quantize_fp8_rows(w)
    assert torch.equal(torch.frexp(scale)[0], torch.full_like(scale, 0.5))          # powers of two
    assert float(scale[2]) == 1.0 and float(q[2].float().abs().max()) == 0.0
    assert float(q.float().abs().max()) <= 448.0
    assert float(scale[5]) == 2.0 ** -5 and float(q[5, 0].float()) == 448.0
    assert torch.equal(deq.float(), q.float() * scale[:, None])
    amax = w.float().abs().amax(1)
    assert bool(((amax / scale <= 448.0) & ((amax / scale > 224.0) | (amax == 0))).all())
    err = (deq.float() - w.float()).abs()
    assert bool((err <= w.float().abs() * 2.0 ** -4 + scale[:, None] * 2.0 ** -10).all())
