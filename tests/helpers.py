"""Shared helpers for the parity tests (fixtures → torch tensors, synthetic weights)."""
import os

import numpy as np
import torch

from nano_qwen3tts_vllm.utils import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def bf(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16)


SHARP = dict(head_sharp=1.0, head_std=0.05)     # as oracle/gen_golden.py: heads whose greedy picks have clear winners

# fixture tag -> (model size, weight seed, extra make_state_dict arguments), as oracle/gen_golden.py::main
FIXTURES = {"tiny": ("tiny", 11, {}), "tiny_id": ("tiny-id", 12, {}), "sharp": ("tiny", 13, SHARP),
            "sharp_id": ("tiny-id", 14, SHARP)}


def fixture_model(tag):
    """(config, state dict, talker weights, predictor weights, fixture arrays) of a golden model family."""
    size, seed, kw = FIXTURES[tag]
    return (*golden_weights(size, seed, **kw), load_npz(f"model_{tag}.npz"))


def golden_weights(size, seed, **kw):
    """The same seeded synthetic checkpoint oracle/gen_golden.py loaded into the reference."""
    cfg = synthetic.make_config(size)
    sd = synthetic.make_state_dict(cfg, seed=seed, norm_jitter=0.1, with_text=True, embed_std=0.05, **kw)
    tw, pw = synthetic.split_state_dict(sd)
    return cfg, sd, tw, pw


def weight_checksum(sd):
    return sum(float(v.float().abs().sum()) for v in sd.values())


def oracle_cfgs(cfg):
    import qtts_oracle as O
    t = cfg["talker_config"]
    p = t["code_predictor_config"]
    tc = O.DecoderCfg(t["hidden_size"], t["num_hidden_layers"], t["num_attention_heads"], t["num_key_value_heads"],
                      t["head_dim"], t["intermediate_size"], t["vocab_size"], t["rms_norm_eps"], t["rope_theta"])
    pc = O.DecoderCfg(p["hidden_size"], p["num_hidden_layers"], p["num_attention_heads"], p["num_key_value_heads"],
                      p["head_dim"], p["intermediate_size"], p["vocab_size"], p["rms_norm_eps"], p["rope_theta"],
                      talker_hidden_size=t["hidden_size"], num_code_groups=p["num_code_groups"])
    return tc, pc


# ---------------------------------------------------------------- pick-by-pick id parity
def ulp_distance(got, want):
    """|got - want| in bf16 steps at the size of the largest |want| of each row (last axis)."""
    import qtts_oracle as O
    got, want = torch.as_tensor(got).float().cpu(), torch.as_tensor(want).float().cpu()
    return (got - want).abs() / O.bf16_ulp(want.abs().amax(dim=-1, keepdim=True))


def check_frames(got, tw, tc, pw, pc, prompts, text_adds, what, tol_ulps=2.0, min_binding=0.9, fixture_frames=None,
                 max_unacceptable=0, oracle_result=None, fixture_tag=None, noise_result=None):
    """got [S,NF,16]: ids a HIP run produced.  The oracle replays the run with those ids fed on (teacher forcing), so
    every one of the S*NF*16 picks is judged against the oracle's logits of the same history: it must be the arg-max,
    or within tol_ulps bf16 steps of it (helpers of qtts_oracle.judge_picks).  No pick is skipped and an early
    near-tie hides nothing that follows.  Asserts that at least min_binding of the picks had a clear winner (only the
    arg-max acceptable).
    fixture_frames: the ids the reference's own modules produced for these requests (tests/golden/model_*.npz).  Up to
    a stream's first pick that differs from them the HIP run and the reference share their history, so that pick must
    be a non-binding one (a binding pick has one acceptable id, the reference's); with fixture_tag, the number of
    frames that differ may not exceed what tests/golden/fixture_frames_floor.json records for that family.
    noise_result: the oracle's result for the same history with its GEMM sums re-ordered
    (qtts_oracle.other_summation_order).  Twenty-eight layers deep two summation orders of the same arithmetic put a
    few picks farther apart than tol_ulps; the picks the re-ordered oracle itself would have made are judged the same
    way, and twice the number of them that fail is allowed on top of max_unacceptable.
    Returns the oracle's result dict (logits of every pick) for further comparisons."""
    import json
    import qtts_oracle as O
    got = torch.as_tensor(got).long().cpu()
    S, NF = got.shape[:2]
    r = oracle_result or O.frame_loop_logits(tw, tc, pw, pc, prompts, text_adds, NF, forced=got)
    a0, b0, e0 = O.judge_picks(got[..., 0], r["t_logits"], tol_ulps)
    a1, b1, e1 = O.judge_picks(got[..., 1:], r["p_logits"], tol_ulps)
    acc = torch.cat([a0.unsqueeze(-1), a1], -1)
    bind = torch.cat([b0.unsqueeze(-1), b1], -1)
    same = torch.cat([e0.unsqueeze(-1), e1], -1)
    n = acc.numel()
    line = (f"[id parity] {what}: {n} picks, {int(bind.sum())} binding ({100.0 * int(bind.sum()) / n:.1f} %), "
            f"{int((~same).sum())} differ from the oracle's arg-max, {int((~acc).sum())} unacceptable")
    floor_bad = 0
    if noise_result is not None:
        n0 = O.judge_picks(noise_result["t_logits"].float().argmax(-1), r["t_logits"], tol_ulps)[0]
        n1 = O.judge_picks(noise_result["p_logits"].float().argmax(-1), r["p_logits"], tol_ulps)[0]
        floor_bad = int((~n0).sum()) + int((~n1).sum())
        line += f" (the oracle with re-ordered sums: {floor_bad})"
    differing = None
    if fixture_frames is not None:
        fx = torch.as_tensor(np.asarray(fixture_frames)).long()
        differing = int((~(got == fx).all(-1)).sum())
        line += f"; {S * NF - differing} of {S * NF} frames equal the reference fixture"
    print(line)
    bad = (~acc).nonzero().tolist()
    if bad:                                  # how far off: the chosen id's logit below the oracle's largest, in bf16 steps
        for (i, f, c) in bad[:8]:
            row = r["t_logits"][i, f] if c == 0 else r["p_logits"][i, f, c - 1]
            top = float(row.max())
            print(f"  pick (stream {i}, frame {f}, code {c}): id {int(got[i, f, c])} has logit {float(row[got[i, f, c]]):.4f}, "
                  f"the oracle's largest is {top:.4f} (id {int(row.argmax())}): {(top - float(row[got[i, f, c]])) / float(O.bf16_ulp(torch.tensor(top))):.1f} bf16 steps")
    assert len(bad) <= max_unacceptable + 2 * floor_bad, f"{what}: picks {bad[:8]} are not the oracle's arg-max nor within {tol_ulps} bf16 steps of it"
    assert int(bind.sum()) >= min_binding * n, f"{what}: only {int(bind.sum())} of {n} picks were binding"
    if fixture_frames is not None:
        for i in range(S):
            d = (got[i] != fx[i]).reshape(-1).nonzero().flatten()
            if len(d):                           # same history up to here: only a near-tie may be decided differently
                f, c = divmod(int(d[0]), got.shape[-1])
                assert not bool(bind[i, f, c]), (f"{what}: stream {i} leaves the reference's ids at frame {f}, code {c} "
                                                  f"({int(got[i, f, c])} instead of {int(fx[i, f, c])}), a pick with a clear winner")
        if fixture_tag is not None:
            allowed = json.load(open(os.path.join(GOLDEN, "fixture_frames_floor.json")))[fixture_tag]
            assert differing <= allowed, f"{what}: {differing} frames differ from the reference fixture, {allowed} recorded for {fixture_tag}"
    return r
