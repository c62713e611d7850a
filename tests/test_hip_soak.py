"""GPU: the soak schedule of tools/soak.py on the fixture-sized model — random single / packed / beside-the-frames /
mid-frame admissions, prompts of 4 to 700 tokens, budgets of 1 to 300 frames, sampling requests, cancellations — with its
invariants (ids in range, budgets met exactly, every KV block returned) and, for a schedule that does not depend on
GPU timing, two runs on fresh engines that must agree id for id, sampled streams included."""
import importlib.util
import json
import os

import pytest
import torch

from helpers import fixture_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_soak_schedule_invariants_and_run_to_run_determinism(tmp_path, monkeypatch):
    spec = importlib.util.spec_from_file_location("qtts_soak", os.path.join(ROOT, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    monkeypatch.setenv("QTTS_SEED", "99")
    cfg, sd, tw, pw, g = fixture_model("sharp")
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    H = cfg["talker_config"]["hidden_size"]
    tv = cfg["talker_config"]["vocab_size"]
    pv = cfg["talker_config"]["code_predictor_config"]["vocab_size"]
    dev = torch.device("cuda", 0)
    args = (str(tmp_path), H, tv, pv, tw, pw, dev)
    finished, counts = soak.run_schedule(3, 600, *args, timing_free=False)
    assert counts["packed"] > 5 and counts["overlapped"] > 5 and counts["cancelled"] > 5 and counts["sampled"] > 10
    assert counts["in_flight"] > 2                           # some requests were admitted from the engine's hook, mid-frame
    a, ca = soak.run_schedule(21, 300, *args, timing_free=True)
    b, _ = soak.run_schedule(21, 300, *args, timing_free=True)
    assert sorted(a) == sorted(b) and len(a) > 20 and ca["sampled"] > 3 and ca["overlapped"] > 3     # (detached prefills replay too)
    assert [sid for sid in a if a[sid] != b[sid]] == []
