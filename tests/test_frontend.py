"""Prompt construction (text / speaker / language / voice-clone material -> talker input embeddings) against
fixtures written by the reference's own prepare_inputs, generate_icl_prompt and prepare_custom_voice_prompt
(oracle/gen_golden.py::gen_frontend; P/utils/generation.py, P/utils/prompt.py).  fp32 gathers and one small
MLP: the results must be bit-identical."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_npz
from nano_qwen3tts_vllm.utils import generation as G
from nano_qwen3tts_vllm.utils import prompt as PR
from nano_qwen3tts_vllm.utils import synthetic
from nano_qwen3tts_vllm.utils.embedding_loader import load_embeddings_only


class CharProcessor:
    """The fixtures' stand-in tokenizer: chat markers are single ids, other characters fold into the vocabulary."""
    MARKS = {"<|im_start|>": 1, "<|im_end|>": 2}

    def __init__(self, vocab):
        self.vocab = vocab

    def __call__(self, text=None, return_tensors="pt", padding=True):
        text = text[0] if isinstance(text, list) else text
        ids, i = [], 0
        while i < len(text):
            for m, v in self.MARKS.items():
                if text.startswith(m, i):
                    ids.append(v)
                    i += len(m)
                    break
            else:
                ids.append(3 + ord(text[i]) % (self.vocab - 10))
                i += 1
        return {"input_ids": torch.tensor([ids], dtype=torch.long)}


@pytest.fixture(scope="module")
def fe(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("frontend_model"))
    meta = json.load(open(os.path.join(GOLDEN, "frontend.json")))
    synthetic.write_model_dir(d, "tiny", seed=meta["seed"], with_text=True, embed_std=0.05, std=0.2)
    cfg, text_emb, in_emb, proj, preds = load_embeddings_only(d, "cpu")
    return dict(cfg=cfg, text=text_emb, codec=in_emb, proj=proj, preds=preds, g=load_npz("frontend.npz"), meta=meta,
                processor=CharProcessor(cfg.talker_config.text_vocab_size), dir=d)


def same(got, g, name):
    emb, tail, pad, mask = got
    assert np.array_equal(emb.numpy(), g[name + "_embeds"]), name
    assert np.array_equal(tail.numpy(), g[name + "_tail"]), name
    assert np.array_equal(pad.numpy(), g[name + "_pad"]), name
    assert np.array_equal(mask.numpy(), g[name + "_mask"]), name


def common(fe):
    return dict(text_embedding=fe["text"], input_embedding=fe["codec"], text_projection=fe["proj"], device="cpu")


def test_voice_design_and_tokenisation(fe):
    g = fe["g"]
    for case in [c for c in fe["meta"]["cases"] if c["kind"] == "design"]:
        ids, ins, spk, langs = PR.prepare_custom_voice_prompt(text=[case["text"]], speaker=[""], language=[case["language"]],
                                                              instruct=[case["instruct"]], processor=fe["processor"])
        assert np.array_equal(ids[0].numpy(), g[case["name"] + "_ids"]) and np.array_equal(ins[0].numpy(), g[case["name"] + "_ins"])
        assert spk == [""] and langs == [case["language"]]
        same(G.prepare_inputs(fe["cfg"], ids, instruct_ids=ins, languages=langs, speakers=None,
                              non_streaming_mode=case["non_streaming"], **common(fe)), g, case["name"])


def test_custom_voice_batch_with_dialect_speaker(fe):
    g = fe["g"]
    for case in [c for c in fe["meta"]["cases"] if c["kind"] == "custom"]:
        ids, ins, spk, langs = PR.prepare_custom_voice_prompt(text=case["texts"], speaker=case["speakers"],
                                                              language=case["languages"], processor=fe["processor"])
        assert ins == [None, None]
        assert np.array_equal(ids[0].numpy(), g["custom_ids0"]) and np.array_equal(ids[1].numpy(), g["custom_ids1"])
        got = G.prepare_inputs(fe["cfg"], ids, instruct_ids=ins, speakers=spk, languages=langs,
                               non_streaming_mode=case["non_streaming"], **common(fe))
        same(got, g, case["name"])
        assert got[3].sum(1).tolist() == [int(m.sum()) for m in g[case["name"] + "_mask"]]      # ragged batch, left padding
    with pytest.raises(NotImplementedError):
        G.prepare_inputs(fe["cfg"], ids, speakers=["nobody", "eric"], languages=langs, **common(fe))
    with pytest.raises(NotImplementedError):
        G.prepare_inputs(fe["cfg"], ids, speakers=spk, languages=["klingon", "english"], **common(fe))
    with pytest.raises(ValueError):
        PR.prepare_custom_voice_prompt(text=["a", "b"], speaker=["x", "y", "z"], language="English", processor=fe["processor"])


def test_voice_clone_xvector_and_in_context(fe):
    g, dev = fe["g"], torch.device("cpu")
    ids = [torch.from_numpy(g["clone_ids"])]
    ref_ids = [torch.from_numpy(g["clone_ref_ids"])]
    spk = torch.from_numpy(g["clone_spk"])
    assert np.array_equal(PR._tokenize_texts([PR._build_ref_text("Reference words.")], fe["processor"], dev)[0].numpy(),
                          g["clone_ref_ids"])

    def icl_fn(**kw):
        return G.generate_icl_prompt(config=fe["cfg"], text_embedding=fe["text"], input_embedding=fe["codec"],
                                     text_projection=fe["proj"], code_predictor_embeddings=fe["preds"], device=dev, **kw)
    spk_fn = lambda p: G.generate_speaker_prompt(p, dev)
    vc = {"ref_code": [None], "ref_spk_embedding": [spk], "x_vector_only_mode": [True], "icl_mode": [False]}
    same(G.prepare_inputs(fe["cfg"], ids, voice_clone_prompt=vc, languages=["Auto"], non_streaming_mode=True,
                          generate_speaker_prompt_fn=spk_fn, generate_icl_prompt_fn=icl_fn, **common(fe)), g, "clone_xvec")
    for case in [c for c in fe["meta"]["cases"] if c["kind"] == "clone_icl"]:
        vc = {"ref_code": [torch.from_numpy(g[case["name"] + "_code"])], "ref_spk_embedding": [spk],
              "x_vector_only_mode": [False], "icl_mode": [True]}
        same(G.prepare_inputs(fe["cfg"], ids, ref_ids=ref_ids, voice_clone_prompt=vc, languages=["English"],
                              non_streaming_mode=case["non_streaming"], generate_speaker_prompt_fn=spk_fn,
                              generate_icl_prompt_fn=icl_fn, **common(fe)), g, case["name"])
    with pytest.raises(ValueError):
        G.prepare_inputs(fe["cfg"], ids, voice_clone_prompt=vc, languages=["English"], **common(fe))       # no speaker source
    with pytest.raises(ValueError):
        G.prepare_inputs(fe["cfg"], ids, voice_clone_prompt=vc, languages=["English"], generate_speaker_prompt_fn=spk_fn,
                         generate_icl_prompt_fn=icl_fn, **common(fe))                                       # ICL without ref_ids


def test_interface_front_end_builds_the_same_prompts(fe):
    """The API object's own preparation (what generate_voice_design / generate_custom_voice /
    generate_voice_clone feed to the engine) on the fixture requests."""
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    g = fe["g"]
    iface = Qwen3TTSInterface(fe["dir"])
    assert iface.processor is None
    with pytest.raises(RuntimeError, match="no tokenizer"):
        iface.generate_custom_voice("hi")
    iface.processor = fe["processor"]
    same(iface._prep_custom_voice("Hello there, world.", "Auto", "", "A calm low voice", True), g, "design_ns")
    same(iface._prep_custom_voice("Hello there, world.", "English", "", "A calm low voice", False), g, "design_stream")
    one = iface._prep_custom_voice("Good morning.", "English", "Vivian")
    n = int(g["custom_mask"][0].sum())
    assert np.array_equal(one[0][0].numpy(), g["custom_embeds"][0, -n:])            # the batch fixture's first row, unpadded
    spk = torch.from_numpy(g["clone_spk"])
    vc = {"ref_code": None, "ref_spk_embedding": spk, "x_vector_only_mode": True, "icl_mode": False}
    same(iface._prep_voice_clone("Clone this voice please.", None, None, None, False, vc, True), g, "clone_xvec")
    vc = {"ref_code": torch.from_numpy(g["clone_icl_short_code"]), "ref_spk_embedding": spk, "x_vector_only_mode": False,
          "icl_mode": True, "ref_text": "Reference words."}
    same(iface._prep_voice_clone("Clone this voice please.", "English", None, None, False, vc, False), g, "clone_icl_short")
    with pytest.raises(ValueError):
        iface._prep_voice_clone("x", "English", None, None, False, dict(vc, ref_text=None), False)
    with pytest.raises(RuntimeError, match="speech_tokenizer not available"):
        iface._prep_voice_clone("x", "English", "ref.wav", "words", False, None, True)


@pytest.mark.gpu
def test_interface_front_end_on_the_gpu_matches_the_reference_fixtures(fe):
    """On a GPU box the API object builds its prompts on the device (fp32 gathers + the text-projection MLP through
    torch's fp32 GEMM): the same fixture requests, compared with the reference's CPU results.  Gathered rows are
    exact; rows that went through the MLP may differ by the GEMM's summation order — tolerance 1e-5 (absolute and
    relative) on O(0.1) values; masks and lengths exact."""
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    g = fe["g"]
    iface = Qwen3TTSInterface(fe["dir"])
    assert iface.device.type == "cuda"
    iface.processor = fe["processor"]

    def close(got, name):
        emb, tail, pad, mask = [t.cpu() for t in got]
        assert emb.dtype == torch.float32
        for t, key in ((emb, "_embeds"), (tail, "_tail"), (pad, "_pad")):
            want = g[name + key]
            assert t.shape == want.shape, (name, key)
            assert np.allclose(t.numpy(), want, atol=1e-5, rtol=1e-5), (name, key, float(np.abs(t.numpy() - want).max()))
        assert np.array_equal(mask.numpy(), g[name + "_mask"]), name

    close(iface._prep_custom_voice("Hello there, world.", "Auto", "", "A calm low voice", True), "design_ns")
    close(iface._prep_custom_voice("Hello there, world.", "English", "", "A calm low voice", False), "design_stream")
    spk = torch.from_numpy(g["clone_spk"])
    vc = {"ref_code": None, "ref_spk_embedding": spk, "x_vector_only_mode": True, "icl_mode": False}
    close(iface._prep_voice_clone("Clone this voice please.", None, None, None, False, vc, True), "clone_xvec")
    vc = {"ref_code": torch.from_numpy(g["clone_icl_short_code"]), "ref_spk_embedding": spk, "x_vector_only_mode": False,
          "icl_mode": True, "ref_text": "Reference words."}
    close(iface._prep_voice_clone("Clone this voice please.", "English", None, None, False, vc, False), "clone_icl_short")
    iface.shutdown()
