"""Codec-decode wrapper (SpeechTokenizerCUDAGraph) driven with a stand-in causal decoder.

The real codec network lives in the absent `qwen_tts` package, so nothing here pins codec arithmetic; what
is checked is the wrapper's own behaviour: windowing/trim index arithmetic (CPU) and graph replay == eager
for every captured length, eager fall-through for the others (GPU)."""
import numpy as np
import pytest
import torch

from nano_qwen3tts_vllm.utils import speech_tokenizer_cudagraph as stc

UP = 8          # samples per frame of the stand-in
RF = 4          # frames of left context the stand-in looks at
VOCAB = 128     # rows of the stand-in's code tables: ids handed to it must be below this


class CausalDecoder(torch.nn.Module):
    """codes [B,16,T] -> wav [B,1,T*UP]; frame t depends on frames t-RF..t only."""
    total_upsample = UP

    def __init__(self, vocab=VOCAB, width=32):
        super().__init__()
        g = torch.Generator().manual_seed(7)
        self.emb = torch.nn.Parameter(torch.randn(16, vocab, width, generator=g) * 0.3)
        self.conv = torch.nn.Conv1d(width, width, RF + 1)
        self.out = torch.nn.Conv1d(width, UP, 1)
        for p in list(self.conv.parameters()) + list(self.out.parameters()):
            p.data = torch.randn(p.shape, generator=g) * 0.2

    def forward(self, codes):
        B, Q, T = codes.shape
        x = sum(self.emb[q][codes[:, q]] for q in range(Q))              # [B,T,W]
        x = torch.nn.functional.pad(x.transpose(1, 2), (RF, 0))          # causal
        y = self.out(torch.tanh(self.conv(x)))                           # [B,UP,T]
        return y.transpose(1, 2).reshape(B, 1, T * UP)


class StubTokenizer:
    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.decoder = CausalDecoder()

        def get_output_sample_rate(self):
            return 24000

    def __init__(self):
        self.model = self._M()

    def decode(self, inputs):
        dev = next(self.model.parameters()).device
        out = []
        for r in inputs:
            c = torch.as_tensor(r["audio_codes"], dtype=torch.long, device=dev)
            wav = self.model.decoder(c.transpose(0, 1).unsqueeze(0))
            out.append(wav.reshape(-1).float().cpu().numpy())
        return out, 24000


def _frames(T, seed=0):
    return torch.randint(0, VOCAB, (T, 16), generator=torch.Generator().manual_seed(seed))


def test_missing_codec_package_is_an_import_error():
    if stc.HAS_SPEECH_TOKENIZER:
        pytest.skip("qwen_tts is installed")
    with pytest.raises(ImportError):
        stc.SpeechTokenizerCUDAGraph("/nonexistent", device="cpu")


def test_chunked_decode_equals_full_decode_for_a_causal_decoder():
    tok = stc.SpeechTokenizerCUDAGraph(tokenizer=StubTokenizer(), device="cpu", dtype=torch.float32, num_graph_lengths=0)
    for T, chunk, left in [(37, 7, RF), (37, 300, 25), (50, 10, 25), (1, 300, 25), (21, 7, 7)]:
        fr = _frames(T, T)
        full, sr = tok.decode([{"audio_codes": fr}])
        got, sr2 = tok.chunked_decode([{"audio_codes": fr.tolist()}], chunk_size=chunk, left_context_size=left)
        assert sr == sr2 == 24000 and got[0].shape == (T * UP,)
        np.testing.assert_allclose(got[0], full[0], atol=1e-6)
    # fewer context frames than the decoder looks back: the chunk seams differ, the first chunk does not
    fr = _frames(30, 3)
    full, _ = tok.decode([{"audio_codes": fr}])
    short, _ = tok.chunked_decode([{"audio_codes": fr}], chunk_size=10, left_context_size=1)
    np.testing.assert_allclose(short[0][:10 * UP], full[0][:10 * UP], atol=1e-6)
    assert np.abs(short[0][10 * UP:] - full[0][10 * UP:]).max() > 1e-4


def test_window_decode_returns_only_the_new_frames():
    tok = stc.SpeechTokenizerCUDAGraph(tokenizer=StubTokenizer(), device="cpu", dtype=torch.float32, num_graph_lengths=0)
    fr = _frames(40, 1)
    full, _ = tok.decode([{"audio_codes": fr}])
    batch = [{"audio_codes": fr[0:4].tolist(), "left_context_frames": 0},          # first chunk, no context
             {"audio_codes": fr[4 - RF:8].tolist(), "left_context_frames": RF},
             {"audio_codes": fr[12:36].tolist(), "left_context_frames": 20},      # window of 24 = 20 context + 4 new
             {"audio_codes": fr[30:40].tolist(), "left_context_frames": None},
             {"audio_codes": [], "left_context_frames": 0}]
    wavs, sr = tok.decode_window_batched(batch)
    assert sr == 24000 and [len(w) for w in wavs] == [4 * UP, 4 * UP, 4 * UP, 10 * UP, 0]
    np.testing.assert_allclose(wavs[0], full[0][:4 * UP], atol=1e-6)
    np.testing.assert_allclose(wavs[1], full[0][4 * UP:8 * UP], atol=1e-6)
    np.testing.assert_allclose(wavs[2], full[0][32 * UP:36 * UP], atol=1e-6)
    assert all(w.dtype == np.float32 for w in wavs)


def test_decode_codec_ids_batches():
    tok = stc.SpeechTokenizerCUDAGraph(tokenizer=StubTokenizer(), device="cpu", dtype=torch.float32, num_graph_lengths=0)
    ids = torch.stack([_frames(9, 5).T, _frames(9, 6).T])            # [2,16,9]
    wavs, sr = tok.decode_codec_ids(ids)
    ref, _ = tok.decode([{"audio_codes": _frames(9, 6)}])
    assert len(wavs) == 2 and sr == 24000
    np.testing.assert_array_equal(wavs[1], ref[0])
    assert tok.sample_rate == 12500          # neither config nor feature extractor on the stand-in (reference :95-100)


@pytest.mark.gpu
def test_graph_replay_equals_eager_for_every_captured_length():
    stub = StubTokenizer()
    stub.model.to("cuda:0")
    eager = stub.model.decoder.forward
    tok = stc.SpeechTokenizerCUDAGraph(tokenizer=stub, device="cuda:0", dtype=torch.float32, num_graph_lengths=12)
    assert sorted(tok.graphs.graphs) == list(range(1, 13))
    for T in [1, 2, 7, 12, 7, 13, 40]:                               # 13 and 40 are not captured: eager path
        codes = _frames(T, 100 + T).T.unsqueeze(0).cuda()
        want = eager(codes).clone()
        got = stub.model.decoder(codes)
        assert got.shape == (1, 1, T * UP)
        torch.testing.assert_close(got, want, atol=1e-6, rtol=0)
    # the tokenizer's own decode and the window decode go through the same graphs
    fr = _frames(12, 9)
    full, _ = tok.decode([{"audio_codes": fr}])
    win, _ = tok.decode_window_batched([{"audio_codes": fr.tolist(), "left_context_frames": 8}])
    np.testing.assert_allclose(win[0], full[0][8 * UP:], atol=1e-6)
    ch, _ = tok.chunked_decode([{"audio_codes": fr.tolist()}], chunk_size=4, left_context_size=RF)
    np.testing.assert_allclose(ch[0], full[0], atol=1e-5)
    tok.graphs.remove()
    assert stub.model.decoder.forward == eager
