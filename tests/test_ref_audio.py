"""Reference clip -> voice-clone prompt (utils/ref_audio.py, Qwen3TTSInterface.create_voice_clone_prompt;
P/interface.py:483-721).  The reference takes the mel filterbank, the resampler and the audio decoder from librosa /
soundfile and the two networks (speech codec encoder, speaker encoder) from qwen_tts; none of them is installed here
or part of the reference's sources, so against THOSE the results are parity-unpinned.  What is pinned:
  * the filterbank against an independent implementation of the same published definition (transformers.audio_utils),
  * the spectrogram against torch.stft called the way the reference calls it,
  * the prompt the API builds from a clip, bit for bit against the reference-generated front-end fixture, with
    stand-in networks that return the fixture's codes and speaker vector."""
import base64
import io
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_npz
from nano_qwen3tts_vllm.utils import ref_audio as RA
from nano_qwen3tts_vllm.utils import synthetic
from test_frontend import CharProcessor, same


def tone(freq, sr, seconds=1.0, amp=0.5):
    t = np.arange(int(sr * seconds)) / sr
    return (amp * np.sin(2 * np.pi * freq * t)).astype(np.float32)


def test_mel_scale():
    assert float(RA.hz_to_mel(0.0)) == 0.0 and abs(float(RA.hz_to_mel(1000.0)) - 15.0) < 1e-12
    assert abs(float(RA.hz_to_mel(6400.0)) - 42.0) < 1e-12                    # 27 log steps per factor 6.4
    f = np.array([0.0, 50.0, 999.0, 1000.0, 1001.0, 4000.0, 12000.0])
    assert np.allclose(RA.mel_to_hz(RA.hz_to_mel(f)), f, rtol=1e-12, atol=1e-9)


def test_filterbank_against_an_independent_implementation():
    au = pytest.importorskip("transformers.audio_utils")
    for sr, n_fft, n_mels, fmin, fmax in [(24000, 1024, 128, 0, 12000), (16000, 400, 80, 0, 8000), (24000, 1024, 40, 60, 7600)]:
        mine = RA.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
        other = au.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin,
                                   max_frequency=fmax, sampling_rate=sr, norm="slaney", mel_scale="slaney").T
        assert mine.dtype == np.float32 and mine.shape == (n_mels, n_fft // 2 + 1)
        assert np.allclose(mine, other, rtol=1e-5, atol=1e-9), float(np.abs(mine - other).max())


def test_filterbank_shape_properties():
    w = RA.mel_filterbank(24000, 1024, 128, 0, 12000)
    assert (w >= 0).all() and (w.sum(1) > 0).all()                             # every band sees at least one bin
    peaks = w.argmax(1)
    assert (np.diff(peaks) >= 0).all() and peaks[0] <= 2 and peaks[-1] >= 480
    area = w.astype(np.float64).sum(1) * (24000 / 1024)                        # unit-area triangles (sampled on the bins)
    assert np.allclose(area[40:], 1.0, atol=0.08), (area.min(), area.max())


def test_spectrogram_against_stft():
    g = torch.Generator().manual_seed(0)
    y = torch.randn(2, 24000 + 300, generator=g) * 0.1
    got = RA.mel_spectrogram(y, **RA.MEL_ARGS)
    pad = (1024 - 256) // 2
    yp = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(yp, 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024), center=False,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    mag = torch.sqrt(torch.view_as_real(spec).pow(2).sum(-1) + 1e-9)
    want = torch.matmul(torch.from_numpy(RA.mel_filterbank(24000, 1024, 128, 0, 12000)), mag)
    assert got.shape == want.shape == (2, 128, (24300 + 2 * pad - 1024) // 256 + 1)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-6), float((got - want).abs().max())


def test_spectrogram_of_a_pure_tone_peaks_in_its_band():
    for freq in (300.0, 3000.0, 9000.0):
        m = RA.mel_spectrogram(torch.from_numpy(tone(freq, 24000)).unsqueeze(0), **RA.MEL_ARGS)[0].mean(1)
        centres = RA.mel_to_hz(np.linspace(RA.hz_to_mel(0), RA.hz_to_mel(12000), 130))[1:-1]
        band = int(m.argmax())
        assert abs(centres[band] - freq) <= max(centres[band + 1] - centres[band - 1], 24000 / 1024), (freq, centres[band])


def wav_bytes(data, sr):
    from scipy.io import wavfile
    buf = io.BytesIO()
    wavfile.write(buf, sr, data)
    return buf.getvalue()


def test_loading_and_input_forms(tmp_path):
    pcm = (tone(440, 16000, 0.25) * 32767).astype(np.int16)
    path = str(tmp_path / "clip.wav")
    open(path, "wb").write(wav_bytes(pcm, 16000))
    wav, sr = RA.load_audio(path)
    assert sr == 16000 and wav.dtype == np.float32 and np.allclose(wav, pcm / 32768.0, atol=1e-7)
    b64 = base64.b64encode(wav_bytes(pcm, 16000)).decode()
    for form in (b64, "data:audio/wav;base64," + b64):
        w2, sr2 = RA.load_audio(form)
        assert sr2 == 16000 and np.array_equal(w2, wav)
    stereo = np.stack([pcm, np.zeros_like(pcm)], axis=1)
    open(path, "wb").write(wav_bytes(stereo, 16000))
    assert np.allclose(RA.load_audio(path)[0], wav / 2, atol=1e-7)                           # channels averaged
    flt = tone(440, 24000, 0.1)
    open(path, "wb").write(wav_bytes(flt, 24000))
    assert np.array_equal(RA.load_audio(path)[0], flt)
    out = RA.normalize_audio_inputs([(torch.from_numpy(stereo.astype(np.float32)), 8000), (flt, 24000), path])
    assert [o[1] for o in out] == [8000, 24000, 24000] and out[0][0].ndim == 1 and out[0][0].dtype == np.float32
    with pytest.raises(ValueError, match="without sampling rate"):
        RA.normalize_audio_inputs(flt)
    with pytest.raises(ValueError, match="Unsupported"):
        RA.normalize_audio_inputs([3.5])


def test_resample_keeps_the_tone():
    x = tone(440, 16000, 1.0)
    y = RA.resample(x, 16000, 24000)
    assert y.dtype == np.float32 and abs(len(y) - 24000) <= 1
    assert RA.resample(x, 24000, 24000) is x
    spec = np.abs(np.fft.rfft(y[:24000] * np.hanning(len(y[:24000]))))
    assert abs(int(spec.argmax()) - 440) <= 1
    assert abs(np.sqrt((y[2000:-2000] ** 2).mean()) - 0.5 / np.sqrt(2)) < 5e-3               # level preserved


class FixtureCodec:
    """Stand-in for interface.speech_tokenizer: `tokenizer.encode(wav, sr=)` in qwen_tts's call form, returning the
    fixture's reference codes whatever the clip."""
    def __init__(self, codes):
        self.calls = []
        self.tokenizer = SimpleNamespace(encode=self._encode)
        self.codes = codes

    def _encode(self, wav, sr=None):
        self.calls.append((wav.shape, wav.dtype, sr))
        return SimpleNamespace(audio_codes=[self.codes])


class FixtureSpeaker(torch.nn.Module):
    """Stand-in speaker encoder: a bf16 parameter (the interface must feed mels in the module's dtype) and the
    fixture's speaker vector as output."""
    def __init__(self, vec):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1, dtype=torch.bfloat16))
        self.vec = vec
        self.seen = []

    def forward(self, mels):
        self.seen.append((tuple(mels.shape), mels.dtype, float(mels.float().mean())))
        return self.vec.unsqueeze(0)


def test_clip_to_prompt_to_the_reference_fixture(tmp_path):
    from nano_qwen3tts_vllm.interface import Qwen3TTSInterface
    meta = json.load(open(os.path.join(GOLDEN, "frontend.json")))
    d = str(tmp_path / "model")
    os.makedirs(d)
    synthetic.write_model_dir(d, "tiny", seed=meta["seed"], with_text=True, embed_std=0.05, std=0.2)
    g = load_npz("frontend.npz")
    iface = Qwen3TTSInterface(d)
    iface.processor = CharProcessor(iface.model_config.talker_config.text_vocab_size)
    clip = (tone(220, 16000, 0.5), 16000)
    with pytest.raises(RuntimeError, match="speech_tokenizer not available"):
        iface.create_voice_clone_prompt(clip, ref_text="Reference words.")
    codec = FixtureCodec(torch.from_numpy(g["clone_icl_short_code"]))
    iface.speech_tokenizer = codec
    with pytest.raises(ValueError, match="ref_text is required"):
        iface.create_voice_clone_prompt(clip)
    with pytest.raises(RuntimeError, match="Speaker encoder not available"):
        iface.create_voice_clone_prompt(clip, ref_text="Reference words.")
    spk = FixtureSpeaker(torch.from_numpy(g["clone_spk"]).to(iface.device))
    iface.speaker_encoder = spk
    vc = iface.create_voice_clone_prompt(clip, ref_text="Reference words.")
    assert vc["icl_mode"] and not vc["x_vector_only_mode"] and vc["ref_text"] == "Reference words."
    assert np.array_equal(vc["ref_code"].numpy(), g["clone_icl_short_code"])
    assert codec.calls[-1] == ((8000,), np.float32, 16000)                                   # the codec sees the clip as given
    shape, dtype, mean = spk.seen[-1]
    assert shape == (1, 12000 // 256, 128) and dtype == torch.bfloat16 and mean > 0           # the encoder sees 24 kHz mels
    if iface.device.type == "cpu":
        same(iface._prep_voice_clone("Clone this voice please.", "English", None, None, False, vc, False), g, "clone_icl_short")
        # ... and straight from the clip, as generate_voice_clone(ref_audio=...) does
        same(iface._prep_voice_clone("Clone this voice please.", "English", clip, "Reference words.", False, None, False),
             g, "clone_icl_short")
    xv = iface.create_voice_clone_prompt(clip, x_vector_only_mode=True)
    assert xv["ref_code"] is None and xv["x_vector_only_mode"] and not xv["icl_mode"]
    if iface.device.type == "cpu":
        same(iface._prep_voice_clone("Clone this voice please.", None, None, None, False, xv, True), g, "clone_xvec")
