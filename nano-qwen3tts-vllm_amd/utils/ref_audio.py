"""Reference audio for voice cloning: loading, resampling and the 128-band mel spectrogram the speaker encoder
takes (the roles of P/interface.py:483-556 and :591-640).

The reference leans on three packages that are not part of it: `soundfile` / `librosa.load` (decoding),
`librosa.resample` (soxr) and `librosa.filters.mel`.  Here:
  * decoding uses `soundfile` when it is installed and scipy's WAV reader otherwise (PCM 8/16/32 bit and float WAV);
  * resampling uses `soxr` when it is installed and a polyphase filter (scipy.signal.resample_poly) otherwise —
    not sample-identical to soxr;
  * the mel filterbank is computed here from librosa's published definition (Slaney's Auditory-Toolbox scale:
    linear below 1 kHz at 200/3 Hz per mel, logarithmic above with 27 steps per factor 6.4; triangular filters
    normalised to unit area), in the same floating-point order (float64 ramps, float32 weights).
The spectrogram itself is plain arithmetic and is stated in full: reflect padding by (n_fft - hop) / 2, frames of
n_fft samples every hop, periodic Hann window, magnitude sqrt(re^2 + im^2 + 1e-9), filterbank product.
It runs once per reference clip on the host; nothing here is on the per-frame path.
"""
from __future__ import annotations

import base64
import io
import math
import urllib.request
from urllib.parse import urlparse

import numpy as np
import torch

SPEAKER_SR = 24000                                   # the speaker encoder's sample rate (P/interface.py:569)
MEL_ARGS = dict(n_fft=1024, num_mels=128, sampling_rate=SPEAKER_SR, hop_size=256, win_size=1024, fmin=0, fmax=12000)


# ---------------------------------------------------------------- loading
def _is_url(x: str) -> bool:
    try:
        u = urlparse(x)
    except ValueError:
        return False
    return bool(u.scheme and u.netloc)


def _looks_like_base64(x: str) -> bool:
    """Long strings that decode as base64 (optionally behind a `data:...,` prefix) are taken as audio bytes."""
    if len(x) <= 100:
        return False
    try:
        base64.b64decode(x.rsplit(",", 1)[-1])
        return True
    except Exception:
        return False


def _decode_wav(f) -> tuple[np.ndarray, int]:
    try:
        import soundfile as sf
        audio, sr = sf.read(f, dtype="float32", always_2d=False)
        return audio, int(sr)
    except ImportError:
        pass
    from scipy.io import wavfile
    sr, data = wavfile.read(f)
    if data.dtype == np.int16:
        audio = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        audio = (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif data.dtype == np.uint8:
        audio = (data.astype(np.float32) - 128.0) / 128.0
    else:
        audio = data.astype(np.float32)
    return audio, int(sr)


def load_audio(x: str) -> tuple[np.ndarray, int]:
    """A path, an http(s) URL or a base64 string -> (mono float32 waveform, sample rate)."""
    if _is_url(x):
        with urllib.request.urlopen(x) as resp:
            audio, sr = _decode_wav(io.BytesIO(resp.read()))
    elif _looks_like_base64(x):
        payload = x.split(",", 1)[1] if x.strip().startswith("data:") and "," in x else x
        audio, sr = _decode_wav(io.BytesIO(base64.b64decode(payload)))
    else:
        audio, sr = _decode_wav(x)
    if audio.ndim > 1:
        audio = audio.mean(axis=-1)
    return audio.astype(np.float32), sr


def normalize_audio_inputs(audios) -> list[tuple[np.ndarray, int]]:
    """One item or a list of: str (path / URL / base64) or (waveform, sample_rate) with a numpy or torch
    waveform, mono or [samples, channels].  A bare array has no sample rate and is refused (P/interface.py:526-556)."""
    out = []
    for item in audios if isinstance(audios, list) else [audios]:
        if isinstance(item, str):
            out.append(load_audio(item))
        elif isinstance(item, tuple) and len(item) == 2:
            wav, sr = item
            if isinstance(wav, torch.Tensor):
                wav = wav.detach().cpu().numpy()
            wav = np.asarray(wav)
            if wav.ndim > 1:
                wav = wav.mean(axis=-1)
            out.append((wav.astype(np.float32), int(sr)))
        elif isinstance(item, np.ndarray):
            raise ValueError("numpy array provided without sampling rate. Use (np.ndarray, sr) tuple.")
        else:
            raise ValueError(f"Unsupported audio input type: {type(item)}")
    return out


def resample(wav: np.ndarray, sr: int, target_sr: int = SPEAKER_SR) -> np.ndarray:
    if sr == target_sr:
        return wav
    try:
        import soxr
        return soxr.resample(wav.astype(np.float32), sr, target_sr, quality="HQ").astype(np.float32)
    except ImportError:
        from scipy.signal import resample_poly
        g = math.gcd(int(sr), int(target_sr))
        return resample_poly(wav.astype(np.float64), target_sr // g, sr // g).astype(np.float32)


# ---------------------------------------------------------------- mel spectrogram
_F_SP = 200.0 / 3.0
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = math.log(6.4) / 27.0


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= _MIN_LOG_HZ, _MIN_LOG_MEL + np.log(np.maximum(f, _MIN_LOG_HZ) / _MIN_LOG_HZ) / _LOGSTEP, f / _F_SP)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= _MIN_LOG_MEL, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), _F_SP * m)


def mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax: float | None = None) -> np.ndarray:
    """float32 [n_mels, n_fft // 2 + 1]: triangles between n_mels + 2 points equally spaced on the mel scale,
    each scaled by 2 / (its upper edge - its lower edge in Hz)."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    bins = np.fft.rfftfreq(n_fft, 1.0 / sr)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - bins[None, :]
    w = np.zeros((n_mels, bins.shape[0]), dtype=np.float32)
    for i in range(n_mels):
        rising = -ramps[i] / width[i]
        falling = ramps[i + 2] / width[i + 1]
        w[i] = np.maximum(0.0, np.minimum(rising, falling))
    area = 2.0 / (edges[2:n_mels + 2] - edges[:n_mels])
    return (w.astype(np.float64) * area[:, None]).astype(np.float32)


def mel_spectrogram(y: torch.Tensor, n_fft: int, num_mels: int, sampling_rate: int, hop_size: int, win_size: int,
                    fmin: float, fmax: float | None = None) -> torch.Tensor:
    """y [batch, samples] float32 -> [batch, num_mels, frames] linear-magnitude mel spectrogram (no log), frames =
    samples // hop_size for clips that are a multiple of the hop (P/interface.py:591-640, which calls torch.stft
    with center=False on the padded clip: the same frames, window and transform)."""
    assert win_size == n_fft, "window shorter than the transform is not used by the speaker front-end"
    basis = torch.from_numpy(mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)).to(y.device)
    pad = (n_fft - hop_size) // 2
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    frames = y.unfold(-1, n_fft, hop_size) * torch.hann_window(win_size, device=y.device)        # [batch, frames, n_fft]
    spec = torch.fft.rfft(frames, dim=-1)
    mag = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-9).transpose(1, 2)                  # [batch, bins, frames]
    return torch.matmul(basis, mag)
