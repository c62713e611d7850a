"""The speech codec behind the API object's `speech_tokenizer` attribute (the role of P/utils/audio.py:35-174).

The codec itself (codes <-> 24 kHz PCM) is the third-party `qwen_tts` package's `Qwen3TTSTokenizer`
(`qwen-tts 0.0.5`, not vendored by the reference, not installed here); this module only adapts the shapes
applications hand over: `decode` takes what the reference's callers pass — a list of 16-id frames, a tensor
[batch, 16, time] or `[{"audio_codes": frames}]` (examples/quick_benchmark.py:66) — and returns
`(list of numpy waveforms, sample_rate)`; `encode` returns codes [batch, 16, time].  Without `qwen_tts` the
constructor raises ImportError, which the interface turns into `speech_tokenizer = None` as the reference does
(P/interface.py:421-440)."""
from __future__ import annotations

import numpy as np
import torch

DEFAULT_CODEC = "Qwen/Qwen3-TTS-Tokenizer-12Hz"        # the hub id the reference hard-codes (P/interface.py:426)


class SpeechTokenizer:
    def __init__(self, model_path: str = DEFAULT_CODEC, dtype=torch.bfloat16, device: str | None = None, tokenizer=None):
        """tokenizer: an already loaded codec object (anything with decode / encode in qwen_tts's call form)."""
        if tokenizer is None:
            try:
                from qwen_tts.inference.qwen3_tts_tokenizer import Qwen3TTSTokenizer
            except ImportError:
                try:
                    from qwen_tts import Qwen3TTSTokenizer
                except ImportError as e:
                    raise ImportError("the speech codec needs the `qwen_tts` package "
                                      "(pip install git+https://github.com/QwenLM/Qwen3-TTS.git)") from e
            device = device or ("cuda:0" if torch.cuda.is_available() else "cpu")
            tokenizer = Qwen3TTSTokenizer.from_pretrained(model_path, device_map=device)
        self.tokenizer = tokenizer
        self.dtype = dtype
        self.sample_rate = int(getattr(tokenizer, "sample_rate", None) or getattr(tokenizer, "sampling_rate", None) or 24000)

    @staticmethod
    def _as_batch(audio_codes) -> torch.Tensor:
        """Whatever the caller holds -> int64 [batch, 16, time]."""
        if isinstance(audio_codes, list) and audio_codes and isinstance(audio_codes[0], dict):
            audio_codes = audio_codes[0]["audio_codes"]
        if isinstance(audio_codes, torch.Tensor):
            t = audio_codes.long()
            return t if t.dim() == 3 else t.unsqueeze(0)
        frames = torch.as_tensor(np.asarray(audio_codes), dtype=torch.long)      # [time, 16]
        if frames.numel() == 0:
            return torch.zeros(1, 16, 0, dtype=torch.long)
        return frames.t().unsqueeze(0)

    @torch.inference_mode()
    def decode(self, audio_codes):
        batch = self._as_batch(audio_codes)
        if batch.shape[-1] == 0:
            return [], self.sample_rate
        wavs, sr = self.tokenizer.decode([{"audio_codes": b.t().contiguous()} for b in batch])      # qwen_tts wants [time, 16]
        return [np.asarray(w) for w in wavs], int(sr)

    @torch.inference_mode()
    def encode(self, audio, sr: int | None = None) -> torch.Tensor:
        a = audio.detach().cpu().numpy() if isinstance(audio, torch.Tensor) else np.asarray(audio)
        if a.ndim == 1:
            a = a[None, :]
        res = self.tokenizer.encode(a, sr=sr or self.sample_rate)
        codes = res.audio_codes if hasattr(res, "audio_codes") else res["audio_codes"]
        return torch.stack([torch.as_tensor(c).long().t() for c in codes])
