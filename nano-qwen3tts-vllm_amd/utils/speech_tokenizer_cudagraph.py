"""Codec decode (16 codebook ids per frame -> PCM) behind hipGraphs, one per window length.

Counterpart of P/utils/speech_tokenizer_cudagraph.py:18-169.  The codec network itself is not part of
that repository: it lives in the `qwen_tts` package (`Qwen3TTSTokenizer`, imported there at :11), which is
not installed here, so this module holds no codec arithmetic.  It holds what the reference's file holds —
graph capture per length for `model.decoder`, replay when the length matches, `decode`,
`chunked_decode`, `decode_codec_ids` — plus `decode_window_batched`, the hook the reference's server looks
for before it falls back to decode-everything-and-trim (examples/server.py:201-219,447-450).

`torch.cuda.CUDAGraph` is a hipGraph on ROCm.  A decoder can be handed in directly (`tokenizer=`) so the
wrapper is usable, and tested, without the `qwen_tts` package.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

try:  # the reference resolves the same class, optionally from a checkout named by QWEN_TTS_PATH (:8-11)
    from qwen_tts.inference.qwen3_tts_tokenizer import Qwen3TTSTokenizer as _Qwen3TTSTokenizer
    HAS_SPEECH_TOKENIZER = True
except ImportError:
    _Qwen3TTSTokenizer = None
    HAS_SPEECH_TOKENIZER = False

NUM_CODEBOOKS = 16


class DecoderGraphs:
    """hipGraphs of `decoder(codes[1, 16, T])` for a set of lengths T, sharing one memory pool.

    Lengths are captured longest first so the pool is sized once (as P/utils/speech_tokenizer_cudagraph.py:32).
    `install()` routes `decoder.forward` through `__call__`: the tokenizer's own `decode` then replays a graph
    whenever a request's length was captured and runs the module eagerly otherwise (:48-56).
    """

    def __init__(self, decoder, device, lengths: Sequence[int], batch: int = 1, code_high: int = 100):
        self.decoder = decoder
        self.eager = decoder.forward
        self.graphs: dict[int, torch.cuda.CUDAGraph] = {}
        self.inputs: dict[int, torch.Tensor] = {}
        self.outputs: dict[int, torch.Tensor] = {}
        self.batch = batch
        decoder.eval()
        pool = None
        stream = torch.cuda.Stream(device=device)
        with torch.inference_mode():
            for T in sorted(set(int(t) for t in lengths), reverse=True):
                codes = torch.randint(0, code_high, (batch, NUM_CODEBOOKS, T), device=device, dtype=torch.long)
                self.eager(codes)                      # lazy initialisation happens outside the capture
                torch.cuda.synchronize(device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, stream=stream):
                    out = self.eager(codes)
                pool = pool or g.pool()
                self.graphs[T], self.inputs[T], self.outputs[T] = g, codes, out
        torch.cuda.synchronize(device)

    @torch.inference_mode()          # the captured buffers are inference tensors
    def __call__(self, codes: torch.Tensor) -> torch.Tensor:
        T = codes.shape[-1]
        g = self.graphs.get(T)
        if g is None or codes.shape[0] != self.batch:
            return self.eager(codes)
        self.inputs[T].copy_(codes)
        g.replay()
        return self.outputs[T]       # overwritten by the next replay of the same length

    def install(self):
        self.decoder.forward = self
        return self

    def remove(self):
        self.decoder.forward = self.eager


def _as_codes(audio_codes, device) -> torch.Tensor:
    """[time, 16] ids (tensor or nested list) -> [1, 16, time] int64 on `device`."""
    codes = audio_codes if torch.is_tensor(audio_codes) else torch.tensor(audio_codes, dtype=torch.long)
    codes = codes.to(device=device, dtype=torch.long)
    if codes.dim() == 2:
        codes = codes.transpose(0, 1).unsqueeze(0)
    return codes


def chunked_decode(decoder, codes: torch.Tensor, chunk_size: int = 300, left_context_size: int = 25) -> torch.Tensor:
    """Decode [1, 16, T] in chunks of `chunk_size` frames, each preceded by up to `left_context_size` frames
    of context whose samples are dropped.  Used when the decoder has no `chunked_decode` of its own (the
    reference calls the decoder's, :152-153)."""
    up = int(decoder.total_upsample)
    T = codes.shape[-1]
    pieces = []
    for start in range(0, T, chunk_size):
        ctx = min(left_context_size, start)
        wav = decoder(codes[..., start - ctx:min(start + chunk_size, T)])
        pieces.append(wav[..., ctx * up:].clone())        # a graph's output buffer is reused by the next chunk
    return torch.cat(pieces, dim=-1)


class SpeechTokenizerCUDAGraph:
    """Same constructor and methods as the reference's class (:59-169).

    model_path: directory (or hub id) of the 12 Hz speech tokenizer; num_graph_lengths: graphs for
    T = 1..num_graph_lengths (0 disables capture).  `tokenizer=` hands in a loaded tokenizer object instead
    (anything with `.model.decoder`, `.decode(inputs)` and, optionally, `.config.sample_rate`).
    """

    def __init__(self, model_path: str | None = None, device: str | None = None, dtype: torch.dtype = torch.bfloat16,
                 num_graph_lengths: int = 50, tokenizer=None):
        device = device or ("cuda:0" if torch.cuda.is_available() else "cpu")
        if tokenizer is None:
            if not HAS_SPEECH_TOKENIZER:
                raise ImportError("qwen_tts not found: the codec network is that package's Qwen3TTSTokenizer; "
                                  "install it or pass tokenizer=")
            tokenizer = _Qwen3TTSTokenizer.from_pretrained(model_path, device_map=device)
        self.tokenizer = tokenizer
        self.tokenizer.model = self.tokenizer.model.to(dtype)
        cfg = getattr(tokenizer, "config", None)
        fe = getattr(tokenizer, "feature_extractor", None)
        if hasattr(cfg, "sample_rate"):
            self.sample_rate = cfg.sample_rate
        elif hasattr(fe, "sampling_rate"):
            self.sample_rate = fe.sampling_rate
        else:
            self.sample_rate = 12500
        self.device, self.dtype = device, dtype
        self.graphs = None
        if str(device).startswith("cuda") and num_graph_lengths > 0:
            self.graphs = DecoderGraphs(self.tokenizer.model.decoder, device,
                                        range(1, num_graph_lengths + 1)).install()

    # ------------------------------------------------------------------ reference surface
    @torch.inference_mode()
    def decode(self, inputs: List[dict]) -> Tuple[List, int]:
        """[{"audio_codes": [time, 16]}...] -> (list of float waveforms, sample rate); the tokenizer's own decode,
        whose decoder call goes through the graphs."""
        return self.tokenizer.decode(inputs)

    @torch.inference_mode()
    def chunked_decode(self, inputs: List[dict], chunk_size: int = 300, left_context_size: int = 25) -> Tuple[List, int]:
        codes = _as_codes(inputs[0]["audio_codes"], self.device)
        decoder = self.tokenizer.model.decoder
        own = getattr(decoder, "chunked_decode", None)
        wav = own(codes, chunk_size, left_context_size) if own is not None else \
            chunked_decode(decoder, codes, chunk_size, left_context_size)
        return [wav.squeeze().to(torch.float32).cpu().numpy()], self._output_rate()

    @torch.inference_mode()
    def decode_codec_ids(self, codec_ids: torch.Tensor) -> Tuple[List, int]:
        """[batch, 16, time] ids -> (waveforms, sample rate)."""
        return self.decode([{"audio_codes": c.transpose(0, 1) if c.dim() == 2 else c} for c in codec_ids])

    # ------------------------------------------------------------------ streaming window decode
    @torch.inference_mode()
    def decode_window_batched(self, batch_inputs: List[dict]) -> Tuple[List[np.ndarray], int]:
        """Each item: {"audio_codes": frames of (context + new), "left_context_frames": n}.  Decodes each window
        (a graph replay for windows of a captured length) and returns only the samples of the new frames — the
        result of the server's fallback (examples/server.py:201-219) without going through `decode`'s
        list/numpy round trip per request."""
        decoder = self.tokenizer.model.decoder
        up = int(decoder.total_upsample)
        wavs = []
        for r in batch_inputs:
            codes = _as_codes(r["audio_codes"], self.device)
            left = int(r.get("left_context_frames") or 0)
            if codes.shape[-1] == 0:
                wavs.append(np.zeros(0, dtype=np.float32))
                continue
            wav = decoder(codes).reshape(-1)[left * up:]
            wavs.append(wav.to(torch.float32).cpu().numpy())
        return wavs, self._output_rate()

    def _output_rate(self) -> int:
        get = getattr(self.tokenizer.model, "get_output_sample_rate", None)
        return int(get()) if get is not None else int(self.sample_rate)
