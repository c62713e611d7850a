"""MI355X-native Qwen3-TTS talker + code-predictor decode engine.

Same module paths and class names as the reference package
(/root/reference/nano-qwen3tts-vllm/__init__.py); the arithmetic of the
`run_step` hot path is hand-written HIP for gfx950 behind a C-ABI library
(csrc/, include/qtts_hip.h)."""
__version__ = "0.1.0"
