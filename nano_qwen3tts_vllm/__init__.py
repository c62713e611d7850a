"""Import shim: the package sources live in `nano-qwen3tts-vllm_amd/` (a directory
name Python cannot import); this makes them importable under the reference's
package name `nano_qwen3tts_vllm` (reference: /root/reference/pyproject.toml:23-24)."""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nano-qwen3tts-vllm_amd")
__path__ = [_src]
with open(_os.path.join(_src, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_src, "__init__.py"), "exec"))
