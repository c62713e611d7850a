"""Makes the measurement scripts use the -DQTTS_TUNING build of the library (tools/libqtts_hip_tuning.so: dispatch
thresholds as settable process-wide variables, in-kernel timestamps, the grid-barrier probe, a stand-alone prompt-GEMM entry).  Import this module before nano_qwen3tts_vllm.hip; the shipped library has none of these entry points.

    python nano-qwen3tts-vllm_amd/csrc/build.py --tuning      # once, in the build container
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# QTTS_TUNING_LIB: another -DQTTS_TUNING build of the same source (an experiment compiled with extra -D flags)
LIB = os.environ.get("QTTS_TUNING_LIB") or os.path.join(ROOT, "tools", "libqtts_hip_tuning.so")
if not os.path.exists(LIB):
    raise SystemExit(f"{LIB} is missing: build it with `python nano-qwen3tts-vllm_amd/csrc/build.py --tuning`")
os.environ["QTTS_LIB"] = LIB
sys.path.insert(0, ROOT)
from nano_qwen3tts_vllm import hip  # noqa: E402

_p, _i = C.c_void_p, C.c_int
for _name, (_res, _args) in {
    "qtts_tune_linear": (None, [_i, _i]),
    "qtts_tune_subtiles": (None, [_i]),
    "qtts_tune_prompt_attention": (None, [_i]),
    "qtts_tune_rows": (None, [_i]),
    "qtts_tune_rows_tm": (None, [_i]),
    "qtts_tune_rows_norm": (None, [_i, _i]),
    "qtts_tune_rows_tm1": (None, [_i]),
    "qtts_tune_attn_wide": (None, [_i]),
    "qtts_tune_shared_prologue": (None, [_i]),
    "qtts_debug_linear": (None, [_i]),
    "qtts_debug_linear_trace": (None, [_p]),
    "qtts_debug_linear_rows": (_i, [_p, _i, _i, _p, _p, C.c_float, _p, _p, _i, _i, _p, _i, _i, _p, _i, _i, _i, _i, _p]),
    "qtts_debug_grid_barrier": (_i, [_p, _i, _i, _i, _i, _i, _i, _p]),
}.items():
    _fn = getattr(hip.lib, _name)
    _fn.restype, _fn.argtypes = _res, _args


def apply_env_flags():
    """The dispatch switches of the tuning build, from the environment (A/B runs of one script)."""
    e = os.environ
    if e.get("QTTS_ROWS_NORM"):                     # "on[,max rows]": prompt GEMMs normalise their rows themselves
        v = [int(t) for t in e["QTTS_ROWS_NORM"].split(",")]
        hip.lib.qtts_tune_rows_norm(v[0], v[1] if len(v) > 1 else 1 << 30)
    if e.get("QTTS_ROWS_TM1"):
        hip.lib.qtts_tune_rows_tm1(int(e["QTTS_ROWS_TM1"]))
    if e.get("QTTS_ATTN_WIDE"):
        hip.lib.qtts_tune_attn_wide(int(e["QTTS_ATTN_WIDE"]))
    if e.get("QTTS_ROWS_TM"):
        hip.lib.qtts_tune_rows_tm(int(e["QTTS_ROWS_TM"]))
    if e.get("QTTS_ROWS_MIN"):
        hip.lib.qtts_tune_rows(int(e["QTTS_ROWS_MIN"]))
    if e.get("QTTS_MFMA_ATTN_ROWS"):
        hip.lib.qtts_tune_prompt_attention(int(e["QTTS_MFMA_ATTN_ROWS"]))
