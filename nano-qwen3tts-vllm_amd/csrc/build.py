"""Builds libqtts_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python nano-qwen3tts-vllm_amd/csrc/build.py [--force] [--tuning]

The library is four translation units over one shared header (qtts_common.h): qtts_gemm.hip (weight packing, the decode
and prompt GEMM kernels, RMSNorm), qtts_attention.hip, qtts_glue.hip (picks, gathers, per-frame glue) and qtts_model.hip
(the decoder stack, graphs).  They are compiled side by side and linked into one shared object.

--tuning builds tools/libqtts_hip_tuning.so instead (-DQTTS_TUNING: dispatch thresholds as settable globals, in-kernel
timestamps, the grid-barrier probe, a stand-alone prompt-GEMM entry) for the measurement scripts under tools/, which
load it through QTTS_LIB.  The shipped library has none of that.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
UNITS = ["qtts_gemm.hip", "qtts_attention.hip", "qtts_glue.hip", "qtts_model.hip"]
SRC = [os.path.join(HERE, u) for u in UNITS]
OUT = os.path.join(HERE, "libqtts_hip.so")
OUT_TUNING = os.path.join(ROOT, "tools", "libqtts_hip_tuning.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def needs_build(out=OUT):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = SRC + [os.path.join(ROOT, "include", "qtts_hip.h"), os.path.join(HERE, "qtts_common.h")]
    if out == OUT_TUNING:
        deps += [os.path.join(HERE, "qtts_tuning.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, tuning=False):
    out = OUT_TUNING if tuning else OUT
    if not force and not needs_build(out):
        return out
    objdir = os.path.join(HERE, "build", "tuning" if tuning else "product")
    os.makedirs(objdir, exist_ok=True)
    # kernel-argument preload: the first 16 dwords of a kernel's scalar arguments arrive in SGPRs with the wave
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=16",
             "-I", os.path.join(ROOT, "include"), "-I", HERE] + (["-DQTTS_TUNING"] if tuning else [])

    def compile_unit(src):
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(SRC)) as pool:
        objs = list(pool.map(compile_unit, SRC))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, tuning="--tuning" in sys.argv)
