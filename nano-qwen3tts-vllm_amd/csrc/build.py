"""Builds libqtts_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python nano-qwen3tts-vllm_amd/csrc/build.py [--force] [--tuning]

--tuning builds libqtts_hip_tuning.so instead (-DQTTS_TUNING: dispatch thresholds as settable globals, in-kernel
timestamps, the grid-barrier probe and the resident burst kernel of qtts_resident.inc) for the measurement scripts
under tools/, which load it through QTTS_LIB.  The shipped library has none of that.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = [os.path.join(HERE, "qtts_kernels.hip")]
OUT = os.path.join(HERE, "libqtts_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


OUT_TUNING = os.path.join(HERE, "libqtts_hip_tuning.so")


def needs_build(out=OUT):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = SRC + [os.path.join(ROOT, "include", "qtts_hip.h")]
    if out == OUT_TUNING:
        deps += [os.path.join(HERE, "qtts_resident.inc"), os.path.join(HERE, "qtts_tuning.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, tuning=False):
    out = OUT_TUNING if tuning else OUT
    if not force and not needs_build(out):
        return out
    # kernel-argument preload: the first 16 dwords of a kernel's scalar arguments arrive in SGPRs with the wave
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-mllvm", "-amdgpu-kernarg-preload-count=16",
           "-I", os.path.join(ROOT, "include"), "-I", HERE, "-o", out] + (["-DQTTS_TUNING"] if tuning else []) + SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, tuning="--tuning" in sys.argv)
