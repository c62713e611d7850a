"""Builds libqtts_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python nano-qwen3tts-vllm_amd/csrc/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = [os.path.join(HERE, "qtts_kernels.hip")]
OUT = os.path.join(HERE, "libqtts_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = SRC + [os.path.join(HERE, "qtts_resident.inc"), os.path.join(ROOT, "include", "qtts_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    # kernel-argument preload: the first 16 dwords of a kernel's scalar arguments arrive in SGPRs with the wave
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-mllvm", "-amdgpu-kernarg-preload-count=16",
           "-I", os.path.join(ROOT, "include"), "-o", OUT] + SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
