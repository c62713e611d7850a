#!/bin/bash
# tests with the current build, then the bench line with the previous build (csrc/libqtts_hip_prev.so, built by hand from
# the last commit) and the current one in turn, on one box
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 600 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_hip_ops.py tests/test_hip_fp8.py tests/test_hip_engine.py tests/test_hip_frame_engine.py tests/test_zz_no_alloc.py > $O/tests.log 2>&1
for i in 1 2 3; do
  for v in prev cur; do
    lib=$R/nano-qwen3tts-vllm_amd/csrc/libqtts_hip.so; [ $v = prev ] && lib=$R/nano-qwen3tts-vllm_amd/csrc/libqtts_hip_prev.so
    extra="--no-arrivals"; [ $i = 1 ] && extra=""
    QTTS_LIB=$lib timeout -k 10 200 python3 -u bench.py --no-cpu-baseline $extra --steps 200 > $O/bench_${v}_$i.json 2>/dev/null
  done
done
for S in 64 128; do
  for v in prev cur; do
    lib=$R/nano-qwen3tts-vllm_amd/csrc/libqtts_hip.so; [ $v = prev ] && lib=$R/nano-qwen3tts-vllm_amd/csrc/libqtts_hip_prev.so
    QTTS_LIB=$lib timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --no-arrivals --streams $S --steps 50 > $O/sweep_${v}_$S.json 2>/dev/null
  done
done
