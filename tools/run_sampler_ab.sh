#!/bin/bash
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sampler_ab; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 300 python3 -u -m pytest -x -q --timeout 200 --timeout-method thread -m gpu tests/test_hip_ops.py -k "sample or argmax" > $O/tests.log 2>&1
for i in 1 2; do
  for v in prev cur; do
    lib=$R/nano-qwen3tts-vllm_amd/csrc/libqtts_hip.so; [ $v = prev ] && lib=$R/nano-qwen3tts-vllm_amd/csrc/libqtts_hip_prev.so
    QTTS_LIB=$lib timeout -k 10 200 python3 -u examples/batch_async_example.py --streams 8 --sample > $O/async_${v}_$i.json 2>/dev/null
  done
done
