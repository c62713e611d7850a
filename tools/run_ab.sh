#!/bin/bash
# op tests, then the prompt-GEMM table, the in-kernel timeline and the bench line of the current build
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
timeout -k 10 600 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_hip_ops.py tests/test_hip_fp8.py tests/test_hip_engine.py tests/test_hip_frame_engine.py > $O/tests.log 2>&1
cd tools
QTTS_ROWS=32,64,200,400 timeout -k 10 200 python3 -u microbench.py rows > $O/rows.txt 2>/dev/null
QTTS_ROWS=200 timeout -k 10 100 python3 -u microbench.py rowtrace > $O/rowtrace.txt 2>/dev/null
QTTS_FRAME_PROMPT=16,48,128,200,400 timeout -k 10 200 python3 -u microbench.py prefill > $O/prefill.txt 2>/dev/null
cd $R
timeout -k 10 200 python3 -u bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null
