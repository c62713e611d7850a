#!/bin/bash
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fp8_ab; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 600 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_hip_fp8.py "tests/test_hip_engine.py::test_folded_input_tables_give_the_same_bits" "tests/test_hip_engine.py::test_full_size_frames_match_oracle" tests/test_hip_frame_engine.py > $O/tests.log 2>&1
for w in bf16 fp8; do
  timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --no-arrivals --weights $w --steps 200 > $O/bench_$w.json 2>/dev/null
done
