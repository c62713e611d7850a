#!/bin/bash
# the whole GPU suite, then a kernel trace of a 200-token prefill
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
bash tools/run_gpu_tests.sh || exit 1
QTTS_FRAME_PROMPT=200 bash tools/run_prefill_trace.sh || exit 1
head -12 $R/gpurun_out/ptrace/breakdown_200.txt | cut -c1-170
