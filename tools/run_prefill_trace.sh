#!/bin/bash
# kernel trace of a prompt's prefill (QTTS_FRAME_PROMPT tokens) reduced to a per-kernel table
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ptrace; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for L in ${QTTS_TRACE_PROMPTS:-200}; do
  QTTS_FRAME_PROMPT=$L timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$L -- python3 $R/tools/microbench.py prefill > /dev/null 2>&1
  python3 $R/tools/prefill_breakdown.py $O/t$L > $O/breakdown_$L.txt
  rm -rf $O/t$L
done
