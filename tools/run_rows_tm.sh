#!/bin/bash
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rowstm; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 600 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_hip_ops.py tests/test_hip_fp8.py > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for tm in 0 4 0 4; do
  echo "== fp8 TM=$tm" >> $O/prefill.txt
  QTTS_WEIGHT_DTYPE=fp8 QTTS_ROWS_TM=$tm QTTS_FRAME_PROMPT=48,96,128,200,400,900 timeout -k 10 200 python3 -u tools/microbench.py prefill >> $O/prefill.txt 2>$O/prefill_$tm.err
done
cat $O/prefill.txt
