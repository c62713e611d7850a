#!/bin/bash
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prefill; rm -rf $O; mkdir -p $O; cd $R
for v in "0" "1" "1,64" "1,128"; do
  echo "== QTTS_ROWS_NORM=$v" >> $O/prefill.txt
  QTTS_ROWS_NORM=$v QTTS_FRAME_PROMPT=24,48,96,128,200,400 timeout -k 10 300 python3 -u tools/microbench.py prefill >> $O/prefill.txt 2>> $O/err.log
done
cat $O/prefill.txt
