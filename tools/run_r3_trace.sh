#!/bin/bash
# in-kernel timelines of the prompt GEMMs at 32 / 64 rows (tuning build)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3trace; rm -rf $O; mkdir -p $O; cd $R
for M in 32 64; do
  for n in 0 1; do
    QTTS_ROWS=$M QTTS_ROWS_NORM=$n QTTS_ROWS_ONLY=gate_up timeout -k 10 120 python3 -u tools/microbench.py rowtrace >> $O/trace.txt 2>> $O/err.log || exit 1
  done
  QTTS_ROWS=$M QTTS_ROWS_NORM=1 QTTS_ROWS_ONLY=qkv timeout -k 10 120 python3 -u tools/microbench.py rowtrace >> $O/trace.txt 2>> $O/err.log || exit 1
  QTTS_ROWS=$M QTTS_ROWS_ONLY=down timeout -k 10 120 python3 -u tools/microbench.py rowtrace >> $O/trace.txt 2>> $O/err.log || exit 1
done
QTTS_ROWS=8 QTTS_ROWS_ONLY=gate_up timeout -k 10 120 python3 -u tools/microbench.py rowtrace >> $O/trace.txt 2>> $O/err.log
cat $O/trace.txt
