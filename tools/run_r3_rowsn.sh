#!/bin/bash
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3rowsn; rm -rf $O; mkdir -p $O; cd $R
for lib in $R/tools/libqtts_hip_tuning*.so; do
  echo "== $(basename $lib)" | tee -a $O/rowsn.txt
  QTTS_TUNING_LIB=$lib timeout -k 10 300 python3 -u tools/microbench.py rowsn >> $O/rowsn.txt 2>> $O/err.log
  for cfg in "gate_up 32 2048" "gate_up 32 1024" "qkv 32 1024"; do
    set -- $cfg
    for n in 0 1; do
      QTTS_TUNING_LIB=$lib QTTS_LINEAR_DBG=64 QTTS_ROWS_NORM=$n QTTS_ROWS_ONLY=$1 QTTS_ROWS=$2 QTTS_ROWS_K=$3 timeout -k 10 120 python3 -u tools/microbench.py rowtrace >> $O/rowsn.txt 2>> $O/err.log
    done
  done
done
cat $O/rowsn.txt
