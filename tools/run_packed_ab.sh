#!/bin/bash
set -e -o pipefail
cd $GRAFT_REPO_ROOT/tools; O=$GRAFT_REPO_ROOT/gpurun_out/packed_ab.txt; : > $O
for dbg in 0 64; do
  echo "== QTTS_LINEAR_DBG=$dbg (64: activations read in fragment order - timing only, wrong values)" >> $O
  QTTS_LINEAR_DBG=$dbg QTTS_ROWS=64,200,400 timeout -k 10 200 python3 -u microbench.py rows >> $O 2>/dev/null
done
