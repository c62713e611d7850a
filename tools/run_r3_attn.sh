#!/bin/bash
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3attn; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 300 python3 -u tools/microbench.py attn > $O/attn.txt 2>> $O/err.log || { tail $O/err.log; exit 1; }
cat $O/attn.txt
