#!/bin/bash
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetch; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
QTTS_FRAME_PROMPT=300 QTTS_FRAMES=4 timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/dec -- python3 $R/tools/microbench.py decode_eager > $O/log.txt 2>&1
python3 $R/tools/fetch_by_kernel.py $O/dec | head -40
rm -rf $O/dec
