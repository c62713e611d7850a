#!/bin/bash
# HBM fetch per launch of every kernel of a frame (8 and 32 streams, 300-token prompts), frames launched eagerly
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetch; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for S in 8 32; do
  QTTS_STREAMS=$S QTTS_FRAME_PROMPT=300 QTTS_FRAMES=4 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/dec$S -- python3 $R/tools/microbench.py decode_eager > $O/log$S.txt 2>&1
  echo "== $S streams, prompts of 300 tokens: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/microbench.py decode_eager (prefill kernels included; fetch = 2 x FETCH_SIZE KiB, gfx950)" >> $O/fetch_by_kernel.txt
  python3 $R/tools/fetch_by_kernel.py $O/dec$S 45 >> $O/fetch_by_kernel.txt
  rm -rf $O/dec$S
done
cat $O/fetch_by_kernel.txt
