#!/bin/bash
# the whole GPU suite, then previous build against current build on one box
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
bash tools/run_gpu_tests.sh; rc=$?
grep -a "^\[logits\]" $R/gpurun_out/gputests/tests.log | sed 's/.*prompts \(([^)]*)\) (\([a-z]*\)) \([a-z]*\): HIP within \([0-9.]*\) .*sums within \([0-9.]*\) .*/\1 \2 \3 \4 \5/' | awk '{n=NF; d=$(n-1); no=$n; b=2*no+2; if (b-d < 12) printf "thin margin: %-60s d.max %5.1f bound %5.1f\n", substr($0,1,60), d, b}'
[ $rc = 0 ] || exit $rc
bash tools/run_ab.sh
