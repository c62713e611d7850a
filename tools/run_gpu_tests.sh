#!/bin/bash
# the whole GPU suite as the round end runs it (one process, -x), output kept under gpurun_out/
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gputests; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 1000 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu -s tests > $O/tests.log 2>&1
rc=$?
grep -a "id parity\|^\[logits\]\|^\[rows\]" $O/tests.log | cut -c1-230 > $O/parity_lines.txt
grep -a "FAILED\|Error\|passed\|failed" $O/tests.log | tail -15
exit $rc
