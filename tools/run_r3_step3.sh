#!/bin/bash
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s3; rm -rf $O; mkdir -p $O; cd $R
for S in 24 32 64; do
  for t in 1 2; do
    QTTS_ROWS_TM1=$t timeout -k 10 300 python3 -u tools/bench_with_flags.py --streams $S --steps 50 --warmup 5 --no-cpu-baseline --no-arrivals > $O/bench_${S}_t$t.json 2>> $O/err.log
    python3 -c "
import json
x=json.loads([l for l in open('$O/bench_${S}_t$t.json') if l.startswith('{')][-1])
print('streams', $S, 'tm1 mode', $t, 'ms/frame', round(x['ms_per_step'],3))"
  done
done
timeout -k 10 600 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_hip_rows_norm.py tests/test_hip_ops.py > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
