#!/bin/bash
# attention op tests + frame engine tests, then the 8-stream bench line twice
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3check; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 600 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_hip_ops.py tests/test_hip_engine.py tests/test_hip_frame_engine.py tests/test_zz_no_alloc.py > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for i in 1 2; do
  timeout -k 10 300 python3 -u bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-arrivals > $O/bench_$i.json 2>> $O/err.log || { tail -5 $O/err.log; exit 1; }
  python3 -c "
import json
x=json.loads([l for l in open('$O/bench_$i.json') if l.startswith('{')][-1]); print('8 streams ms/frame', round(x['ms_per_step'],3))"
done
