#!/bin/bash
# decode attention: in-kernel timeline, launch times per (rows, context), then the op tests and a bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3attn; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 300 python3 -u tools/microbench.py attntrace > $O/attntrace.txt 2>> $O/err.log || { tail -5 $O/err.log; exit 1; }
QTTS_ATTN_CTX=150,290,500 timeout -k 10 300 python3 -u tools/microbench.py attn > $O/attn.txt 2>> $O/err.log || { tail -5 $O/err.log; exit 1; }
cat $O/attntrace.txt $O/attn.txt
timeout -k 10 600 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_hip_ops.py tests/test_zz_no_alloc.py tests/test_hip_frame_engine.py > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for S in 8 32 64; do
  timeout -k 10 300 python3 -u bench.py --streams $S --steps 50 --warmup 5 --no-cpu-baseline --no-arrivals > $O/bench_$S.json 2>> $O/err.log || { tail -5 $O/err.log; exit 1; }
  python3 -c "
import json
x=json.loads([l for l in open('$O/bench_$S.json') if l.startswith('{')][-1]); print('streams', $S, 'ms/frame', round(x['ms_per_step'],3))"
done
