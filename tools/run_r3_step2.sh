#!/bin/bash
# table-fed burst steps beyond 16 rows: parity tests, then frames at 16..128 streams; decode attention workgroup size A/B at 32 / 64
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s2; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 900 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_hip_rows_norm.py tests/test_hip_engine.py tests/test_hip_frame_engine.py > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for S in 8 16 24 32 64 128; do
  timeout -k 10 300 python3 -u bench.py --streams $S --steps 50 --warmup 5 --no-cpu-baseline --no-arrivals > $O/bench_prod_$S.json 2>> $O/err.log
  python3 -c "
import json
x=json.loads([l for l in open('$O/bench_prod_$S.json') if l.startswith('{')][-1])
print('product library: streams', $S, 'ms/frame', round(x['ms_per_step'],3), 'tok/s', round(x['value']))"
done
for S in 32 64; do
  for w in 0 1 2; do
    QTTS_ATTN_WIDE=$w timeout -k 10 300 python3 -u tools/bench_with_flags.py --streams $S --steps 50 --warmup 5 --no-cpu-baseline --no-arrivals > $O/bench_${S}_w$w.json 2>> $O/err.log
    python3 -c "
import json
x=json.loads([l for l in open('$O/bench_${S}_w$w.json') if l.startswith('{')][-1])
print('streams', $S, 'attn_wide', $w, 'ms/frame', round(x['ms_per_step'],3))"
  done
done
