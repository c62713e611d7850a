#!/bin/bash
# parity tests first, then frames at 32 / 64 streams: norm launches (0) or rows normalised in the GEMMs (1) x one-row-tile forms off / on
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s1; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 600 python3 -u -m pytest -q --timeout 300 --timeout-method thread -m gpu -s tests/test_hip_rows_norm.py > $O/tests_rows.log 2>&1 || { grep -a "AssertionError\|^FAILED\|Error" $O/tests_rows.log | head -40; exit 1; }
tail -2 $O/tests_rows.log
timeout -k 10 900 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_hip_ops.py tests/test_hip_fp8.py > $O/tests_ops.log 2>&1 || { tail -30 $O/tests_ops.log; exit 1; }
tail -2 $O/tests_ops.log
for S in 32 64; do
  for v in "0 0" "1 0" "0 1" "1 1"; do
    set -- $v
    QTTS_ROWS_NORM=$1 QTTS_ROWS_TM1=$2 timeout -k 10 300 python3 -u tools/bench_with_flags.py --streams $S --steps 50 --warmup 5 --no-cpu-baseline --no-arrivals > $O/bench_${S}_n$1_t$2.json 2>> $O/err.log
    python3 -c "
import json
x=json.loads([l for l in open('$O/bench_${S}_n$1_t$2.json') if l.startswith('{')][-1])
print('streams', $S, 'rows_norm', $1, 'tm1', $2, 'ms/frame', round(x['ms_per_step'],3))"
  done
done
for S in 8 16 32 64; do
  timeout -k 10 300 python3 -u bench.py --streams $S --steps 50 --warmup 5 --no-cpu-baseline --no-arrivals > $O/bench_prod_$S.json 2>> $O/err.log
  python3 -c "
import json
x=json.loads([l for l in open('$O/bench_prod_$S.json') if l.startswith('{')][-1])
print('product library: streams', $S, 'ms/frame', round(x['ms_per_step'],3))"
done
QTTS_TUNING_LIB=$R/tools/libqtts_hip_tuning.so timeout -k 10 300 python3 -u tools/microbench.py rowsn > $O/rowsn.txt 2>> $O/err.log
cat $O/rowsn.txt
timeout -k 10 900 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu -s tests/test_hip_engine.py tests/test_hip_frame_engine.py tests/test_hip_interface.py > $O/tests_main.log 2>&1 || { grep -a "id parity\|FAILED\|Error" $O/tests_main.log | tail -30; exit 1; }
grep -a "id parity" $O/tests_main.log | cut -c1-250; tail -2 $O/tests_main.log
