#!/bin/bash
# Runs on the GPU box: the measurements kept under profiles/ (bench line, kernel trace summary, PMC passes,
# API-level examples, prefill table).  Output goes to gpurun_out/refresh/; every step has its own time limit.
set -e -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
step() { echo "== $1" >&2; }

step bench
timeout -k 10 400 python3 -u $R/bench.py --steps 100 --warmup 10 > $O/bench_line.json 2> $O/bench.err
step bench-trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_line_profiled.json 2> $O/trace.err
step pmc-fetch
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/microbench.py dominant > /dev/null 2>&1
step pmc-write
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/microbench.py dominant > /dev/null 2>&1
step pmc-mfma
QTTS_ROWS=200 timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/tools/microbench.py rows > /dev/null 2>&1
step prefill
QTTS_FRAME_PROMPT=16,48,128,200,400,900 timeout -k 10 300 python3 -u $R/tools/microbench.py prefill > $O/prefill.txt 2>/dev/null
QTTS_FRAME_PROMPT=200 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prefill_trace -- python3 $R/tools/microbench.py prefill > /dev/null 2>&1
step examples
cd $R
timeout -k 10 300 python3 -u examples/quick_benchmark.py > $O/api_quick_benchmark.json 2> $O/qb.err
timeout -k 10 300 python3 -u examples/batch_async_example.py --streams 8 > $O/api_batch_async_fused.json 2> $O/ba.err
timeout -k 10 300 python3 -u examples/batch_async_example.py --streams 8 --engine workers > $O/api_batch_async_workers.json 2> $O/bw.err
step sweep
for S in 1 16 32 64 128; do
  timeout -k 10 300 python3 -u bench.py --streams $S --steps 50 --warmup 5 --no-cpu-baseline > $O/sweep_$S.json 2>/dev/null
done
step rows
QTTS_ROWS=32,64,128,200,400 timeout -k 10 300 python3 -u tools/microbench.py rows > $O/rows.txt 2>/dev/null
echo done >&2
