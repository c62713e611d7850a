#!/bin/bash
# frames at 8 / 32 / 64 streams, the per-kernel profile of a 32-stream frame, the prompt GEMMs at 32..200 rows
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3base; rm -rf $O; mkdir -p $O; cd $R
for S in 8 32 64; do
  timeout -k 10 300 python3 -u bench.py --streams $S --steps 50 --warmup 5 --no-cpu-baseline --no-arrivals > $O/bench_$S.json 2>> $O/err.log
  tail -c 600 $O/bench_$S.json | head -c 300; echo
done
QTTS_LINEAR_DBG=64 QTTS_ROWS=32,64,128,200 timeout -k 10 300 python3 -u tools/microbench.py rows > $O/rows_frag.txt 2>> $O/err.log
cat $O/rows_frag.txt
for M in 32 64; do
  QTTS_LINEAR_DBG=64 QTTS_ROWS=$M QTTS_ROWS_ONLY=gate_up timeout -k 10 120 python3 -u tools/microbench.py rowtrace >> $O/rowtrace.txt 2>> $O/err.log
  QTTS_LINEAR_DBG=64 QTTS_ROWS=$M QTTS_ROWS_ONLY=o_proj timeout -k 10 120 python3 -u tools/microbench.py rowtrace >> $O/rowtrace.txt 2>> $O/err.log
done
cat $O/rowtrace.txt
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --streams 32 --steps 30 --warmup 5 --no-cpu-baseline --no-arrivals > $O/bench_32_profiled.json 2> $O/trace.err
python3 $R/tools/profile_summary.py $O/trace/* $O/summary_32.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --streams 32 --steps 30 --warmup 5" > /dev/null
rm -rf $O/trace
