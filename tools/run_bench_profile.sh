#!/bin/bash
# kernel trace of the bench command, reduced to the text summary (the raw trace stays on the box)
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/benchprof; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-arrivals > $O/bench_line_profiled.json 2> $O/trace.err
python3 $R/tools/profile_summary.py $O/trace/* $O/summary.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-arrivals" > /dev/null
rm -rf $O/trace
