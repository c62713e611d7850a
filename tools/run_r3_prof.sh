#!/bin/bash
# per-kernel profile of a frame at S streams (product library), S from $1 (default 32)
set -e -o pipefail
S=${1:-32}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof; mkdir -p $O; cd $R
timeout -k 10 300 python3 -u bench.py --streams $S --steps 50 --warmup 5 --no-cpu-baseline --no-arrivals > $O/bench_$S.json 2>> $O/err.log
python3 -c "
import json
x=json.loads([l for l in open('$O/bench_$S.json') if l.startswith('{')][-1])
print('streams', $S, 'ms/frame', round(x['ms_per_step'],3))"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$S -- python3 $R/bench.py --streams $S --steps 30 --warmup 5 --no-cpu-baseline --no-arrivals > $O/bench_${S}_profiled.json 2> $O/trace.err
python3 $R/tools/profile_summary.py $O/trace_$S/* $O/summary_$S.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --streams $S --steps 30 --warmup 5" > /dev/null
rm -rf $O/trace_$S
sed -n '/last frame/,$p' $O/summary_$S.txt
