#!/bin/bash
# the whole GPU suite as the round end runs it, then the default bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
bash tools/run_gpu_tests.sh || exit 1
O=$R/gpurun_out/r3full; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -u bench.py > $O/bench_line.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 -c "
import json
x = json.loads([l for l in open('$O/bench_line.json') if l.startswith('{')][-1])
print('ms/frame', round(x['ms_per_step'], 3), 'value', round(x['value']), 'frame frac', round(x['frame_roofline']['frac'], 3), 'roofline', x['roofline']['kernel'], round(x['roofline']['frac'], 3))
"
