# Runs on the GPU box: the frame-engine GPU tests, then the arrivals workload with the prefill beside / between frames.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2arr; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python3 -u -m pytest tests/test_hip_frame_engine.py tests/test_hip_interface.py -m gpu -x -q --timeout 60 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for ov in 1 0; do
  QTTS_OVERLAP_PREFILL=$ov timeout -k 10 300 python3 -u bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_overlap$ov.json 2> $O/bench$ov.err || { tail -20 $O/bench$ov.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('$O/bench_overlap$ov.json').read().strip().splitlines()[-1])
a=d['arrivals']; print('overlap=$ov', 'ms/frame', round(d['ms_per_step'],3), {k:(round(v,2) if isinstance(v,float) else v) for k,v in a.items() if k.endswith('_p50') or k.endswith('_p99') or k in ('codec_tokens_per_s','elapsed_s','rtf_per_stream_mean')})"
done
