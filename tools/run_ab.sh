# Runs on the GPU box: one GPU test file, then bench.py twice with an environment variable at 1 / 0 (A/B of a switch).
#   bash tools/run_ab.sh <ENV_NAME> [pytest -k expression]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_$1; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python3 -u -m pytest tests/test_hip_frame_engine.py tests/test_hip_engine.py -m gpu -x -q --timeout 60 --timeout-method thread ${2:+-k "$2"} > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for v in 1 0 1 0; do
  env $1=$v timeout -k 10 300 python3 -u bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-arrivals > $O/bench_$v.json 2> $O/bench_$v.err || { tail -20 $O/bench_$v.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$1=$v', 'ms/frame', round(d['ms_per_step'],4), 'tokens/s', round(d['value']), 'dominant us', round(d['roofline']['us_per_launch'],2))"
done
