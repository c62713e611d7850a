#!/bin/bash
# frame-engine tests, then the arrivals workload under the three admission policies on one box
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/adm; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 600 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_hip_frame_engine.py tests/test_hip_soak.py tests/test_hip_interface.py tests/test_replicas.py > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do
for m in serial beside join; do
  QTTS_ADMISSION=$m timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --steps 50 > $O/bench_${m}_$i.json 2>$O/bench_${m}_$i.err
  python3 - $m $O/bench_${m}_$i.json <<'P'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
a = d["arrivals"]
print(sys.argv[1], "first p50/p99", round(a["first_chunk_ms_p50"], 2), round(a["first_chunk_ms_p99"], 2), "inter p50/p99", round(a["inter_chunk_ms_p50"], 2), round(a["inter_chunk_ms_p99"], 2),
      "tok/s", round(a["codec_tokens_per_s"]), "| all at once first", round(a["all_at_once"]["first_chunk_ms_p50"], 2), "| steady ms", round(d["ms_per_step"], 3), flush=True)
P
done
done
