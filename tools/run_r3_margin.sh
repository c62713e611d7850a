#!/bin/bash
# arrivals workload with the adaptive launch margin against the fixed 350 us
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3margin; rm -rf $O; mkdir -p $O; cd $R
for rep in 1 2; do
  for m in adaptive 350 250; do
    if [ $m = adaptive ]; then unset QTTS_LAUNCH_MARGIN_US; else export QTTS_LAUNCH_MARGIN_US=$m; fi
    timeout -k 10 300 python3 -u bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_${m}_$rep.json 2>> $O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
python3 - <<'P'
import json, os, glob
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3margin")
for f in sorted(glob.glob(O + "/bench_*.json")):
    x = json.loads([l for l in open(f) if l.startswith("{")][-1]); a = x["arrivals"]
    print(os.path.basename(f), "ms/frame", round(x["ms_per_step"], 3), "first chunk p50/p99", round(a["first_chunk_ms_p50"], 2), round(a["first_chunk_ms_p99"], 2),
          "pooled", round(a["first_chunk_ms_pooled"]["p50"], 2), round(a["first_chunk_ms_pooled"]["p99"], 2),
          "inter p50/p99", round(a["inter_chunk_ms_p50"], 3), round(a["inter_chunk_ms_p99"], 2), "margin", round(x["host"]["launch_margin_ms"], 3),
          "polling ms/frame", round(x["host"]["polling"]["ms_per_frame"], 3))
P
