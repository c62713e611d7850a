#!/bin/bash
# the bench line as the driver runs it (default flags), then the GPU tests that changed
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3bench; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 600 python3 -u bench.py > $O/bench_line.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<'P'
import json, os
x = json.loads([l for l in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3bench/bench_line.json")) if l.startswith("{")][-1])
print("ms/frame", round(x["ms_per_step"], 3), "value", round(x["value"]), "frame frac", round(x["frame_roofline"]["frac"], 3))
print("roofline kernel:", x["roofline"]["kernel"], "frac", round(x["roofline"]["frac"], 3), "us", round(x["roofline"]["us_per_launch"], 2))
for r in x["roofline"]["by_launch"]:
    print(f"  {r['launch']:48s} x{r['per_frame']:3d} {r['us']:6.2f} us  {r['us_per_frame']:7.1f} us/frame  {r['achieved_GBps']/1e3:5.2f} TB/s  {100*r['frac']:4.1f} %")
print("host:", json.dumps(x["host"])[:600])
a = x.get("arrivals") or {}
print("arrivals first chunk p50/p99", a.get("first_chunk_ms_p50"), a.get("first_chunk_ms_p99"), "inter-chunk p50/p99", a.get("inter_chunk_ms_p50"), a.get("inter_chunk_ms_p99"))
print("first_chunk_ms_p50 (idle engine)", x["first_chunk_ms_p50"], "cpu", x.get("cpu_baseline", {}).get("value"))
P
timeout -k 10 900 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_hip_frame_engine.py tests/test_hip_interface.py tests/test_replicas.py tests/test_hip_soak.py > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
