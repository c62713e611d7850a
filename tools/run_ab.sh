#!/bin/bash
# the bench line with the previous build (tools/libqtts_hip_prev.so, built by hand from an earlier commit) and the
# current one in turn, on one box: 8 streams three times each, then 16 / 32 / 64 streams
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; rm -rf $O; mkdir -p $O; cd $R
run() {   # name lib streams steps
  QTTS_LIB=$2 timeout -k 10 300 python3 -u bench.py --no-cpu-baseline --no-arrivals --streams $3 --steps $4 > $O/$1.json 2>> $O/err.log || { tail -5 $O/err.log; exit 1; }
  python3 -c "
import json
x=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); print('$1', 'streams', $3, 'ms/frame', round(x['ms_per_step'],4))"
}
for i in 1 2 3; do
  run prev_8_$i $R/tools/libqtts_hip_prev.so 8 200
  run cur_8_$i $R/nano-qwen3tts-vllm_amd/csrc/libqtts_hip.so 8 200
done
for S in 16 32 64; do
  run prev_$S $R/tools/libqtts_hip_prev.so $S 50
  run cur_$S $R/nano-qwen3tts-vllm_amd/csrc/libqtts_hip.so $S 50
done
