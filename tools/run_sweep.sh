#!/bin/bash
# streams-per-GPU sweep of the current build (the S = 8 point is the bench line)
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sweep; rm -rf $O; mkdir -p $O; cd $R
for S in 1 8 16 24 32 64 128 256; do
  timeout -k 10 300 python3 -u bench.py --streams $S --steps 50 --warmup 5 --no-cpu-baseline --no-arrivals > $O/sweep_$S.json 2>> $O/err.log
done
python3 - <<'P'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "sweep")
sweep = []
for S in (1, 8, 16, 24, 32, 64, 128, 256):
    x = json.loads([l for l in open(os.path.join(O, f"sweep_{S}.json")) if l.startswith("{")][-1])
    sweep.append({"streams_per_gpu": S, "codec_tokens_per_s": round(x["value"]), "ms_per_frame": round(x["ms_per_step"], 3),
                  "rtf": round(x["rtf"], 4), "frame_roofline_frac": round(x["frame_roofline"]["frac"], 3), "steps": x["steps"]})
    print(sweep[-1])
json.dump({"note": "python bench.py --streams S --steps 50 --no-arrivals (1 GPU, MI355X, 1.7B shapes, greedy), one box; the contract line is S=8", "sweep": sweep},
          open(os.path.join(O, "r03_streams_sweep.json"), "w"), indent=1)
P
