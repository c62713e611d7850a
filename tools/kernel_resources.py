"""Prints VGPR / scratch / occupancy per kernel of csrc/qtts_kernels.hip (compile-only)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "nano-qwen3tts-vllm_amd", "csrc", "qtts_kernels.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-I", os.path.join(ROOT, "include"),
                      "-Rpass-analysis=kernel-resource-usage", src, "-o", "/tmp/qtts_res.o"], capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    for key in ["VGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:", "SGPRs:"]:
        if key in line and cur:
            rows[cur][key] = line.split(key)[1].split()[0]
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for n, (k, v) in zip(names, rows.items()):
    n = n.split("(")[0].replace("void ", "")
    flag = "  <-- SPILL" if v.get("ScratchSize [bytes/lane]:") != "0" else ""
    print(f"{n:50s} vgpr {v.get('VGPRs:'):>4} scratch {v.get('ScratchSize [bytes/lane]:'):>4} occ {v.get('Occupancy [waves/SIMD]:')} sgpr {v.get('SGPRs:')}{flag}")
