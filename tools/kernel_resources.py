"""Prints VGPR / scratch / occupancy per kernel of the library's translation units (compile-only); an argument
selects the units whose name contains it (gemm, attention, glue, model)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nano-qwen3tts-vllm_amd", "csrc")
units = [u for u in ("qtts_gemm.hip", "qtts_attention.hip", "qtts_glue.hip", "qtts_model.hip") if len(sys.argv) < 2 or sys.argv[1] in u]
out = "".join(subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-I", os.path.join(ROOT, "include"),
                              "-I", CSRC, "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, u), "-o", "/tmp/qtts_res.o"],
                             capture_output=True, text=True).stderr for u in units)
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
