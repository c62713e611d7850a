"""Reduces the rocprofv3 passes of tools/run_attn_pmc.sh to profiles/r02_pmc_attention.json."""
import collections
import csv
import glob
import json
import os
import statistics
import sys

O = sys.argv[1]


def counter(dirname, name, match):
    f = glob.glob(os.path.join(O, dirname, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"] and r["Counter_Name"] == name:
            per[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))].append(float(r["Counter_Value"]))
    return per


def durations(dirname, match):
    f = glob.glob(os.path.join(O, dirname, "**", "*_kernel_trace.csv"), recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"]:
            grid = r.get("Grid_Size") or "x".join(r[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            per[(r["Kernel_Name"].split("(")[0].replace("void ", ""), grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    return per


def by_total_grid(d):
    """{kernel: {total grid threads: values}} (the counter files give one grid number, the traces three)."""
    out = collections.defaultdict(dict)
    for (k, grid), v in d.items():
        n = 1
        for p in str(grid).split("x"):
            n *= int(p) if p else 1
        out[k].setdefault(n, []).extend(v)
    return out


out = {"source": "tools/run_attn_pmc.sh: rocprofv3 --pmc <one group> --kernel-trace, separate passes; durations from a plain --kernel-trace of the same command",
       "prompt_attention": [], "decode_attention": []}
mf = by_total_grid(counter("pre_mfma", "SQ_VALU_MFMA_BUSY_CYCLES", "prefill_attn_mfma_kernel"))
ft = by_total_grid(counter("pre_fetch", "FETCH_SIZE", "prefill_attn_mfma_kernel"))
du = by_total_grid(durations("pre_trace", "prefill_attn_mfma_kernel"))
for k in mf:
    for grid, v in sorted(mf[k].items()):
        us = statistics.median(du[k][grid]) if grid in du.get(k, {}) else None
        rows = grid // 512 // 8 * 16                      # grid = (row tiles, 8 kv heads) x 512 threads
        busy = statistics.median(v)
        e = {"kernel": k, "prompt_rows_padded": rows, "launches": len(v), "mfma_busy_cycles_per_launch": busy, "us_per_launch": us}
        if us:
            e["mfma_utilisation"] = busy / (us * 1e-6 * 2.4e9 * 1024)
        if grid in ft.get(k, {}):
            kib = statistics.median(ft[k][grid])
            e["hbm_fetch_bytes_per_launch"] = 2 * kib * 1024       # gfx950 tallies 128-B read requests at 64 B
            e["algorithmic_kv_bytes"] = rows * 8 * 128 * 2 * 2     # every K and V row of the prompt once (8 kv heads x 128 dims, bf16)
        out["prompt_attention"].append(e)
ft = by_total_grid(counter("dec_fetch", "FETCH_SIZE", "attn_decode_fused_kernel"))
du = by_total_grid(durations("dec_trace", "attn_decode_fused_kernel"))
for k in ft:
    for grid, v in sorted(ft[k].items()):
        us = statistics.median(du[k][grid]) if grid in du.get(k, {}) else None
        kib = statistics.median(v)
        e = {"kernel": k, "grid_threads": grid, "launches": len(v), "hbm_fetch_bytes_per_launch": 2 * kib * 1024, "us_per_launch": us}
        if us:
            e["hbm_fetch_gb_per_s"] = 2 * kib * 1024 / (us * 1e-6) / 1e9
        out["decode_attention"].append(e)
out["note"] = ("prompt attention: one workgroup per (16 query rows, kv head), 104-456 workgroups on 256 CUs, two MFMA products per 32-token step; "
               "decode attention (8 streams): the talker's launches read contexts of 130-350 tokens per row (K and V rows of 256 B per kv head), the "
               "predictor's at most 17; both are bound by the latency of their dependent requests, not by bytes or the matrix cores")
json.dump(out, open(os.path.join(O, "r02_pmc_attention.json"), "w"), indent=1)
for sec in ("prompt_attention", "decode_attention"):
    for e in out[sec]:
        print(sec, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items()})
