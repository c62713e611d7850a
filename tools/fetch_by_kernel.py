"""HBM fetch per launch of every kernel of a run, from a rocprofv3 --pmc FETCH_SIZE --kernel-trace pass:
    python tools/fetch_by_kernel.py <output dir of the pass>
Prints kernel, grid, launches, median bytes fetched (gfx950: FETCH_SIZE counts 128-B read requests as 64 B, hence x2)."""
import collections
import csv
import glob
import os
import statistics
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        per[(r["Kernel_Name"].replace("void ", "")[:70], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))].append(float(r["Counter_Value"]))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 6          # (lines; `| head` would end this script with a broken pipe)
shown = 0
for (k, g, w), v in sorted(per.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1])):
    if "at::native" in k or "rocclr" in k:
        continue
    shown += 1
    if shown > limit:
        break
    print(f"{k:72s} grid {g:>8s} wg {w:>5s} x{len(v):4d}  fetch {2 * statistics.median(v) * 1024 / 1e6:9.3f} MB")
