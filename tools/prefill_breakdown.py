"""Per-kernel time of the last talker prefill in a rocprofv3 kernel trace of `tools/microbench.py prefill`.

    python tools/prefill_breakdown.py <dir with *_kernel_trace.csv>
"""
import collections
import csv
import glob
import os
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    # a prefill ends with the pick of its first token; the one before marks the previous prefill's end
    ends = [i for i, r in enumerate(rows) if "argmax" in r["Kernel_Name"] or "pick" in r["Kernel_Name"] or "sample" in r["Kernel_Name"]]
    last = rows[ends[-2] + 1:ends[-1] + 1] if len(ends) >= 2 else rows[-500:]
    last = [r for r in last if "at::native" not in r["Kernel_Name"] and "Cijk" not in r["Kernel_Name"]]
    per = collections.defaultdict(list)
    for r in last:
        per[r["Kernel_Name"].replace("void ", "")[:64] + f" grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"].append(dur(r))
    span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
    print(f"last prefill: {len(last)} launches, span {span / 1e3:.1f} us, busy {sum(dur(r) for r in last) / 1e3:.1f} us")
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {k:96s} x{len(v):3d} avg {sum(v) / len(v) / 1e3:7.2f} us  sum {sum(v) / 1e3:8.1f} us")


if __name__ == "__main__":
    main()
