"""Condenses a rocprofv3 kernel trace of bench.py into the text summary kept under profiles/.

    python tools/profile_summary.py <dir with *_kernel_trace.csv and *_kernel_stats.csv> <out.txt> "<header line>"

Besides the per-kernel totals it lists the decode gate/up kernels' launches by grid (the decode launches bench.py times)
and the launches of the run's last frame.
"""
import collections
import csv
import glob
import os
import statistics
import sys

# the decode gate/up launches (RMSNorm + GEMM + SwiGLU) of the talker and of the predictor: the launch with the most
# bytes, and the family with the largest summed time per frame
DOMINANT = ("linear2_kernel<8, 1, true, 1,", "linear2_kernel<8, 1, false, 1,")


def main():
    d, out, header = sys.argv[1], sys.argv[2], sys.argv[3]
    trace = glob.glob(os.path.join(d, "*_kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append(dur(r))
    total = sum(sum(v) for v in per.values())
    lines = [header, f"kernels launched {len(rows)}, busy time {total / 1e6:.2f} ms", ""]
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        name = k.replace("void ", "")[:60]
        lines.append(f"{name:60s} calls {len(v):6d} avg {statistics.mean(v) / 1e3:8.2f} us  total {sum(v) / 1e6:8.2f} ms "
                     f"{100 * sum(v) / total:5.1f}%")
    for dom in DOMINANT:
        lines += ["", f"{dom} by grid (x = workgroups*512 threads):"]
        by = collections.defaultdict(list)
        for r in rows:
            if dom in r["Kernel_Name"]:
                by[(int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))].append(dur(r))
        smallest = min(by) if by else None
        for k, v in sorted(by.items()):
            lines.append(f"  grid {k}: calls {len(v):5d} avg {statistics.mean(v) / 1e3:7.2f} us  median {statistics.median(v) / 1e3:7.2f} us"
                         f"  min {min(v) / 1e3:7.2f} us" + ("   <- the decode launch bench.py times" if k == smallest else ""))
    # one steady-state frame: from one predictor_input launch to the next
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("predictor_input_kernel")]
    if len(idx) > 3:
        a, b = idx[-2], idx[-1]
        fr = rows[a:b]
        span = int(rows[b]["Start_Timestamp"]) - int(fr[0]["Start_Timestamp"])
        agg = collections.defaultdict(list)
        for r in fr:
            agg[r["Kernel_Name"].replace("void ", "")[:48] + f" grid({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']})"].append(dur(r))
        lines += ["", f"last frame of the run: {len(fr)} launches, span {span / 1e3:.1f} us under the profiler, "
                      f"busy {sum(dur(r) for r in fr) / 1e3:.1f} us"]
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f"  {k:75s} x{len(v):3d} avg {statistics.mean(v) / 1e3:6.2f} us  sum {sum(v) / 1e3:7.1f} us")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
