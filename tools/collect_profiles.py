"""Turns the output of tools/refresh_profiles.sh (gpurun_out/refresh/) into the files kept under profiles/.

    python tools/collect_profiles.py [tag]        # tag defaults to v1: {RND}_bench_line_<tag>.json ...
"""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import subprocess
import sys

RND = "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", "refresh")
P = os.environ.get("QTTS_PROFILES_OUT", os.path.join(ROOT, "profiles"))     # refresh_profiles.sh collects on the box


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def med(name, counter, match):
    f = glob.glob(os.path.join(R, name, "**", "*counter_collection.csv"), recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if match in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return statistics.median(v), len(v)


def write_traffic(P):
    """The PMC passes over tools/microbench.py dominant -> {RND}_pmc_traffic.json (+ the raw counter files of the
    predictor's passes)."""
    os.makedirs(P, exist_ok=True)
    fs, n = med("pmc_fetch", "FETCH_SIZE", "linear2_kernel<8, 1, true, 1")
    ws, _ = med("pmc_write", "WRITE_SIZE", "linear2_kernel<8, 1, true, 1")
    traffic = {"kernel": "linear2_kernel<8,1,true,1> (talker RMSNorm + gate/up GEMM + SwiGLU launch, M=8, 1.7B shape: the most bytes per launch)",
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/microbench.py dominant",
               "fetch_size_kib_median": fs, "write_size_kib_median": ws,
               "correction": "HBM bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 tallies 128-B streaming read requests at 64 B)",
               "traffic_bytes_per_launch": 2 * fs * 1024 + ws * 1024, "algorithmic_bytes_per_launch": 50462720,
               "launches_sampled": n}
    if glob.glob(os.path.join(R, "pmc_fetch_pred", "**", "*counter_collection.csv"), recursive=True):
        pf, pn = med("pmc_fetch_pred", "FETCH_SIZE", "linear2_kernel<8, 1, false, 1")
        pw, _ = med("pmc_write_pred", "WRITE_SIZE", "linear2_kernel<8, 1, false, 1")
        traffic["predictor_gate_up"] = {
            "kernel": "linear2_kernel<8,1,false,1> (predictor RMSNorm + gate/up GEMM + SwiGLU launch, M=8, hidden 1024, intermediate 3072: "
                      "75 launches per frame, the family with the largest summed time)",
            "source": "QTTS_DOMINANT=predictor, same two passes (five weight matrices in turn, 63 MB: more than the 32 MB of L2)",
            "fetch_size_kib_median": pf, "write_size_kib_median": pw,
            "traffic_bytes_per_launch": 2 * pf * 1024 + pw * 1024, "algorithmic_bytes_per_launch": 2 * 3072 * 1024 * 2 + 8 * 1024 * 2 + 8 * 3072 * 2,
            "launches_sampled": pn}
        for name in ("fetch_pred", "write_pred"):
            shutil.copy(glob.glob(os.path.join(R, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)[0],
                        os.path.join(P, f"{RND}_pmc_{name}_counter_collection.csv"))
    json.dump(traffic, open(os.path.join(P, f"{RND}_pmc_traffic.json"), "w"), indent=1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic-only":      # (on the box, before the bench line that quotes the file)
        write_traffic(os.path.join(ROOT, "profiles"))
        return
    tag = sys.argv[1] if len(sys.argv) > 1 else "v1"
    os.makedirs(P, exist_ok=True)
    for old in glob.glob(os.path.join(P, f"{RND}_bench_*_v*")):
        os.remove(old)
    bench = last_json(os.path.join(R, "bench_line.json"))
    json.dump(bench, open(os.path.join(P, f"{RND}_bench_line_{tag}.json"), "w"))
    header = (f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline   (MI355X; "
              f"bench line of the same build without the profiler: profiles/{RND}_bench_line_{tag}.json)")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profile_summary.py"), glob.glob(os.path.join(R, "trace", "*"))[0],
                    os.path.join(P, f"{RND}_bench_summary_{tag}.txt"), header], check=True, stdout=subprocess.DEVNULL)
    shutil.copy(glob.glob(os.path.join(R, "trace", "*", "*_kernel_stats.csv"))[0], os.path.join(P, f"{RND}_bench_kernel_stats_{tag}.csv"))
    for name in ("fetch", "write"):
        shutil.copy(glob.glob(os.path.join(R, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)[0],
                    os.path.join(P, f"{RND}_pmc_{name}_counter_collection.csv"))
    for f in ("config_0.6B_1stream", "config_fp8_8streams", "bench_2rank_rehearsal", "api_quick_benchmark", "api_batch_async_fused", "api_batch_async_workers",
              "api_batch_async_fused_sampling", "api_quick_benchmark_sampling"):
        src = os.path.join(R, f + ".json")
        if os.path.exists(src):
            json.dump(last_json(src), open(os.path.join(P, f"{RND}_{f}.json"), "w"))

    if os.path.exists(os.path.join(R, "bench_admission_serial.json")):
        json.dump({"source": "python bench.py (arrivals workload: 8 streams 50 ms apart, 125 frames each), same box, same build",
                   "join (default): requests admitted while a frame is in flight, prefill on a second stream, the next frame waits for it": bench["arrivals"],
                   "serial (QTTS_ADMISSION=serial): prefill between two frames": last_json(os.path.join(R, "bench_admission_serial.json"))["arrivals"],
                   "beside (QTTS_ADMISSION=beside): prefill on a second stream, joined once finished": last_json(os.path.join(R, "bench_admission_beside.json"))["arrivals"]},
                  open(os.path.join(P, f"{RND}_arrivals_admission_ab.json"), "w"), indent=1)
    write_traffic(P)

    rows_txt = open(os.path.join(R, "rows.txt")).read()
    us = {}
    for line in rows_txt.splitlines():
        parts = line.split()
        if len(parts) > 4 and parts[1] == "M=" and parts[2] == "200:":
            us[parts[0]] = float(parts[4])
    f = glob.glob(os.path.join(R, "pmc_mfma", "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "linear_rows" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    which = {"linear_rows_kernel<8, 4, 1, false, false>": "gate_up", "linear_rows_kernel<8, 4, 2, false, false>": "down",
             "linear_rows_kernel<8, 4, 0, false, false>": "qkv"}
    out = {"source": "QTTS_ROWS=200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -- python3 tools/microbench.py rows (200 prompt rows, talker 1.7B shapes)",
           "note": "SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles x MFMA instructions (v_mfma_f32_16x16x32_bf16); utilisation = busy cycles / (launch duration without the profiler x 2.4 GHz x 1024 SIMDs); durations from tools/microbench.py rows of the same build",
           "kernels": {}}
    for k, v in acc.items():
        b = statistics.median(v["SQ_VALU_MFMA_BUSY_CYCLES"])
        e = {"mfma_busy_cycles_per_launch": b, "mfma_instructions": b / 16, "launches": len(v["SQ_VALU_MFMA_BUSY_CYCLES"])}
        if which.get(k) in us:
            t = us[which[k]]
            e.update(us_per_launch=t, mfma_utilisation=b / (t * 1e-6 * 2.4e9 * 1024), tflops_issued=b / 16 * 16384 / (t * 1e-6) / 1e12)
        out["kernels"][k] = e
    json.dump(out, open(os.path.join(P, f"{RND}_pmc_mfma.json"), "w"), indent=1)

    sweep = []
    for S in (1, 8, 16, 24, 32, 64, 128, 256):
        x = bench if S == 8 else last_json(os.path.join(R, f"sweep_{S}.json"))
        sweep.append({"streams_per_gpu": S, "codec_tokens_per_s": round(x["value"]), "ms_per_frame": round(x["ms_per_step"], 3),
                      "rtf": round(x["rtf"], 4), "frame_roofline_frac": round(x["frame_roofline"]["frac"], 3), "steps": x["steps"]})
    json.dump({"note": "python bench.py --streams S (1 GPU, MI355X, 1.7B shapes, greedy); the contract line is S=8", "sweep": sweep},
              open(os.path.join(P, f"{RND}_streams_sweep.json"), "w"), indent=1)

    if os.path.isdir(os.path.join(R, "trace32")):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profile_summary.py"), glob.glob(os.path.join(R, "trace32", "*"))[0],
                        os.path.join(P, f"{RND}_bench_summary_32streams.txt"),
                        "rocprofv3 --kernel-trace --stats -- python3 bench.py --streams 32 --steps 30 --warmup 5 --no-cpu-baseline --no-arrivals   (MI355X)"],
                       check=True, stdout=subprocess.DEVNULL)
    for name in ("soak", "soak_32rows"):
        if os.path.exists(os.path.join(R, name + ".json")):
            try:
                json.dump(last_json(os.path.join(R, name + ".json")), open(os.path.join(P, f"{RND}_{name}.json"), "w"), indent=1)
            except Exception as e:
                print(name, "output not collected:", e)
    pb = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prefill_breakdown.py"), os.path.join(R, "prefill_trace")],
                        capture_output=True, text=True, check=True).stdout
    with open(os.path.join(P, f"{RND}_prefill.txt"), "w") as fo:
        fo.write("tools/microbench.py prefill (talker 1.7B shapes, one prompt per call, wall time of FrameEngine.add_stream, MI355X):\n")
        fo.write(open(os.path.join(R, "prefill.txt")).read() + "\nrocprofv3 --kernel-trace of the same, 200 tokens:\n" + pb)
        fo.write("\ntools/microbench.py rows (prompt GEMMs alone; cold = 28 weight matrices in turn, hot = one):\n" + rows_txt)
    print("bench", round(bench["value"]), round(bench["ms_per_step"], 3), "first chunk", round(bench["first_chunk_ms_p50"], 2),
          "dominant us", round(bench["roofline"]["us_per_launch"], 2), "frac", round(bench["roofline"]["frac"], 3),
          "frame frac", round(bench["frame_roofline"]["frac"], 3), "cpu", round(bench["cpu_baseline"]["value"]))
    for s in sweep:
        print(s)
    for f in sorted(glob.glob(os.path.join(P, f"{RND}_api_*.json"))):
        d = json.load(open(f))
        print(os.path.basename(f), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items()
                                    if k in ("rtf", "first_frame_ms_p50", "first_chunk_ms_p50", "inter_chunk_ms_p50", "inter_chunk_ms_p99", "rtf_per_stream_p50")})
    print(open(os.path.join(R, "prefill.txt")).read())


if __name__ == "__main__":
    main()
