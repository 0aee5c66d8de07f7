#!/usr/bin/env python3
"""Developer aid: turn the raw rocprofv3 output of tools/collect_profiles.sh into the summaries kept under profiles/rNN/:
per workload the kernel-stats CSV as rocprofv3 wrote it, and a JSON with per-kernel mean duration (first, cold launch
listed separately) and HBM traffic per launch from the PMC passes (FETCH_SIZE doubled: on gfx950 it tallies 64 B per
128-B request of a wide streaming read, MI355X_MICROARCH.md §HBM; WRITE_SIZE exact; both reported in KiB by rocprofv3).

    python3 tools/summarize_profiles.py gpurun_out/r02
"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def read_csv(path):
    with open(path, newline="") as fh:
        return list(csv.DictReader(fh))


def find(dirpath, suffix):
    hits = glob.glob(os.path.join(dirpath, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def main():
    base = sys.argv[1]
    raw = os.path.join(base, "raw")
    try:
        head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except Exception:
        head = ""
    head = head or os.environ.get("FXC_HEAD", "unknown (no .git on the GPU box: see the commit that added this file)")
    names = sorted({os.path.basename(p)[:-len("_trace")] for p in glob.glob(os.path.join(raw, "*_trace"))})
    for name in names:
        summary = {"workload": name, "head": head, "kernels": {}}
        stats = find(os.path.join(raw, name + "_trace"), "kernel_stats.csv")
        if stats:
            shutil.copy(stats, os.path.join(base, "kernel_stats_%s.csv" % name))
        trace = find(os.path.join(raw, name + "_trace"), "kernel_trace.csv")
        if trace:
            dur = defaultdict(list)
            for row in read_csv(trace):
                dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
            for k, v in dur.items():
                if sum(v) < 50.0:          # under 50 us in all: not worth a line
                    continue
                warm = v[1:] if len(v) > 1 else v
                summary["kernels"][k] = {"launches": len(v), "first_launch_us": round(v[0], 1),
                                         "mean_us_excluding_first": round(sum(warm) / len(warm), 1),
                                         "min_us": round(min(v), 1), "max_us": round(max(v), 1)}
                summary["kernels"][k]["mean_us_all_launches"] = round(sum(v) / len(v), 1)
                if len(v) <= 32:       # launch by launch: the first launches after idle run slower (clock ramp)
                    summary["kernels"][k]["durations_us"] = [round(t, 1) for t in v]
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cc = find(os.path.join(raw, "%s_%s" % (name, counter)), "counter_collection.csv")
            if not cc:
                continue
            per = defaultdict(list)
            for row in read_csv(cc):
                if row["Counter_Name"] == counter:
                    per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]) * 1024.0)     # KiB -> bytes
            for k, v in per.items():
                if k in summary["kernels"]:
                    warm = v[1:] if len(v) > 1 else v
                    summary["kernels"][k][counter + "_bytes_per_launch_raw"] = round(sum(warm) / len(warm))
        # effective shader clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the duration of the same dispatch
        # in the same pass -- under 2.4 GHz means the chip is holding its clock down under load (power cap)
        gdir = os.path.join(raw, name + "_GRBM_GUI_ACTIVE")
        cc, tr = find(gdir, "counter_collection.csv"), find(gdir, "kernel_trace.csv")
        if cc and tr:
            dur = {row["Dispatch_Id"]: (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) for row in read_csv(tr)}
            per = defaultdict(list)
            for row in read_csv(cc):
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and dur.get(row["Dispatch_Id"], 0) > 0:
                    per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]) / 8.0 / dur[row["Dispatch_Id"]])
            for k, v in per.items():
                if k in summary["kernels"]:
                    warm = v[1:] if len(v) > 1 else v
                    summary["kernels"][k]["effective_clock_GHz_under_pmc"] = round(sum(warm) / len(warm), 3)
        for k, d in summary["kernels"].items():
            if "FETCH_SIZE_bytes_per_launch_raw" in d:
                d["hbm_read_bytes_per_launch"] = 2 * d["FETCH_SIZE_bytes_per_launch_raw"]
                d["hbm_traffic_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d.get("WRITE_SIZE_bytes_per_launch_raw", 0)
        log = os.path.join(raw, name + "_trace.log")
        if os.path.isfile(log):          # bench.py's own JSON line from the same profiled process (HIP-event timing)
            for line in open(log):
                if line.startswith("{") and "roofline" in line:
                    try:
                        d = json.loads(line)
                        summary["bench_line_of_this_profiled_run"] = {
                            "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"], "value": d["value"],
                            "roofline": d["roofline"]}
                    except ValueError:
                        pass
        summary["note"] = ("durations: rocprofv3 --kernel-trace; traffic: separate --pmc passes, FETCH_SIZE x 2 (gfx950 wide "
                           "streaming reads), WRITE_SIZE exact; means exclude each kernel's first launch")
        with open(os.path.join(base, "summary_%s.json" % name), "w") as fh:
            json.dump(summary, fh, indent=1)
        if name == "bench":      # what bench.py reports as roofline.traffic (profiles/*/pmc_hbm_traffic.json)
            for k, d in summary["kernels"].items():
                if k.startswith("fx_fused4096_kernel") and "hbm_traffic_bytes_per_launch" in d:
                    frames, algo = 10000, 10000 * 2 * 262144 * 8
                    with open(os.path.join(base, "pmc_hbm_traffic.json"), "w") as fh:
                        json.dump({"command": "bash tools/collect_profiles.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate "
                                              "passes, --kernel-trace, on python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "
                                              "--no-power --no-other-configs --no-verify)",
                                   "head": head, "kernel": k, "frames_per_launch": frames, "algorithmic_bytes_per_launch": algo,
                                   "FETCH_SIZE_bytes_raw": d["FETCH_SIZE_bytes_per_launch_raw"],
                                   "FETCH_SIZE_bytes_corrected_x2": d["hbm_read_bytes_per_launch"],
                                   "WRITE_SIZE_bytes": d.get("WRITE_SIZE_bytes_per_launch_raw", 0),
                                   "hbm_traffic_bytes_per_launch": d["hbm_traffic_bytes_per_launch"],
                                   "traffic_over_algorithmic": d["hbm_traffic_bytes_per_launch"] / algo,
                                   "note": "gfx950: FETCH_SIZE counts 64 B per 128 B request for wide streaming reads -> doubled "
                                           "(MI355X_MICROARCH.md HBM section); WRITE_SIZE exact.  Writes: one 32 KiB float32 raw row "
                                           "per 16 chunk pairs + one leading-part row per workgroup (fx_fused4096.h::RangeWalk)."},
                                  fh, indent=1)
        print(name, json.dumps(summary["kernels"])[:600])


if __name__ == "__main__":
    main()
