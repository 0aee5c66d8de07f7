#!/usr/bin/env python3
"""Developer aid: per-launch SQ / GRBM counter means of the headline kernel from tools/collect_sq.sh, with the ratios
DESIGN.md §6 quotes (per wave and spectrum pair; SQ cycle counters are in quad-cycles, MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    base = sys.argv[1]
    vals = defaultdict(list)
    dur = []
    for cc in glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True):
        with open(cc, newline="") as fh:
            for row in csv.DictReader(fh):
                if "fx_fused4096_kernel" in row["Kernel_Name"]:
                    vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    if row["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
                        dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    mean = {k: sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0] for k, v in vals.items()}   # first launch dropped
    out = {"counters_per_launch": {k: round(v, 1) for k, v in sorted(mean.items())}, "launch_us_under_pmc": [round(t, 1) for t in dur]}
    frames, waves = 10000 * 64, 256 * 8            # spectrum pairs per launch, waves per launch
    steps_per_wave = frames / 256.0
    d = {}
    if "SQ_INSTS_VALU" in mean:
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if k in mean:
                d[k.replace("SQ_INSTS_", "insts_per_wave_step_")] = round(mean[k] / waves / steps_per_wave, 1)
    if "SQ_WAVE_CYCLES" in mean:
        wc = mean["SQ_WAVE_CYCLES"]
        d["wave_cycles_per_step"] = round(wc * 4 / waves / steps_per_wave, 0)
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in mean:
                d[k + "_over_wave_cycles"] = round(mean[k] / wc, 4)
    out["derived"] = d
    out["note"] = ("SQ_* cycle counters count quad-cycles summed over waves; the ACTIVE_INST_* group comes from a different "
                   "pass than SQ_WAVE_CYCLES (compare within a pass); GRBM_GUI_ACTIVE / 8 / launch time = effective clock")
    if "GRBM_GUI_ACTIVE" in mean and dur:
        out["derived"]["effective_clock_GHz_under_pmc"] = round(mean["GRBM_GUI_ACTIVE"] / 8 / (sum(dur[-3:]) / len(dur[-3:]) * 1e-6) / 1e9, 3)
    with open(os.path.join(base, "..", "sq_counters.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
