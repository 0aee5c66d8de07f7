#!/usr/bin/env python3
"""Developer aid: per-launch means of the SQ counters tools/collect_spec.sh collected, for the kernel whose name contains
argv[2] (first launch dropped), with the ratios that say where a wave's time goes (SQ cycle counters are quad-cycles summed
over waves, MI355X_MICROARCH.md; compare counters within one pass only)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    base, pat = sys.argv[1], sys.argv[2]
    vals = defaultdict(list)
    dur = []
    for cc in glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True):
        with open(cc, newline="") as fh:
            for row in csv.DictReader(fh):
                if pat in row["Kernel_Name"]:
                    vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    if row["Counter_Name"] == "SQ_WAVE_CYCLES":
                        dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    mean = {k: sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0] for k, v in vals.items()}
    out = {"kernel": pat, "counters_per_launch": {k: round(v, 1) for k, v in sorted(mean.items())},
           "launch_us_under_pmc": [round(t, 1) for t in dur]}
    d = {}
    if "SQ_WAVE_CYCLES" in mean:
        wc = mean["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in mean:
                d[k + "_over_wave_cycles"] = round(mean[k] / wc, 4)
        if "SQ_WAVES" in mean:
            d["wave_cycles_per_wave"] = round(wc * 4 / mean["SQ_WAVES"], 0)
    if "SQ_INSTS_VALU" in mean and "SQ_WAVES" in mean:
        pass
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
        if k in mean:
            d[k.lower() + "_per_launch_M"] = round(mean[k] / 1e6, 2)
    if "SQ_LDS_BANK_CONFLICT" in mean and "SQ_LDS_IDX_ACTIVE" in mean and mean["SQ_LDS_IDX_ACTIVE"] > 0:
        d["lds_bank_conflict_share_of_lds_cycles"] = round(mean["SQ_LDS_BANK_CONFLICT"] / mean["SQ_LDS_IDX_ACTIVE"], 4)
    out["derived"] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
