#!/usr/bin/env python3
"""Developer aid: time the fused kernel alone (HIP events around each launch) on device-resident
synthetic frames.  FXCORR_LIB selects the library build, so variants can be A/B-ed in one gpurun:

    python tools/kbench.py --frames 4096 --reps 8 [--rows]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--num-samp", type=int, default=262144)
    ap.add_argument("--rows", action="store_true")
    ap.add_argument("--nchan", type=int, default=4096)
    ap.add_argument("--ntaps", type=int, default=4)
    ap.add_argument("--path", default=None)
    ap.add_argument("--tag", default=os.environ.get("FXCORR_LIB", "in-tree"))
    args = ap.parse_args()
    import torch
    from effex_amd.plan import FxPlan, synth_fill
    x = torch.empty((args.frames, 2, args.num_samp), dtype=torch.complex64, device="cuda")
    synth_fill(x, 1234)
    import numpy as np
    window = np.array([0.4, 0.3, 0.2, 0.1][: args.ntaps]) if args.nchan == 1 else None
    plan = FxPlan(2, args.nchan, args.ntaps, args.num_samp, window=window, path=args.path)
    for _ in range(2):
        (plan.fx_rows(x) if args.rows else plan.fx_accumulate(x))
    plan.sync()
    plan.kernel_profiling(True)
    times = []
    for _ in range(args.reps):
        (plan.fx_rows(x) if args.rows else plan.fx_accumulate(x))
        ms, n = plan.kernel_time(reset=True)
        times.append(ms / n)
    times.sort()
    gb = args.frames * 2 * args.num_samp * 8 / 1e9
    med = times[len(times) // 2]
    print(json.dumps({"tag": args.tag, "path": plan.path, "nchan": args.nchan, "ntaps": args.ntaps, "frames": args.frames, "median_ms": round(med, 4), "min_ms": round(times[0], 4),
                      "GBps_median": round(gb / med * 1e3, 1), "GBps_best": round(gb / times[0] * 1e3, 1),
                      "frac_8TBs": round(gb / med * 1e3 / 8000, 4)}))


if __name__ == "__main__":
    main()
