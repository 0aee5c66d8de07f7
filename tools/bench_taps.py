#!/usr/bin/env python3
"""Developer aid: the tiled path at ntaps > 4 (PFB pre-filter pass + one-tap tiled kernel) and at nchan 8192.

    python tools/bench_taps.py [--reps 5]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cases", default="2048:32,4096:8,512:16,4096:32,1024:8,8192:4,8192:8")
    args = ap.parse_args()
    import torch
    from effex_amd.plan import FxPlan, synth_fill
    num_samp, n_chunks = 2 ** 18, 1024
    x = torch.empty((n_chunks, 2, num_samp), dtype=torch.complex64, device="cuda")
    synth_fill(x, 1234)
    for case in args.cases.split(","):
        nchan, ntaps = (int(v) for v in case.split(":"))
        with FxPlan(2, nchan, ntaps, num_samp, dev=bool(os.environ.get("FXC_PREFILTER"))) as plan:      # (route knobs: developer library only)
            plan.fx_accumulate(x)
            plan.finalize()
            ms = []
            for _ in range(args.reps):
                plan.timer_start()
                plan.fx_accumulate(x)
                ms.append(plan.timer_stop())
                plan.finalize()
            ms.sort()
            med = ms[len(ms) // 2]
            gb = n_chunks * 2 * num_samp * 8 / 1e9
            print(json.dumps({"tag": os.environ.get("FXC_PREFILTER", "default"), "nchan": nchan, "ntaps": ntaps, "path": plan.path,
                              "median_ms": round(med, 3), "algorithmic_GBps": round(gb / med * 1e3, 1),
                              "frac_of_8TBs": round(gb / med * 1e3 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()
