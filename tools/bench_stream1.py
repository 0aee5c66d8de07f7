#!/usr/bin/env python3
"""Developer aid: the continuum streaming limit (BASELINE configs[2]: nchan 1, num_samp 2^20, one scalar per chunk pair) on 2 048
device-resident chunk pairs -- ms per call and fraction of 8 TB/s.  FXCORR_LIB selects a variant build.

    python tools/bench_stream1.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    from effex_amd.plan import FxPlan, synth_fill
    num_samp, n_chunks = 2 ** 20, 2048
    x = torch.empty((n_chunks, 2, num_samp), dtype=torch.complex64, device="cuda")
    synth_fill(x, 1234)
    with FxPlan(2, 1, 4, num_samp, window=np.array([0.4, 0.3, 0.2, 0.1])) as plan:
        plan.fx_rows(x, "CONTINUUM", 2.4e6)
        plan.sync()
        ms = []
        for _ in range(7):
            plan.timer_start()
            plan.fx_rows(x, "CONTINUUM", 2.4e6)
            ms.append(plan.timer_stop())
        ms.sort()
        med = ms[len(ms) // 2]
        gb = n_chunks * 2 * num_samp * 8 / 1e9
        print(json.dumps({"tag": os.environ.get("FXCORR_LIB", "in-tree"), "median_ms": round(med, 4), "frac_of_8TBs": round(gb / med / 8000 * 1e3, 4)}))


if __name__ == "__main__":
    main()
