#!/usr/bin/env python3
"""Developer aid: F + X integrated over 4.3 GB of samples for several antenna counts at one channel count.

    python tools/bench_ants.py [nchan=1000] [ants=3,4,8] [ntaps=4]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from effex_amd.plan import FxPlan, synth_fill
    nchan = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    ants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "3,4,8").split(",")]
    ntaps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    num_samp = 2 ** 18
    for n_ant in ants:
        n_chunks = 2048 // n_ant
        x = torch.empty((n_chunks, n_ant, num_samp), dtype=torch.complex64, device="cuda")
        synth_fill(x, 1234, delays=None if n_ant <= 8 else [a % 7 for a in range(n_ant)])
        with FxPlan(n_ant, nchan, ntaps, num_samp) as plan:
            plan.fx_accumulate(x)
            plan.finalize()
            ms = []
            for _ in range(5):
                plan.timer_start()
                plan.fx_accumulate(x)
                ms.append(plan.timer_stop())
                plan.finalize()
            ms.sort()
            gb = n_chunks * n_ant * num_samp * 8 / 1e9
            print(json.dumps({"n_ant": n_ant, "nchan": nchan, "ntaps": ntaps, "path": plan.path, "median_ms": round(ms[2], 3),
                              "frac_of_8TBs": round(gb / ms[2] * 1e3 / 8000, 4)}), flush=True)
        del x


if __name__ == "__main__":
    main()
