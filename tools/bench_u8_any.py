#!/usr/bin/env python3
"""Developer aid: receiver bytes (uint8 I,Q) with DC removal through F + X at any channel count, 1 024 chunk pairs of 2^18 samples.

    python tools/bench_u8_any.py 1000,96,360      (FXC_MIXED_U8=0: the conversion pass in front, as before round 4's last step)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from effex_amd.plan import FxPlan
    num_samp, n_chunks = 2 ** 18, 1024
    u8 = torch.randint(0, 256, (n_chunks, 2, num_samp, 2), dtype=torch.uint8, device="cuda")
    for nchan in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1000").split(",")]:
        with FxPlan(2, nchan, 4, num_samp, dev=bool(os.environ.get("FXC_MIXED_U8"))) as plan:      # (route knobs: developer library only)
            plan.fx_accumulate_u8(u8, remove_dc=True)
            plan.finalize()
            ms = []
            for _ in range(5):
                plan.timer_start()
                plan.fx_accumulate_u8(u8, remove_dc=True)
                ms.append(plan.timer_stop())
                plan.finalize()
            ms.sort()
            print(json.dumps({"nchan": nchan, "path": plan.path, "median_ms": round(ms[2], 3),
                              "Msamples_per_s": round(n_chunks * num_samp / ms[2] / 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
