#!/usr/bin/env python3
"""Developer aid: F+X straight from RTL-SDR bytes on the headline shape (uint8 I,Q resident in HBM: 4 B per sample
pair instead of 16) vs the complex64 path.  Kernel time from the plan's HIP events."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from effex_amd.plan import FxPlan
    num_samp, frames = 262144, int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    nchan = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    g = torch.Generator(device="cuda").manual_seed(1)
    u8 = torch.randint(0, 256, (frames, 2, num_samp, 2), dtype=torch.uint8, device="cuda", generator=g)
    with FxPlan(2, nchan, 4, num_samp) as plan:
        for remove_dc in (True, False):
            for _ in range(2):
                plan.fx_accumulate_u8(u8, remove_dc=remove_dc)
            plan.sync()
            plan.kernel_profiling(True)
            plan.kernel_time(reset=True)
            ts = []
            for _ in range(8):
                plan.timer_start()
                plan.fx_accumulate_u8(u8, remove_dc=remove_dc)
                ts.append(plan.timer_stop())
            kms, n = plan.kernel_time(reset=True)
            plan.kernel_profiling(False)
            ts.sort()
            ms = ts[len(ts) // 2]
            print(json.dumps({"nchan": nchan, "remove_dc": remove_dc, "frames": frames, "call_ms": round(ms, 3), "fused_kernel_ms": round(kms / n, 3),
                              "Msamples_per_s_call": round(frames * num_samp / ms / 1e3, 1),
                              "Msamples_per_s_kernel": round(frames * num_samp / (kms / n) / 1e3, 1),
                              "u8_GBps_kernel": round(frames * num_samp * 4 / (kms / n) / 1e6, 1)}))


def two_step(frames=2500):
    """The same byte source through the separate device steps: convert (+ DC removal) to complex64, then F+X."""
    import torch
    from effex_amd.plan import FxPlan
    num_samp = 262144
    g = torch.Generator(device="cuda").manual_seed(1)
    u8 = torch.randint(0, 256, (frames, 2, num_samp, 2), dtype=torch.uint8, device="cuda", generator=g)
    with FxPlan(2, 4096, 4, num_samp) as plan:
        for _ in range(2):
            plan.fx_accumulate(plan.convert_u8(u8, remove_dc=True))
        plan.sync()
        ts = []
        for _ in range(5):
            plan.timer_start()
            plan.fx_accumulate(plan.convert_u8(u8, remove_dc=True))
            ts.append(plan.timer_stop())
        ts.sort()
        ms = ts[len(ts) // 2]
        print(json.dumps({"two_step": "convert_u8(remove_dc) then fx_accumulate", "frames": frames, "call_ms": round(ms, 3),
                          "Msamples_per_s_call": round(frames * num_samp / ms / 1e3, 1)}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "two-step":
        two_step()
        sys.exit(0)
    main()
