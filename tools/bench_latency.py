#!/usr/bin/env python3
"""Developer aid: latency of one reference-sized call (one chunk pair, device resident) on the headline shape:
automatic plan (frames split over workgroups for small calls) vs an explicit one-workgroup-per-chunk plan."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from effex_amd.plan import FxPlan, synth_fill
    num_samp = 262144
    for n_chunks in (1, 4, 16, 64):
        x = torch.empty((n_chunks, 2, num_samp), dtype=torch.complex64, device="cuda")
        synth_fill(x, 1234)
        for path in (None, "fused"):
            with FxPlan(2, 4096, 4, num_samp, path=path) as plan:
                for _ in range(5):
                    plan.fx_rows(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 200
                for _ in range(reps):
                    plan.fx_rows(x)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
                print(json.dumps({"n_chunks": n_chunks, "plan": path or "auto", "us_per_call": round(dt * 1e6, 1),
                                  "Msamples_per_s": round(n_chunks * num_samp / dt / 1e6, 1)}))


if __name__ == "__main__":
    main()
