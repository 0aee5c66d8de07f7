#!/usr/bin/env python3
"""Developer aid: more than 8 antennas (F-only tiled kernel + X-engine) at the headline channel count, integrated and as rows,
for A/B runs of the X-engine (FXC_XENGINE=block: the vector kernel over blocks of 8 antennas; default: the matrix-core kernel).
One JSON line per shape; the kernels' own times are in a rocprofv3 trace of this script."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ants", type=int, nargs="+", default=[12, 16, 24, 32, 48, 64])
    ap.add_argument("--nchan", type=int, nargs="+", default=[4096])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rows", action="store_true")
    ap.add_argument("--copy-out", action="store_true", help="collect results through the plan's slot + a host copy (fxc_finalize_async)")
    args = ap.parse_args()
    import torch
    from effex_amd.plan import FxPlan, pinned_empty, synth_fill
    num_samp = 2 ** 18
    for nchan in args.nchan:
        for n_ant in args.ants:
            n_chunks = max(1, 2048 // n_ant)
            x = torch.empty((n_chunks, n_ant, num_samp), dtype=torch.complex64, device="cuda")
            synth_fill(x, 1234, delays=[a % 7 for a in range(n_ant)])
            with FxPlan(n_ant, nchan, 4, num_samp, dev=bool(os.environ.get("FXC_XENGINE"))) as plan:      # (route knobs: developer library only)
                outs = [pinned_empty((plan.n_baselines, nchan), "complex128") for _ in range(2)]

                def fn(k):
                    for j in range(k):
                        if args.rows:
                            plan.fx_rows(x)
                        else:
                            plan.fx_accumulate(x)
                            plan.finalize_async("SPECTRUM", 2.4e6, reset=True, out=None if args.copy_out else outs[j & 1])
                            if j > 0:
                                plan.finalize_wait()
                    if not args.rows:
                        plan.finalize_wait()
                fn(2)
                plan.sync()
                ms = []
                for _ in range(args.reps):
                    plan.timer_start()
                    fn(4)
                    ms.append(plan.timer_stop() / 4)
                ms.sort()
                med = ms[len(ms) // 2]
                algo = n_chunks * n_ant * num_samp * 8
                print(json.dumps({"xengine": os.environ.get("FXC_XENGINE", "mfma"), "n_ant": n_ant, "n_baselines": plan.n_baselines,
                                  "nchan": nchan, "n_chunks": n_chunks, "rows": args.rows, "median_ms": round(med, 3),
                                  "algorithmic_GBps": round(algo / med / 1e6, 1), "frac_of_8TBs": round(algo / med / 1e6 / 8000, 4)}),
                      flush=True)
            del x
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
