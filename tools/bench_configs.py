#!/usr/bin/env python3
"""Measure the other BASELINE.json configs on one MI355X: continuum (nchan = 1 streaming
limit and reference CONTINUUM semantics at N = 4096, S = 2^20), 8-antenna / 28-baseline, and the
per-chunk rows (time-series) variant of the headline config.  One JSON line per config.

    python tools/bench_configs.py [--reps 5]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(plan, fn, reps, calls=4):
    """fn(k) = k calls back to back (an integration's result is collected while the next one runs, as bench.py does);
    HIP events around them, per call."""
    fn(2)
    plan.sync()
    best = []
    for _ in range(reps):
        plan.timer_start()
        fn(calls)
        best.append(plan.timer_stop() / calls)
    best.sort()
    return best[len(best) // 2], best[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import numpy as np
    import torch
    from effex_amd.plan import FxPlan, pinned_empty, synth_fill
    from effex_amd.window import design_window
    bw = 2.4e6
    out = []

    def report(name, n_ant, nchan, ntaps, num_samp, n_chunks, mode, window=None, rows=False):
        x = torch.empty((n_chunks, n_ant, num_samp), dtype=torch.complex64, device="cuda")
        synth_fill(x, 1234, delays=None if n_ant <= 8 else [a % 7 for a in range(n_ant)])
        plan = FxPlan(n_ant, nchan, ntaps, num_samp, window=window)
        # results are delivered into pinned buffers named up front (fxc_finalize_async_to): nothing is copied on the host
        shape = (plan.n_baselines, nchan) if mode == "SPECTRUM" else (plan.n_baselines,)
        outs = [pinned_empty(shape, np.complex128) for _ in range(2)]
        if rows:
            def fn(k):
                for _ in range(k):
                    plan.fx_rows(x, mode, bw)
        else:
            def fn(k):
                for j in range(k):
                    plan.fx_accumulate(x)
                    plan.finalize_async(mode, bw, reset=True, out=outs[j & 1])
                    if j > 0:
                        plan.finalize_wait()
                plan.finalize_wait()
        med, best = timed(plan, fn, args.reps)
        samples = n_chunks * num_samp
        algo = n_chunks * n_ant * num_samp * 8
        line = {"config": name, "path": plan.path, "n_ant": n_ant, "nchan": nchan, "ntaps": ntaps, "num_samp": num_samp,
                "n_chunks": n_chunks, "mode": mode, "rows": rows, "median_ms": round(med, 3),
                "Msamples_per_s": round(samples / med / 1e3, 1), "algorithmic_GBps": round(algo / med / 1e6, 1),
                "frac_of_8TBs": round(algo / med / 1e6 / 8000, 4)}
        print(json.dumps(line), flush=True)
        out.append(line)
        plan.close()
        del x
        torch.cuda.empty_cache()

    report("configs[1] headline, integrate", 2, 4096, 4, 2 ** 18, 4096, "SPECTRUM")
    report("configs[1] headline, one row per frame (reference time series)", 2, 4096, 4, 2 ** 18, 4096, "SPECTRUM", rows=True)
    report("configs[2](ii) continuum, reference semantics N=4096, S=2^20", 2, 4096, 4, 2 ** 20, 1024, "CONTINUUM", rows=True)
    report("configs[2](i) continuum streaming limit nchan=1, S=2^20", 2, 1, 4, 2 ** 20, 2048, "CONTINUUM",
           window=np.array([0.4, 0.3, 0.2, 0.1]), rows=True)
    report("configs[4] 8 antennas, 28 baselines, N=4096", 8, 4096, 4, 2 ** 18, 512, "SPECTRUM")
    report("8 antennas, 28 baselines, N=2048 (tiled F-only kernel + X-engine)", 8, 2048, 4, 2 ** 18, 512, "SPECTRUM")
    report("8 antennas, 28 baselines, N=1024 (tiled F-only kernel + X-engine)", 8, 1024, 4, 2 ** 18, 512, "SPECTRUM")
    report("4 antennas, 6 baselines, N=4096", 4, 4096, 4, 2 ** 18, 1024, "SPECTRUM")
    report("16 antennas, 120 baselines, N=4096 (tiled F-only kernel + matrix-core X-engine)", 16, 4096, 4, 2 ** 18, 128, "SPECTRUM")
    report("32 antennas, 496 baselines, N=4096 (tiled F-only kernel + matrix-core X-engine)", 32, 4096, 4, 2 ** 18, 64, "SPECTRUM")
    report("N=2048 T=32 (reference test shape), 2 antennas", 2, 2048, 32, 2 ** 18, 256, "SPECTRUM")
    for nfft in (16, 32, 64, 128, 256, 512, 1024, 2048):   # the reference's --nfft at its fixed ntaps = 4 (effex.py:115,778)
        report("--nfft %d, integrate" % nfft, 2, nfft, 4, 2 ** 18, 4096, "SPECTRUM")
    report("--nfft 8192, integrate (split into two 4096-channel problems)", 2, 8192, 4, 2 ** 18, 1024, "SPECTRUM")
    report("headline shape, one chunk pair per call (the reference's call pattern)", 2, 4096, 4, 2 ** 18, 1, "SPECTRUM",
           rows=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "configs.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
