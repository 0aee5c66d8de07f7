#!/usr/bin/env python3
"""Developer aid: throughput of fxc_channelize (the _spectrometer_poly drop-in, F-stage only, spectra written to
HBM) on device-resident streams: algorithmic traffic = 8 B in + 8 B out per sample."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from effex_amd.plan import FxPlan, synth_fill
    num_samp = 262144
    cases = ((4096, 4), (2048, 4), (1024, 4), (256, 4), (64, 4), (16, 4), (2048, 32), (4096, 32), (8192, 4))
    if len(sys.argv) > 1:       # "1000:4,96:4"
        cases = tuple(tuple(int(v) for v in c.split(":")) for c in sys.argv[1].split(","))
    for nchan, ntaps in cases:
        n_streams = 1024
        x = torch.empty((n_streams // 2, 2, num_samp), dtype=torch.complex64, device="cuda")
        synth_fill(x, 1234)
        xs = x.view(n_streams, num_samp)
        with FxPlan(1, nchan, ntaps, num_samp, dev=bool(os.environ.get("FXC_DEV"))) as plan:      # (FXC_DEV=1: the developer library and its knobs)
            out = plan.channelize(xs)
            plan.sync()
            ts = []
            for _ in range(5):
                del out
                plan.timer_start()
                out = plan.channelize(xs)
                ts.append(plan.timer_stop())
            ts.sort()
            ms = ts[len(ts) // 2]
            gb = n_streams * num_samp * 8 / 1e9
            print(json.dumps({"tag": os.environ.get("FXC_TAG", ""), "nchan": nchan, "ntaps": ntaps, "streams": n_streams, "median_ms": round(ms, 3),
                              "Msamples_per_s": round(n_streams * num_samp / ms / 1e3, 1),
                              "in_plus_out_GBps": round(2 * gb / ms * 1e3, 1)}), flush=True)
            del out
        del x


if __name__ == "__main__":
    main()
