#!/usr/bin/env python3
"""Developer aid: time per `fxc_channelize` call over ONE device-resident stream of 2^18 samples (the drop-in's
`_spectrometer_poly`, effex.py:530-555), 200 calls queued back to back, at several channel counts.

    python tools/probe_channelize_latency.py
"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from effex_amd import synth
from effex_amd.plan import FxPlan
for nchan in (8192, 4096, 3000, 1000, 6000):
    x = torch.from_numpy(synth.synth_iq(5, 1, 1, 2 ** 18)[0]).cuda()
    with FxPlan(1, nchan, 4, 2 ** 18) as p:
        for _ in range(10): out = p.channelize(x)
        p.sync(); t0 = time.perf_counter()
        for _ in range(200): out = p.channelize(x)
        p.sync()
        print(nchan, round((time.perf_counter() - t0) / 200 * 1e3, 4), "ms per channelize of one stream", flush=True)
