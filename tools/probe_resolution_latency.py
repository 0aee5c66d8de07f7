#!/usr/bin/env python3
"""Developer aid: wall time of the drop-in's `_run_task()` (pinned staging buffers in, host row out) per chunk pair of 2^18
samples at several `--resolution` values, powers of two and not.

    [PROBE_MODE=CONTINUUM] python tools/probe_resolution_latency.py [4096,8192,...]
"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from effex_amd import synth
from effex_amd.correlator import Correlator, SyntheticSource
x = synth.synth_iq(5, 1, 2, 2 ** 18)[0]
for nb in [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (4096, 1000, 96, 3000, 997, 12000, 8192, 2048, 6000):
    cor = Correlator(source=SyntheticSource(), mode=os.environ.get("PROBE_MODE", "SPECTRUM"), nbins=nb)
    cor._state = 'RUN'
    cor.gpu_iq_0[:] = x[0]; cor.gpu_iq_1[:] = x[1]
    for _ in range(20): cor._run_task()
    t0 = time.perf_counter()
    for _ in range(200): cor._run_task()
    print(nb, round((time.perf_counter() - t0) / 200 * 1e3, 4), "ms per _run_task", flush=True)
    cor.close()
