#!/usr/bin/env python3
"""Developer aid: what the FIRST Correlator of a process costs at a --resolution -- plan creation (the kernel built for the channel count:
pre-built code object, run-time cache, or hiprtc) and the first _run_task() -- against later ones.

    python tools/probe_first_use.py 1000        (one channel count per process: run it once per count)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t_import = time.perf_counter()
import numpy as np
import torch
from effex_amd import synth
from effex_amd.correlator import Correlator, SyntheticSource
torch.zeros(1, device="cuda")          # the HIP context is not what is being measured
t_import = time.perf_counter() - t_import
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
x = synth.synth_iq(5, 1, 2, 2 ** 18)[0]
t0 = time.perf_counter()
cor = Correlator(source=SyntheticSource(), mode="SPECTRUM", nbins=nb)
t_create = time.perf_counter() - t0
cor._state = 'RUN'
cor.gpu_iq_0[:] = x[0]
cor.gpu_iq_1[:] = x[1]
t0 = time.perf_counter()
cor._run_task()
t_first = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(50):
    cor._run_task()
t_later = (time.perf_counter() - t0) / 50
plan = getattr(cor, "_fx_plan", None)
info = plan.info if plan is not None else {}
print(json.dumps({"nbins": nb, "correlator_create_s": round(t_create, 4), "first_run_task_s": round(t_first, 4), "later_run_task_ms": round(t_later * 1e3, 4),
                  "specialised": info.get("specialised"), "code_object": {0: None, 1: "built by hiprtc", 2: "run-time cache", 3: "pre-built"}.get(info.get("spec_source")),
                  "spec_seconds": round(float(info.get("spec_seconds", 0.0)), 4), "imports_and_context_s": round(t_import, 2)}), flush=True)
cor.close()
