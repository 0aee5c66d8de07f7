#!/usr/bin/env python3
"""Developer aid (for a rocprofv3 kernel trace): one reference-sized call -- one chunk pair, device resident -- a few hundred
times, SPECTRUM and CONTINUUM, with and without the DC pre-pass: which kernels a call is made of and how long each runs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effex_amd.plan import FxPlan, synth_fill

x = torch.empty((1, 2, 262144), dtype=torch.complex64, device="cuda")
synth_fill(x, 1234)
with FxPlan(2, 4096, 4, 262144) as plan:
    for mode, dc in (("SPECTRUM", False), ("CONTINUUM", False), ("SPECTRUM", True)):
        for _ in range(100):
            plan.fx_rows(x, mode, 2.4e6, remove_dc=dc)
        torch.cuda.synchronize()
