#!/usr/bin/env python3
"""Developer aid: run the 8-antenna / 28-baseline config a few times (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effex_amd.plan import FxPlan, synth_fill

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.empty((n_chunks, 8, 262144), dtype=torch.complex64, device="cuda")
synth_fill(x, 1234)
plan = FxPlan(8, 4096, 4, 262144)
for _ in range(6):
    plan.acc_reset()
    plan.fx_accumulate(x)
    plan.finalize("SPECTRUM")
plan.sync()
